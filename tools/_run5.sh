cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
bash tools/ab_conv.sh "diag7 diag15 diag8" --dtype 1 --what dgrad --iters 20 --only "dec1" > gpurun_out/r5/ab.log 2>&1
bash tools/ab_conv.sh "diag7 diag15 diag8" --dtype 1 --what dgrad --iters 20 --only "enc17" >> gpurun_out/r5/ab.log 2>&1
cat gpurun_out/r5/ab.log
