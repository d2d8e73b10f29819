cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
bash tools/ab_conv.sh "diag1 diag2 diag4 diag7" --dtype 1 --what fwd,dgrad --iters 20 > gpurun_out/r4/ab_diag.log 2>&1
cat gpurun_out/r4/ab_diag.log
