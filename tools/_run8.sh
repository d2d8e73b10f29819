cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r8
bash tools/ab_bench.sh "EGZ_EW_CAP=8192" "EGZ_EW_CAP=2048" "EGZ_EW_CAP=1024" "EGZ_EW_CAP=512" "EGZ_EW_CAP=256" "EGZ_EW_CAP=8192" "EGZ_EW_CAP=1024" > gpurun_out/r8/ab_bench.log 2>&1
cat gpurun_out/r8/ab_bench.log | grep -v "^{"; tail -5 gpurun_out/ab_bench.err
