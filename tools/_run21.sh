cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r21
timeout 900 python -m pytest tests/test_hip_lf.py -m gpu -q -x > gpurun_out/r21/pytest.log 2>&1; tail -3 gpurun_out/r21/pytest.log
for k in "A=0" "EGAZE_WGRAD_AFTER_DGRAD=1" "EGAZE_WGRAD_AFTER_DGRAD=1 EGAZE_BNSUMS_FUSE=0" "EGAZE_STREAMS=0" "EGAZE_STREAMS=0 EGAZE_BNSUMS_FUSE=0" "EGAZE_DETACH_WGRAD=0" ; do
echo "=== $k"; env $k timeout 300 python tools/bench_lf.py --steps 40 2>&1 | grep "metric=off"
done
