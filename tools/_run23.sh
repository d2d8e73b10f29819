cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r23
timeout 900 python -m pytest tests/test_hip_lf.py tests/test_hip_config5.py -m gpu -q -x > gpurun_out/r23/pytest.log 2>&1; tail -3 gpurun_out/r23/pytest.log
for k in "A=0" "EGZ_BN_FIN_FUSE=0" "EGAZE_STREAMS=0"; do
echo "=== $k"; env $k timeout 300 python tools/bench_lf.py --steps 40 2>&1 | grep "metric=off"
done
echo "=== timeline"
rm -rf /tmp/lfprof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lfprof -o lf -- python $GRAFT_REPO_ROOT/tools/bench_lf.py --steps 10 > /dev/null 2>&1)
python tools/lf_timeline.py /tmp/lfprof > gpurun_out/r23/timeline.txt 2>&1; tail -22 gpurun_out/r23/timeline.txt
