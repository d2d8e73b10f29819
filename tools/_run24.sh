cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r24
timeout 900 python -m pytest tests/test_hip_lf.py tests/test_hip_config5.py tests/test_hip_ops.py -m gpu -q -x > gpurun_out/r24/pytest.log 2>&1; tail -3 gpurun_out/r24/pytest.log
for k in "A=0" "EGZ_FIRST_DIRECT=0"; do
echo "=== $k"; env $k timeout 300 python tools/bench_lf.py --steps 40 2>&1 | grep "metric=off"
done
echo "=== timeline"
rm -rf /tmp/lfprof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lfprof -o lf -- python $GRAFT_REPO_ROOT/tools/bench_lf.py --steps 10 > /dev/null 2>&1)
python tools/lf_timeline.py /tmp/lfprof > gpurun_out/r24/timeline.txt 2>&1; head -12 gpurun_out/r24/timeline.txt
