cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r18
timeout 900 python -m pytest tests/test_hip_lf.py tests/test_hip_ops.py tests/test_hip_config5.py -m gpu -q -x > gpurun_out/r18/pytest.log 2>&1; tail -5 gpurun_out/r18/pytest.log
echo "=== LF fused BN sums"; timeout 300 python tools/bench_lf.py --steps 40 2>&1 | grep -v amdgpu | tail -8
echo "=== LF separate reduce"; EGAZE_BNSUMS_FUSE=0 timeout 300 python tools/bench_lf.py --steps 40 2>&1 | grep -v amdgpu | tail -8
for m in 1; do
echo "=== timeline BNSUMS_FUSE=$m"
rm -rf /tmp/lfprof
(cd /tmp && EGAZE_BNSUMS_FUSE=$m timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lfprof -o lf -- python $GRAFT_REPO_ROOT/tools/bench_lf.py --steps 10 > /dev/null 2>&1)
python tools/lf_timeline.py /tmp/lfprof > gpurun_out/r18/timeline_$m.txt 2>&1; tail -64 gpurun_out/r18/timeline_$m.txt
done
