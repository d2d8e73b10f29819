cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r19
timeout 900 python -m pytest tests/test_hip_lf.py tests/test_hip_config5.py tests/test_hip_drivers.py -m gpu -q -x > gpurun_out/r19/pytest.log 2>&1; tail -5 gpurun_out/r19/pytest.log
echo "=== LF default"; timeout 300 python tools/bench_lf.py --steps 40 2>&1 | grep -v amdgpu | tail -8
echo "=== LF first-fuse off"; EGAZE_FIRST_FUSE=0 timeout 300 python tools/bench_lf.py --steps 40 2>&1 | grep -v amdgpu | tail -8
echo "=== LF first-fuse on, bnsums off"; EGAZE_BNSUMS_FUSE=0 timeout 300 python tools/bench_lf.py --steps 40 2>&1 | grep -v amdgpu | tail -8
echo "=== timeline"
rm -rf /tmp/lfprof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lfprof -o lf -- python $GRAFT_REPO_ROOT/tools/bench_lf.py --steps 10 > /dev/null 2>&1)
python tools/lf_timeline.py /tmp/lfprof > gpurun_out/r19/timeline.txt 2>&1; tail -30 gpurun_out/r19/timeline.txt
