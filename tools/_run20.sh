cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r20
timeout 900 python -m pytest tests/test_hip_lf.py -m gpu -q -x > gpurun_out/r20/pytest.log 2>&1; tail -3 gpurun_out/r20/pytest.log
echo "=== LF default"; timeout 300 python tools/bench_lf.py --steps 40 2>&1 | grep -v amdgpu | tail -8
echo "=== LF main stream high priority"; EGAZE_MAIN_PRIO=-1 timeout 300 python tools/bench_lf.py --steps 40 2>&1 | grep -v amdgpu | tail -8
echo "=== LF main stream high priority, bnsums off"; EGAZE_BNSUMS_FUSE=0 EGAZE_MAIN_PRIO=-1 timeout 300 python tools/bench_lf.py --steps 40 2>&1 | grep -v amdgpu | tail -8
echo "=== timeline (prio)"
rm -rf /tmp/lfprof
(cd /tmp && EGAZE_MAIN_PRIO=-1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lfprof -o lf -- python $GRAFT_REPO_ROOT/tools/bench_lf.py --steps 10 > /dev/null 2>&1)
python tools/lf_timeline.py /tmp/lfprof > gpurun_out/r20/timeline.txt 2>&1; tail -22 gpurun_out/r20/timeline.txt
