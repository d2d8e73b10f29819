cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r16
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_lf.py -m gpu -q > gpurun_out/r16/pytest.log 2>&1; tail -3 gpurun_out/r16/pytest.log
echo "=== new (no spills in the 64-column configuration)"; timeout 200 python tools/bench_conv.py --dtype 1 --what fwd,dgrad --iters 20 2>&1 | grep -v amdgpu
echo "=== again"; timeout 200 python tools/bench_conv.py --dtype 1 --what fwd,dgrad --iters 20 2>&1 | grep "TOTAL\|enc3\|dec26\|enc7\|dec24"
