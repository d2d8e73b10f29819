cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r15
timeout 600 python -m pytest tests/test_hip_lf.py tests/test_hip_drivers.py -m gpu -q > gpurun_out/r15/pytest.log 2>&1; tail -12 gpurun_out/r15/pytest.log
timeout 300 python tools/bench_lf.py > gpurun_out/r15/lf.log 2>&1; cat gpurun_out/r15/lf.log
