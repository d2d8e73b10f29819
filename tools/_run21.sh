cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r21
python -m pytest tests -m gpu -q --maxfail=20 > gpurun_out/r21/pytest.log 2>&1; tail -3 gpurun_out/r21/pytest.log
EGAZE_PRECISION=f32 python -m pytest tests/test_hip_model_sp.py tests/test_hip_lf.py tests/test_hip_drivers.py -m gpu -q --maxfail=20 -k "not trajectory" > gpurun_out/r21/pytest_f32.log 2>&1; tail -3 gpurun_out/r21/pytest_f32.log
EGAZE_STREAMS=0 python -m pytest tests/test_hip_model_sp.py tests/test_hip_drivers.py -m gpu -q --maxfail=20 -k "not trajectory" > gpurun_out/r21/pytest_streams0.log 2>&1; tail -3 gpurun_out/r21/pytest_streams0.log
EGAZE_GRAD_SPLIT=bf16 python -m pytest tests/test_hip_model_sp.py -m gpu -q --maxfail=20 -k "not trajectory" > gpurun_out/r21/pytest_bf16.log 2>&1; tail -3 gpurun_out/r21/pytest_bf16.log
EGAZE_FWD_SCALE=0 python -m pytest tests/test_hip_model_sp.py tests/test_hip_lf.py -m gpu -q --maxfail=20 -k "not trajectory" > gpurun_out/r21/pytest_noscale.log 2>&1; tail -3 gpurun_out/r21/pytest_noscale.log
