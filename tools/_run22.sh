cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r22
python -m pytest tests/test_hip_ops.py tests/test_hip_lf.py tests/test_hip_model_sp.py -m gpu -q -k "bn_relu or lf or LF or train_step or narrow or determin" > gpurun_out/r22/pytest.log 2>&1; tail -3 gpurun_out/r22/pytest.log
python tools/bench_lf.py 2>&1 | grep -v amdgpu
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-180
