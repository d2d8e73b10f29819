cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r7
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg > gpurun_out/r7/bench.json 2> gpurun_out/r7/bench.err
tail -20 gpurun_out/r7/bench.err
python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r7/pytest.log 2>&1
tail -5 gpurun_out/r7/pytest.log
