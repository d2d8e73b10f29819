cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r19
python -m pytest tests -m gpu -q --maxfail=20 > gpurun_out/r19/pytest.log 2>&1; tail -4 gpurun_out/r19/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r19/smoke.log 2>&1; tail -2 gpurun_out/r19/smoke.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r19/bench.json 2>/dev/null; cat gpurun_out/r19/bench.json | cut -c1-200
