cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r1
python -m pytest tests -m gpu -q --maxfail=40 -x -k "not trajectory" > gpurun_out/r1/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r1/pytest.log
python -m pytest tests/test_hip_model_sp.py -m gpu -q -s -k "trajectory" > gpurun_out/r1/traj.log 2>&1
python bench.py --steps 10 --warmup 3 > gpurun_out/r1/bench.json 2> gpurun_out/r1/bench.err
EGAZE_FWD_SCALE=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r1/bench_noscale.json 2> gpurun_out/r1/bench_noscale.err
python tests/report_grad_seeds.py 0 12 > gpurun_out/r1/seeds.log 2>&1
python tools/cpu_baseline_b32.py 16 > gpurun_out/r1/cpu_b32.log 2>&1
tail -5 gpurun_out/r1/pytest.log
