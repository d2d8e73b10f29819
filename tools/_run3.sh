cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
bash tools/ab_conv.sh "upsd0" --dtype 1 --what dgrad --only dec --iters 20 > gpurun_out/r3/ab_upsd.log 2>&1
bash tools/ab_conv.sh "upsd0" --dtype 1 --what dgrad --only dec --iters 20 >> gpurun_out/r3/ab_upsd.log 2>&1
python -m pytest tests/test_hip_ops.py -m gpu -q -k "ups_dgrad or upsample" > gpurun_out/r3/pytest_ups.log 2>&1
python -m pytest tests/test_hip_model_sp.py -m gpu -q -s -k "trajectory" > gpurun_out/r3/traj.log 2>&1
cat gpurun_out/r3/ab_upsd.log; tail -3 gpurun_out/r3/pytest_ups.log; tail -3 gpurun_out/r3/traj.log
