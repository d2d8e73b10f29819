cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r13
V=$PWD/egocentric-gaze-prediction_amd/csrc/variants/libegaze_hip_fine0.so
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -k "wgrad or backward or absmax or scaling or first_conv or fuzz" > gpurun_out/r13/pytest.log 2>&1; tail -4 gpurun_out/r13/pytest.log
echo "=== new"; timeout 200 python tools/bench_conv.py --dtype 1 --what wgrad --iters 20 2>&1 | grep -v amdgpu
echo "=== old wgrad (fine0 variant lib)"; EGAZE_HIP_LIB=$V timeout 200 python tools/bench_conv.py --dtype 1 --what wgrad --iters 20 2>&1 | grep -v amdgpu
