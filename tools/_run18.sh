cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r18
V=$PWD/egocentric-gaze-prediction_amd/csrc/variants/libegaze_hip_prev.so
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -k "wgrad or backward or absmax or scaling or first_conv or fuzz" 2>&1 | tail -2
echo "=== new"; timeout 200 python tools/bench_conv.py --dtype 1 --what wgrad --iters 30 2>&1 | grep -v amdgpu
echo "=== prev"; EGAZE_HIP_LIB=$V timeout 200 python tools/bench_conv.py --dtype 1 --what wgrad --iters 30 2>&1 | grep -v amdgpu
