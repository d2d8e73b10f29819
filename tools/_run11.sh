cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r11
V=$PWD/egocentric-gaze-prediction_amd/csrc/variants/libegaze_hip_fineall.so
EGAZE_HIP_LIB=$V timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -k "streamed or split or tile" > gpurun_out/r11/pytest_fineall.log 2>&1; tail -4 gpurun_out/r11/pytest_fineall.log
echo "=== default"; timeout 200 python tools/bench_conv.py --dtype 1 --what fwd,dgrad --iters 20 2>&1 | grep -v amdgpu | grep TOTAL
echo "=== fineall (default tiles, fine interleave)"; EGAZE_HIP_LIB=$V timeout 200 python tools/bench_conv.py --dtype 1 --what fwd,dgrad --iters 20 2>&1 | grep -v amdgpu
echo "=== default"; timeout 200 python tools/bench_conv.py --dtype 1 --what fwd,dgrad --iters 20 2>&1 | grep -v amdgpu | grep TOTAL
