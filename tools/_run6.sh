cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
V=$PWD/egocentric-gaze-prediction_amd/csrc/variants/libegaze_hip_wave0.so
bash tools/ab_conv.sh "wave0" --dtype 1 --what fwd,dgrad --iters 20 > gpurun_out/r6/ab_conv.log 2>&1
bash tools/ab_bench.sh "EGAZE_X=1" "EGAZE_HIP_LIB=$V" "EGAZE_X=1" "EGAZE_HIP_LIB=$V" > gpurun_out/r6/ab_bench.log 2>&1
python -m pytest tests/test_hip_ops.py -m gpu -q -k "streamed or split" > gpurun_out/r6/pytest.log 2>&1
grep -v "^enc\|^dec" gpurun_out/r6/ab_conv.log; cat gpurun_out/r6/ab_bench.log; tail -3 gpurun_out/r6/pytest.log
