cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r12
V=$PWD/egocentric-gaze-prediction_amd/csrc/variants/libegaze_hip_fine0.so
bash tools/ab_bench.sh "EGAZE_X=1" "EGAZE_HIP_LIB=$V" "EGAZE_X=1" "EGAZE_HIP_LIB=$V" > gpurun_out/r12/ab_bench.log 2>&1
grep -v "^{" gpurun_out/r12/ab_bench.log
