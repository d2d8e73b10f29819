cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r20
timeout 600 python -m pytest tests/test_hip_model_sp.py -m gpu -q -k "overlapped or train_step or determin" > gpurun_out/r20/pytest.log 2>&1; tail -15 gpurun_out/r20/pytest.log
bash tools/ab_bench.sh "EGAZE_OVERLAP_ADAM=1" "EGAZE_OVERLAP_ADAM=0" "EGAZE_OVERLAP_ADAM=1" "EGAZE_OVERLAP_ADAM=0" 2>&1 | grep -v "^{"
