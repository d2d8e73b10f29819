cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r9
python -m pytest tests/test_hip_lf.py -m gpu -q > gpurun_out/r9/pytest.log 2>&1; tail -3 gpurun_out/r9/pytest.log
python tools/bench_config1.py > gpurun_out/r9/config1.log 2>&1; cat gpurun_out/r9/config1.log
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r9/prof -- python $GRAFT_REPO_ROOT/tools/bench_config1.py --frames 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r9/prof -name "*stats*" | head
