cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r10
python tools/bench_at_loop.py --n 2000 > gpurun_out/r10/at_loop.log 2>&1; tail -1 gpurun_out/r10/at_loop.log
EGAZE_AT_GRAPH=0 python tools/bench_at_loop.py --n 2000 >> gpurun_out/r10/at_loop.log 2>&1; tail -1 gpurun_out/r10/at_loop.log
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r10/prof2 -- python $GRAFT_REPO_ROOT/tools/bench_at_loop.py --n 200 > /dev/null 2>&1
