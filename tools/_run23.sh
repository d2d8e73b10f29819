cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r23
rm -rf /tmp/lft
EGAZE_LF_BENCH_EAGER_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/lft -o p -- python $R/tools/bench_lf.py --steps 10 > /dev/null 2>&1
python $R/tools/lf_timeline.py /tmp/lft 8 > $R/gpurun_out/r23/lf_timeline.txt 2>&1
tail -80 $R/gpurun_out/r23/lf_timeline.txt
