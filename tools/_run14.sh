cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r14
BENCH="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-leg --no-roofline"
rm -rf /tmp/ks1
rocprofv3 --kernel-trace --output-format csv -d /tmp/ks1 -o p -- $BENCH > $R/gpurun_out/r14/bench.json 2>/dev/null
python $R/tools/trace_mfma.py /tmp/ks1 > $R/gpurun_out/r14/exposed.txt 2>&1
python $R/tools/prof_summary.py /tmp/ks1 $R/gpurun_out/r14/kernel_stats.txt "bench" > /dev/null
cat $R/gpurun_out/r14/exposed.txt | head -40
