cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r17
for m in 1 0; do
echo "=== timeline BNSUMS_FUSE=$m"
rm -rf /tmp/lfprof
(cd /tmp && EGAZE_BNSUMS_FUSE=$m timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lfprof -o lf -- python $GRAFT_REPO_ROOT/tools/bench_lf.py --steps 10 > /dev/null 2>&1)
python tools/lf_timeline.py /tmp/lfprof > gpurun_out/r17/timeline_$m.txt 2>&1; tail -64 gpurun_out/r17/timeline_$m.txt
done
