cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r25
for k in "A=0" "EGAZE_STREAMS=0" "EGZ_FIRST_DIRECT=0" "EGAZE_FIRST_FUSE=0"; do
echo "=== $k"; env $k timeout 300 python tools/bench_lf.py --steps 60 2>&1 | grep "metric=off"
done
echo "=== serial timeline"
rm -rf /tmp/lfprof
(cd /tmp && EGAZE_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lfprof -o lf -- python $GRAFT_REPO_ROOT/tools/bench_lf.py --steps 10 > /dev/null 2>&1)
python tools/lf_timeline.py /tmp/lfprof > gpurun_out/r25/timeline_serial.txt 2>&1; cat gpurun_out/r25/timeline_serial.txt | awk '{print $3, $6, $7}' | head -60
