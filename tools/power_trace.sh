#!/bin/bash
# Samples rocm-smi (socket power, clocks, power cap) every 0.5 s while `python bench.py --steps 200` runs.
# Usage (GPU box): bash tools/power_trace.sh <out.txt>
OUT=$1
rocm-smi --showmaxpower --showpower --showclocks > $OUT.idle 2>&1
( python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline > $OUT.bench 2>/dev/null ) &
BP=$!
sleep 12
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' ' >> $OUT
  echo >> $OUT
  sleep 0.5
done
wait $BP
cat $OUT.bench | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench:', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')" >> $OUT
