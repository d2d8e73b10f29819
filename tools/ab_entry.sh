#!/bin/bash
# A/B of library build variants on the bench's per-entry-point kernel time.  Usage: bash tools/ab_entry.sh <entry substring> "<env>" ...
E=$1; shift
for v in "$@"; do
  echo "=== $v"
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg 2>/dev/null | E=$E python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=d['kernel_ms_breakdown']
print('frames/s %.1f  ms/step %.2f ' % (d['value'], d['ms_per_step']), {k: v for k, v in b.items() if os.environ['E'] in k})"
done
