#!/bin/bash
# A/B of environment settings on the full default bench legs that matter (no CPU baseline, no f32 leg).
# Usage: bash tools/ab_env.sh "<env settings>" ...
for v in "$@"; do
  echo "=== $v"
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('frames/s %.1f  ms/step %.2f  at %.2f ms  pcie %s' % (d['value'], d['ms_per_step'], d['extra']['at_ms_per_step'], {k: round(v, 2) for k, v in (d['extra']['pcie_inclusive_ms_per_step'] or {}).items()}))"
done
