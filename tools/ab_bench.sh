#!/bin/bash
# A/B of runtime knobs on the headline bench (no CPU baseline / f32 leg).  Usage: bash tools/ab_bench.sh "<env settings>" ...
for v in "$@"; do
  echo "=== $v"
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg 2>>gpurun_out/ab_bench.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('frames/s %.1f  ms/step %.2f  igemm avg %.1f us  loss %.6f' % (d['value'], d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms'], d['loss']))
b=d['kernel_ms_breakdown']; print({k:b[k] for k in list(b)[:8]}, b['_sum_kernel_ms'])"
done
