#!/bin/bash
# small-batch training step (the reference's default --batch_size_sp is 8) with / without the split-K launches
for b in 8 4; do for v in "EGAZE_NOOP=0"; do   # (the split-K switch is a module constant now: hipops.SPLITK)
  echo "=== batch $b $v"
  env $v python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('frames/s %.1f  ms/step %.2f' % (d['value'], d['ms_per_step']))"
done; done
