#!/bin/bash
# GPU busy fraction / idle gaps of the LF step and the host time to issue it
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lfg
rocprofv3 --kernel-trace --output-format csv -d /tmp/lfg -o p -- python $R/tools/bench_lf.py --steps 30 > /dev/null 2>&1
python $R/tools/trace_gaps.py /tmp/lfg 0.4
python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import egaze_amd
from egaze_amd.models.late_fusion import late_fusion
from egaze_amd.floss import floss
from egaze_amd.optim import FusedAdam
dev = torch.device("cuda", 0)
model = late_fusion().to(dev); model.train(); crit = floss().to(dev); opt = FusedAdam(model.parameters(), lr=1e-4)
im, feat, gt = (torch.rand(32, 1, 224, 224, device=dev) for _ in range(3))
def step():
    out = model(feat, im); loss = crit(out, gt); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"LF host issue {(t1-t0)/50*1e3:.2f} ms per step, total {(t2-t0)/50*1e3:.2f} ms per step")
PY
