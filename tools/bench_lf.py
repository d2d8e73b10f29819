"""Late-fusion (LF.trainLate body, LF.py:83-100) step timing: late_fusion fwd + floss + device AAE/AUC + bwd + Adam at
B frames of 224 x 224 (BASELINE config 5's last stage).  Usage: python tools/bench_lf.py [--batch 32] [--steps 20]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egaze_amd  # noqa
from egaze_amd.models.late_fusion import late_fusion
from egaze_amd.floss import floss
from egaze_amd.optim import FusedAdam
from egaze_amd.utils import computeAAEAUC
from egaze_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
model = late_fusion().to(dev); model.train()
crit = floss().to(dev)
opt = FusedAdam(model.parameters(), lr=1e-4)
b = synthetic.lf_batch(a.batch, 224, dev, seed=3) if hasattr(synthetic, "lf_batch") else None
if b is None:
    im, feat, gt = (torch.rand(a.batch, 1, 224, 224, device=dev) for _ in range(3))
else:
    im, feat, gt = b["im"], b["feat"], b["gt"]
if os.environ.get("EGAZE_MAIN_PRIO"):       # experiment: the layer chain on a high-priority stream (helper streams stay normal)
    _hp = torch.cuda.Stream(priority=int(os.environ["EGAZE_MAIN_PRIO"]))
    _hp.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(_hp)
    print("main stream priority", _hp.priority, "range", torch.cuda.Stream.priority_range())
def step(metric):
    out = model(feat, im)
    loss = crit(out, gt)
    if metric:
        computeAAEAUC(out.detach(), gt)
    opt.zero_grad(); loss.backward(); opt.step()
for metric in (False, True):
    for _ in range(3): step(metric)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps): step(metric)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    print(f"LF step B={a.batch} metric={'device AAE/AUC' if metric else 'off'}: {dt*1e3:.2f} ms  {a.batch/dt:.0f} frames/s")

# the same step captured into one hipGraph and replayed (graphs.GraphedTrainStep; what LF._run does by default)
from egaze_amd.graphs import GraphedTrainStep
def fwd_loss(feat_, im_, gt_):
    o = model(feat_, im_)
    return crit(o, gt_), o
g = GraphedTrainStep(fwd_loss, opt, (feat, im, gt))
for metric in (False, True):
    for _ in range(4): 
        l, o = g(feat, im, gt)
        if metric: computeAAEAUC(o, gt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        l, o = g(feat, im, gt)
        if metric: computeAAEAUC(o, gt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    print(f"LF step B={a.batch} metric={'device AAE/AUC' if metric else 'off'} hipGraph replay: {dt*1e3:.2f} ms  {a.batch/dt:.0f} frames/s")
g.close()

# LF.trainLate's whole iteration as LF._run issues it (LF.GraphedLateIteration): the batch metric inside the captured step, loss / AAE /
# AUC read back 16 iterations at a time
from egaze_amd.LF import GraphedLateIteration
it = GraphedLateIteration(model, crit, opt, (feat, im, gt))
for _ in range(4): it(feat, im, gt)
it.drain()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    it(feat, im, gt)
    if it.full: it.drain()
it.drain()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(f"LF iteration B={a.batch} (forward + floss + computeAAEAUC + backward + Adam, one hipGraph replay, deferred read-back): {dt*1e3:.2f} ms  {a.batch/dt:.0f} frames/s")
it.close()
