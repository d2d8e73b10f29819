"""How long does the HOST need to issue one SP training step (all launches, no synchronisation) vs the GPU time of the step?
If the two are close the step is launch-bound and a captured hipGraph would pay."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egaze_amd  # noqa
from egaze_amd.models.model_SP import model_SP
from egaze_amd.utils import make_layers, cfg
from egaze_amd.floss import floss
from egaze_amd.optim import FusedAdam
from egaze_amd import synthetic

dev = torch.device("cuda", 0)
model = model_SP(make_layers(cfg['D'], 3), make_layers(cfg['D'], 20)).to(dev).train()
crit = floss().to(dev)
opt = FusedAdam(model.parameters(), lr=1e-7)
b = synthetic.sp_batch(32, 224, dev, seed=100)
def step():
    out = model(b["image"], b["flow"])
    loss = crit(out, b["gt"].view(out.size()))
    loss.backward()
    opt.step()
    opt.zero_grad()
opt.zero_grad()
for _ in range(3):
    step()
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(10):
    h0 = time.perf_counter()
    step()
    host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 10
print("host issue per step: first %.2f ms, min %.2f ms, median %.2f ms; wall per step %.2f ms" %
      (host[0] * 1e3, min(host) * 1e3, sorted(host)[5] * 1e3, tot * 1e3))
