"""cProfile of the host side of the SP training step at a small batch (where the step is host-bound)."""
import cProfile, os, pstats, sys
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
b = synthetic.sp_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 4, 224, dev, seed=100)
def step():
    out = model(b["image"], b["flow"])
    loss = crit(out, b["gt"].view(out.size()))
    loss.backward()
    opt.step()
    opt.zero_grad()
opt.zero_grad()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumtime").print_stats(45)
