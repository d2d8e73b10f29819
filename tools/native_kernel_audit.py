"""Which stock-torch (at::native) kernels does a training step still launch, and from where?  Runs SP, AT (T = 16, B = 32) and LF
steps under torch.profiler with Python stacks and prints, per aten op that launched a device kernel, the count per step and the
input shapes and, where the profiler has it, the innermost frame inside this package.  Usage: python tools/native_kernel_audit.py"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import egaze_amd  # noqa: E402,F401
from egaze_amd import synthetic  # noqa: E402
from egaze_amd.floss import floss  # noqa: E402
from egaze_amd.functions import MSELoss  # noqa: E402
from egaze_amd.models.LSTMnet import lstmnet  # noqa: E402
from egaze_amd.models.late_fusion import late_fusion  # noqa: E402
from egaze_amd.models.model_SP import model_SP  # noqa: E402
from egaze_amd.optim import FusedAdam  # noqa: E402
from egaze_amd.utils import cfg, computeAAEAUC, make_layers  # noqa: E402

dev = torch.device("cuda:0")
B = 8
sp = model_SP(make_layers(cfg['D'], 3), make_layers(cfg['D'], 20)).to(dev).train()
crit = floss().to(dev)
osp = FusedAdam(sp.parameters(), lr=1e-7)
bsp = synthetic.sp_batch(B, 224, dev, seed=1)
lstm = lstmnet().to(dev).train()
oat = FusedAdam(lstm.parameters(), lr=1e-4)
atb = synthetic.at_batch(16, 32, dev, seed=2)
at_in, at_tgt = atb["input"], torch.tanh(atb["gt"])
h0, c0 = torch.zeros(2, 32, 512, device=dev), torch.zeros(2, 32, 512, device=dev)
lf = late_fusion().to(dev).train()
olf = FusedAdam(lf.parameters(), lr=1e-4)
im, feat, gt = (torch.rand(B, 1, 224, 224, device=dev) for _ in range(3))


def sp_step():
    out = sp(bsp["image"], bsp["flow"])
    crit(out, bsp["gt"].view(out.size())).backward()
    osp.step(); osp.zero_grad()


def at_step():
    pred, _ = lstm(at_in, (h0, c0))
    MSELoss.apply(pred, at_tgt).backward()
    oat.step(); oat.zero_grad()


def lf_step():
    out = lf(feat, im)
    loss = crit(out, gt)
    computeAAEAUC(out.detach(), gt)
    olf.zero_grad(); loss.backward(); olf.step()


for name, fn in (("SP step (SP.trainSP body)", sp_step), ("AT step (T = 16, B = 32)", at_step), ("LF iteration (LF.trainLate body, eager)", lf_step)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    N = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for _ in range(N):
            fn()
        torch.cuda.synchronize()
    ops = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and any(
                k.name.startswith("void at::native") or "at::native" in k.name for k in ev.kernels):
            frame = next((f for f in ev.stack if "egocentric-gaze-prediction_amd" in f or "egaze_amd" in f), ev.stack[0] if ev.stack else "?")
            ops[(ev.name, str(ev.input_shapes)[:60] + "  " + frame.split("egocentric-gaze-prediction_amd/")[-1][:70])] += 1
    print(f"--- {name}: stock-torch device kernels per step")
    if not ops:
        print("    none")
    for (op, frame), n in sorted(ops.items(), key=lambda kv: -kv[1]):
        print(f"    {n / N:5.1f} x {op:28s} {frame}")
