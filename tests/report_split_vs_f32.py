"""Debug report: where do the split-half and exact-f32 HIP modes diverge in one small SP step (intermediate grads)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_hip_model_sp as T
import egaze_amd.hipops as H
from egaze_amd.functions import FusionBlock
from egaze_amd.floss import floss
size = int(sys.argv[1]) if len(sys.argv) > 1 else 32
res = {}
for mode in ("f32", "split"):
    H.PRECISION = mode
    model, sd0 = T.build_model()
    x_s, x_t, gt, _ = T.synth.synth_sp_batch(3, size, seed=5)
    model.train()
    xt = model.features_t(x_t.to(T.DEV)); xs = model.features_s(x_s.to(T.DEV))
    xt.retain_grad(); xs.retain_grad()
    bn = model.bn
    fused = FusionBlock.apply(xs, xt, model.fusion.weight, model.fusion.bias, bn.weight, bn.bias, bn.running_mean,
                              bn.running_var, True, float(bn.momentum), float(bn.eps))
    fused.retain_grad()
    out = model.decoder(fused, fuse_sigmoid=True)
    floss()(out, gt.to(T.DEV).view(out.size())).backward()
    torch.cuda.synchronize()
    res[mode] = {"xs": xs.detach().double().cpu(), "xt": xt.detach().double().cpu(), "fused": fused.detach().double().cpu(),
                 "d_fused": fused.grad.double().cpu(), "d_xs": xs.grad.double().cpu(), "d_xt": xt.grad.double().cpu(),
                 "out": out.detach().double().cpu()}
    for k, p in model.named_parameters():
        res[mode]["g/" + k] = p.grad.detach().double().cpu()
for k in res["f32"]:
    a, b = res["f32"][k], res["split"][k]
    if k.startswith("g/") and not (k.endswith("0.weight") or ".40." in k or ".41." in k or ".37." in k or ".38." in k): continue
    if a.abs().max().item() < 1e-9: continue
    print("%-8s max rel diff %.2e   (max|.| %.2e)" % (k, (a - b).abs().max().item() / a.abs().max().item(), a.abs().max().item()))
