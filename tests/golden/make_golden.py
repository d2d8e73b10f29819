#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by importing the REAL reference.

Runs only in the build container (needs /root/reference, read-only).  Nothing from the
reference is copied: the fixtures are inputs' seeds + the reference's numerical outputs.
Weights / inputs are regenerated on the consumer side from oracle/synth.py with the same
seeds, so only results are stored.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Reference entry points exercised (file:line in /root/reference):
  models/model_SP.py:5-65   model_SP            utils.py:64-76   make_layers / cfg
  floss.py:5-41             floss               SP.py:126-138    train-step replay
  models/LSTMnet.py:15-37   lstmnet             AT.py:118-147    trainLSTM replay
  models/late_fusion.py     late_fusion         utils.py:96-140  computeAAEAUC
  utils.py:78-94            change_key_names    AT.py:25-66      crop_feature / get_weighted
"""
import os
import sys
import types
import collections

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

# cv2 / skimage are absent in this image and only used for image I/O by the reference.
for name in ("cv2", "skimage", "skimage.io", "skimage.transform"):
    sys.modules.setdefault(name, types.ModuleType(name))

from oracle import synth  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def shapes_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def load_synth(module, seed, head_gain=1.0):
    sd = synth.synth_state_dict(shapes_of(module), seed=seed, head_gain=head_gain)
    module.load_state_dict(sd)
    return sd


def grad_summary(named_params):
    out = {}
    for k, p in named_params:
        g = p.grad.detach().double()
        out[k] = np.array([g.norm().item(), g.sum().item(), g.abs().max().item()])
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_model_sp(size, batch, tag, head_gain):
    from models.model_SP import model_SP
    from utils import make_layers, cfg
    from floss import floss

    model = model_SP(make_layers(cfg["D"], 3), make_layers(cfg["D"], 20))
    load_synth(model, seed=1, head_gain=head_gain)
    x_s, x_t, gt, _ = synth.synth_sp_batch(batch, size, seed=0)
    arrs = {}

    # ---- eval-mode forward (running statistics)
    model.eval()
    feats = []
    h = model._modules.get("features_s").register_forward_hook(lambda m, i, o: feats.append(o))
    with torch.no_grad():
        out_eval = model(x_s, x_t)
    h.remove()
    arrs["eval_out"] = out_eval.numpy()
    arrs["eval_features_s_sum"] = feats[0].double().sum(dim=(2, 3)).numpy()     # (B,512)
    arrs["eval_features_s_b0c0"] = feats[0][0, 0].numpy()

    # ---- one literal SP.trainSP iteration (SP.py:126-138), train-mode BN, floss, Adam
    lr = 1e-4
    model.train()
    criterion = floss()
    optimizer = torch.optim.Adam(model.parameters(), lr=lr)
    optimizer.zero_grad()
    output = model(x_s, x_t)
    target = gt.view(output.size())
    loss = criterion(output, target)
    loss.backward()
    gs = grad_summary(model.named_parameters())
    full_grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()
                  if p.numel() <= 512 or k in ("decoder.28.weight",)}
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    optimizer.step()
    optimizer.zero_grad()
    arrs["train_out"] = output.detach().numpy()
    arrs["train_loss"] = np.array(loss.item())
    arrs["lr"] = np.array(lr)
    for k, v in gs.items():
        arrs["gsum/" + k] = v
    for k, v in full_grads.items():
        arrs["grad/" + k] = v.numpy()
    sd_after = model.state_dict()
    for k, v in sd_after.items():
        if "running_" in k and v.numel() <= 64:
            arrs["after/" + k] = v.numpy()
        elif "running_" in k:
            arrs["after_sum/" + k] = np.array([v.double().sum().item(), v.double().norm().item()])
    for k, p in model.named_parameters():
        d = (p.detach() - before[k]).double()
        arrs["delta/" + k] = np.array([d.sum().item(), d.abs().max().item()])
    print(tag, "eval out range", out_eval.min().item(), out_eval.max().item(),
          "train out range", output.min().item(), output.max().item(), "loss", loss.item())
    save(f"model_sp_{tag}.npz", **arrs)


def gen_floss():
    from floss import floss
    rs = np.random.RandomState(5)
    size = 224
    gt = synth.synth_gt(3, size, rs)                       # quantised gaussians -> plateau ties
    single = np.zeros((1, 1, size, size), np.float32)
    single[0, 0, 37, 181] = 1.0                             # single peak
    flat = np.full((1, 1, size, size), 0.25, np.float32)    # all equal -> centroid = centre
    two = np.zeros((1, 1, size, size), np.float32)
    two[0, 0, 10, 20] = 0.5
    two[0, 0, 200, 101] = 0.5                               # two tied maxima far apart
    target = np.concatenate([gt, single, flat, two], 0)
    x = rs.uniform(0.02, 0.98, target.shape).astype(np.float32)
    x[0, 0, 0, :4] = [0.0, 1.0, 1e-30, 1 - 1e-7]            # exercise the -100 log clamp
    crit = floss()
    w = crit.build_weight_from_target(torch.from_numpy(target))
    xin = torch.from_numpy(x).requires_grad_(True)
    loss = crit(xin, torch.from_numpy(target))
    loss.backward()
    n_ties = [(target[b] == target[b].max()).sum() for b in range(target.shape[0])]
    print("floss ties per sample", n_ties, "loss", loss.item())
    # the 224 gradient map is large: keep two samples fully + sums of the rest
    save("floss.npz", target_seed=np.array(5), x=x[[0, 3]], weights_rows=w[:, 0, ::37, :],
         weights_sum=w.astype(np.float64).sum(axis=(1, 2, 3)), loss=np.array(loss.item()),
         grad_b0=xin.grad[0, 0].numpy(), grad_b3=xin.grad[3, 0].numpy(),
         grad_sum=xin.grad.double().sum(dim=(1, 2, 3)).numpy(),
         grad_abs_sum=xin.grad.double().abs().sum(dim=(1, 2, 3)).numpy())


def gen_lstm():
    from models.LSTMnet import lstmnet
    from utils import repackage_hidden
    net = lstmnet()
    load_synth(net, seed=2)
    arrs = {}
    # T=3, B=2 with an explicit zero hidden + backward
    inp, tgt = synth.synth_at_batch(3, 2, seed=3)
    h0 = torch.zeros(2, 2, 512)
    c0 = torch.zeros(2, 2, 512)
    out, (hn, cn) = net(inp, (h0, c0))
    loss = torch.nn.MSELoss()(out, torch.tanh(tgt))
    loss.backward()
    arrs["t3b2_out"] = out.detach().numpy()
    arrs["t3b2_hn"] = hn.detach().numpy()
    arrs["t3b2_cn"] = cn.detach().numpy()
    arrs["t3b2_loss"] = np.array(loss.item())
    for k, v in grad_summary(net.named_parameters()).items():
        arrs["t3b2_gsum/" + k] = v
    arrs["t3b2_grad/lin.bias"] = net.lin.bias.grad.numpy().copy()
    arrs["t3b2_grad/lstm.bias_ih_l1"] = net.lstm.bias_ih_l1.grad.numpy().copy()
    # T=1, B=1 with hidden=None
    net.zero_grad()
    inp1, _ = synth.synth_at_batch(1, 1, seed=4)
    with torch.no_grad():
        out1, (h1, c1) = net(inp1, None)
    arrs["t1b1_out"] = out1.numpy()
    arrs["t1b1_hn"] = h1.numpy()
    # B>1 with hidden=None raises in the reference
    try:
        net(inp, None)
        arrs["b2_none_raises"] = np.array(0)
    except RuntimeError:
        arrs["b2_none_raises"] = np.array(1)
    # AT.trainLSTM replay (AT.py:118-147) over 5 samples, video break at sample 3
    net = lstmnet()
    load_synth(net, seed=2)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    crit = torch.nn.MSELoss()
    ins, tgts = synth.synth_at_batch(5, 1, seed=6)
    same = [1, 1, 1, 0, 1]
    hidden, pred, losses = None, None, []
    tanh = torch.nn.Tanh()
    for i in range(5):
        if int(same[i]) == 0:
            hidden = None
        a = ins[i].unsqueeze(0)
        t = tgts[i].unsqueeze(0)
        if pred is not None:
            l = crit(pred, tanh(t))
            opt.zero_grad()
            l.backward()
            opt.step()
            losses.append(l.item())
        hidden = repackage_hidden(hidden)
        pred, hidden = net(a, hidden)
    arrs["replay_losses"] = np.array(losses)
    arrs["replay_final_pred"] = pred.detach().numpy()
    arrs["replay_lin_bias"] = net.lin.bias.detach().numpy()
    arrs["replay_w_sum"] = np.array([p.detach().double().sum().item() for p in net.parameters()])
    save("lstmnet.npz", **arrs)


def gen_late_fusion():
    from models.late_fusion import late_fusion
    from floss import floss
    arrs = {}
    for size, tag in ((32, "s32"), (224, "s224")):
        net = late_fusion()
        load_synth(net, seed=3, head_gain=0.5)
        im, feat, gt = synth.synth_lf_batch(2, size, seed=7)
        net.eval()
        with torch.no_grad():
            arrs[f"{tag}_eval_out"] = net(feat, im).numpy()
            arrs[f"{tag}_eval_out_swapped_sum"] = np.array(net(im, feat).double().sum().item())
        net.train()
        out = net(feat, im)                                   # LF.py:90 argument order
        loss = floss()(out, gt)
        loss.backward()
        arrs[f"{tag}_train_out"] = out.detach().numpy()
        arrs[f"{tag}_loss"] = np.array(loss.item())
        for k, p in net.named_parameters():
            arrs[f"{tag}_grad/{k}"] = p.grad.numpy().copy()
        for k, v in net.state_dict().items():
            if "running_" in k:
                arrs[f"{tag}_after/{k}"] = v.numpy().copy()
    save("late_fusion.npz", **arrs)


def gen_metrics_and_glue():
    import utils as rutils
    arrs = {}
    rs = np.random.RandomState(11)
    gt = synth.synth_gt(3, 224, rs)[:, 0]
    pred = synth.synth_gt(3, 224, rs)[:, 0] * 0.8 + rs.uniform(0, 0.05, (3, 224, 224)).astype(np.float32)
    aae, auc, gp = rutils.computeAAEAUC(pred, gt)
    arrs["batch_aae_auc"] = np.array([aae, auc])
    arrs["batch_gp"] = np.array(gp)
    aae1, auc1, gp1 = rutils.computeAAEAUC(pred[1], gt[1])
    arrs["single_aae_auc"] = np.array([aae1, auc1])
    arrs["single_gp"] = np.array(gp1)
    # uint8-scaled prediction as AT.extract_late feeds it (AT.py:230,233)
    aae2, auc2, gp2 = rutils.computeAAEAUC(np.uint8(255 * np.clip(pred[2], 0, 1)), gt[2])
    arrs["u8_aae_auc"] = np.array([aae2, auc2])
    # change_key_names on a synthetic vgg16_bn-shaped ordered dict
    od = collections.OrderedDict()
    krs = np.random.RandomState(12)
    od["features.0.weight"] = torch.from_numpy(krs.standard_normal((64, 3, 3, 3)).astype(np.float32))
    for n in range(1, 30):
        od[f"features.k{n}"] = torch.from_numpy(krs.standard_normal((4,)).astype(np.float32))
    new = rutils.change_key_names(od, 20)
    arrs["ckn_keys"] = np.array(list(new.keys()))
    arrs["ckn_w0"] = new["features.0.weight"].numpy()
    # AT glue: crop_feature + get_weighted (AT.py:25-66).  AT.py imports cleanly with the stubs.
    import AT as rat
    feat = torch.from_numpy(np.abs(krs.standard_normal((2, 512, 14, 14))).astype(np.float32))
    gps = [[5, 220], [117, 60]]
    cf = rat.crop_feature(feat, gps, 3)
    arrs["crop_feature"] = cf.numpy()
    w = cf.contiguous().view(2, 512, -1).mean(2)
    arrs["get_weighted"] = rat.get_weighted(w[0], feat[0:1]).numpy()
    save("metrics_glue.npz", **arrs)


def gen_config1():
    """BASELINE config 1: run_spatialstream.py plumbing on one synthetic 224x224x3 image, CPU.  The script parses
    argv and runs at import, so only its class / function definitions are extracted (ast) and executed."""
    import ast
    import contextlib
    import io
    import math
    import torch.nn as nn
    import utils as rutils
    from models.late_fusion import late_fusion
    from scipy import ndimage
    src = open(os.path.join(REF, "run_spatialstream.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef))]
    ns = {"nn": nn, "math": math, "np": np, "torch": torch}
    ns.update({k: getattr(rutils, k) for k in dir(rutils) if not k.startswith("_")})
    exec(compile(ast.Module(body=keep, type_ignores=[]), "run_spatialstream_defs", "exec"), ns)
    model = ns["VGG"](rutils.make_layers(rutils.cfg["D"], 3))
    load_synth(model, seed=4, head_gain=0.25)
    model.eval()
    lf = late_fusion()
    load_synth(lf, seed=3, head_gain=0.5)
    lf.eval()
    rs = np.random.RandomState(21)
    im_u8 = rs.randint(0, 256, (224, 224, 3)).astype(np.uint8)          # "cv2.imread + resize" result (BGR)
    im = ns["totensor"](im_u8)
    with torch.no_grad():
        out, feat = model(im)
    imq = ns["toim"](out)
    predicted = ndimage.center_of_mass(imq)
    with contextlib.redirect_stdout(io.StringIO()):
        vec = ns["crop_feature1"](feat, predicted, 3)
    vec = vec.contiguous().view(vec.size(0), vec.size(1), -1)
    vec = torch.mean(vec, 2).squeeze()
    weighted = ns["get_weighted"](vec, feat)
    weighted = torch.nn.functional.interpolate(weighted, scale_factor=16, mode="bilinear")   # F.upsample default
    with torch.no_grad():
        fin = lf(out, weighted)
    save("config1.npz", out=out.numpy(), feat_sum=feat.double().sum(dim=(2, 3)).numpy(), imq=imq,
         predicted=np.array(predicted), vec=vec.numpy(), weighted=weighted.numpy(), fin=fin.numpy(),
         fin_u8=ns["toim"](fin), n_params=np.array(sum(p.numel() for p in model.parameters())))
    print("config1 out range", out.min().item(), out.max().item(), "fin range", fin.min().item(), fin.max().item())


def bilinear_u8(arr, size):
    """Stand-in for cv2.resize(arr, size) (INTER_LINEAR) on a uint8 map -- cv2 is absent in the build container, so the
    reference's ONE call to it (AT.py:251, the 14x14 -> 224x224 AT map) is emulated in float with half-pixel centres and
    edge clamping, rounded to nearest.  cv2 itself works in 11-bit fixed point and can differ by 1 LSB; consumers compare
    the exact 14x14 map and treat the 224x224 one as data."""
    h, w = arr.shape
    W2, H2 = size
    ys = np.clip((np.arange(H2) + 0.5) * h / H2 - 0.5, 0, h - 1)
    xs = np.clip((np.arange(W2) + 0.5) * w / W2 - 0.5, 0, w - 1)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = arr.astype(np.float64)
    out = (a[y0][:, x0] * (1 - fy) * (1 - fx) + a[y0][:, x1] * (1 - fy) * fx
           + a[y1][:, x0] * fy * (1 - fx) + a[y1][:, x1] * fy * fx)
    return np.uint8(np.floor(out + 0.5))


def gen_extract_lstm():
    """extractLSTMw.py: crop_feature_var for every gaze cell of a 14x14 map (float clip + int() slicing, :46-58) and the
    fixation state machine of extractw (:60-112) driven through the reference's own function with a fake loader/model."""
    import tempfile
    import extractLSTMw as rex
    arrs = {}
    rs = np.random.RandomState(51)
    feat = torch.from_numpy(np.abs(rs.standard_normal((1, 6, 14, 14))).astype(np.float32))
    for size in (2, 3, 4):
        means, shapes = [], []
        for ind in range(196):
            c = rex.crop_feature_var(feat, torch.tensor([[ind]]), size).contiguous()
            shapes.append(c.shape[2:])
            means.append(c.view(1, 6, -1).mean(2).numpy()[0])
        arrs[f"var_mean_s{size}"] = np.array(means)
        arrs[f"var_shape_s{size}"] = np.array(shapes)
    # state machine: which frames get extracted, and what is stored for them (AvgPool2d(16) arg-max cell)
    fix = [0, 1, 1, 1, 0, 0, 1, 1, 0, 1, 1, 1, 1, 0]
    gts = synth.synth_gt(len(fix), 224, np.random.RandomState(52))
    ims = rs.standard_normal((len(fix), 3, 224, 224)).astype(np.float32)
    loader = [{"fixsac": torch.tensor([[float(f)]]), "imname": ["vid_%05d.jpg" % i],
               "image": torch.from_numpy(ims[i:i + 1]), "gt": torch.from_numpy(gts[i:i + 1])} for i, f in enumerate(fix)]

    class Fake(torch.nn.Module):            # a cheap deterministic stand-in for features_s: (1,3,224,224) -> (1,512,14,14)
        def forward(self, x):
            p = torch.nn.functional.avg_pool2d(x, 16)                                  # (1,3,14,14)
            k = torch.arange(512, dtype=torch.float32).view(1, 512, 1, 1)
            return torch.relu(p[:, 0:1] * torch.sin(k * 0.37) + p[:, 1:2] * torch.cos(k * 0.11) + p[:, 2:3] * 0.5)

    with tempfile.TemporaryDirectory() as d:
        rex.extractw(loader, Fake(), d, crop_size=3, device="cpu", align=False)
        names = sorted(os.listdir(d))
        arrs["extractw_names"] = np.array(names)
        arrs["extractw_vecs"] = np.stack([torch.load(os.path.join(d, n)).numpy() for n in names])
    arrs["extractw_fix"] = np.array(fix)
    save("extract_lstm.npz", **arrs)


def gen_config5():
    """BASELINE config 5, the hand-over between the stages: AT.extract_late (AT.py:199-253) on three seeded frames --
    SP gaze map -> uint8; hooked features_s; GT gaze point; 3x3 crop mean; LSTM branch on saccade frames with the hidden
    state carried over; channel-weighted map -> uint8 -> resize -- followed by ONE LF.trainLate step (LF.py:83-100) on what
    was extracted.  AT.__init__ / LF.__init__ need CUDA and dataset folders, so the objects are assembled with __new__ and
    the reference METHODS run unmodified.  cv2 is stubbed: imwrite captures the arrays, resize = bilinear_u8 above."""
    import tempfile
    import AT as rat
    import utils as rutils
    from models.model_SP import model_SP
    from models.LSTMnet import lstmnet
    from models.late_fusion import late_fusion
    from floss import floss
    written = {}
    resized_in = []
    cv2 = sys.modules["cv2"]
    cv2.imwrite = lambda path, arr: written.__setitem__(path, np.array(arr, copy=True))
    def _resize(arr, size):
        resized_in.append(np.array(arr, copy=True))
        return bilinear_u8(arr, size)
    cv2.resize = _resize
    rat.cv2 = cv2
    at = rat.AT.__new__(rat.AT)
    at.device = torch.device("cpu")
    at.align, at.crop_size = False, 3
    at.model = model_SP(rutils.make_layers(rutils.cfg["D"], 3), rutils.make_layers(rutils.cfg["D"], 20))
    load_synth(at.model, seed=1, head_gain=0.25)
    at.model._modules.get(rat.hook_name).register_forward_hook(rat.hook_feature)
    at.lstm = lstmnet()
    load_synth(at.lstm, seed=2)
    n = 3
    x_s, x_t, gt, _ = synth.synth_sp_batch(n, 224, seed=31)
    fixsac = [1.0, 0.0, 0.0]              # a fixation frame, then two saccade frames (LSTM branch, hidden carried over)
    loader = [{"imname": ["f%d.png" % i], "fixsac": torch.tensor([[fixsac[i]]]), "image": x_s[i:i + 1],
               "flow": x_t[i:i + 1], "gt": gt[i:i + 1]} for i in range(n)]
    with tempfile.TemporaryDirectory() as d:
        pred_dir, feat_dir = os.path.join(d, "pred") + "/", os.path.join(d, "feat") + "/"
        at.extract_late(loader, pred_dir, feat_dir)
        pred = np.stack([written[os.path.join(pred_dir, "f%d.png" % i)] for i in range(n)])
        feat = np.stack([written[os.path.join(feat_dir, "f%d.png" % i)] for i in range(n)])
    feat14 = np.stack(resized_in)
    assert pred.dtype == np.uint8 and feat.dtype == np.uint8 and feat14.shape == (n, 14, 14)
    # one LF.trainLate step on the extracted maps (tensors as data/lateDataset.py:22-33 builds them)
    lf = late_fusion()
    load_synth(lf, seed=3, head_gain=0.5)
    crit = floss()
    opt = torch.optim.Adam(lf.parameters(), lr=1e-4)
    im = torch.from_numpy(pred).float().div(255).unsqueeze(1)
    ft = torch.from_numpy(feat).float().div(255).unsqueeze(1)
    gtq = torch.from_numpy(np.uint8(np.round(gt.numpy() * 255))).float().div(255)       # the gt as an 8-bit image
    out = lf(ft, im)                                                                     # LF.py:90 argument order
    loss = crit(out, gtq)
    aae, auc, _ = rutils.computeAAEAUC(out.detach().numpy().squeeze(), gtq.numpy().squeeze())
    opt.zero_grad()
    loss.backward()
    opt.step()
    arrs = dict(pred_u8=pred, feat14_u8=feat14, feat_u8=feat, fixsac=np.array(fixsac), lf_out=out.detach().numpy(),
                lf_loss=np.array(loss.item()), lf_aae_auc=np.array([aae, auc]))
    for k, p in lf.named_parameters():
        arrs["lf_after_sum/" + k] = np.array([p.detach().double().sum().item(), p.detach().double().norm().item()])
    print("config5: pred range", pred.min(), pred.max(), "feat14 range", feat14.min(), feat14.max(), "loss", loss.item())
    save("config5.npz", **arrs)



if __name__ == "__main__":
    if len(sys.argv) > 1:
        for name in sys.argv[1:]:
            {"config1": gen_config1, "config5": gen_config5, "extract_lstm": gen_extract_lstm}[name]()
        sys.exit(0)
    gen_floss()
    gen_lstm()
    gen_late_fusion()
    gen_metrics_and_glue()
    gen_model_sp(32, 2, "s32", head_gain=0.25)
    gen_model_sp(224, 2, "s224", head_gain=0.25)
    gen_config1()
    gen_extract_lstm()
    gen_config5()
