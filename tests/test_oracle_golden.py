"""Pins the CPU oracle (oracle/egaze_oracle.py) to golden vectors produced by the REAL
reference (tests/golden/make_golden.py).  CPU only."""
import collections
import os

import numpy as np
import pytest
import torch

from oracle import egaze_oracle as O
from oracle import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def g(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def gsum(t):
    t = t.double()
    return np.array([t.norm().item(), t.sum().item(), t.abs().max().item()])


def test_floss_weights_and_grad():
    gold = g("floss.npz")
    rs = np.random.RandomState(5)
    size = 224
    gt = synth.synth_gt(3, size, rs)
    single = np.zeros((1, 1, size, size), np.float32); single[0, 0, 37, 181] = 1.0
    flat = np.full((1, 1, size, size), 0.25, np.float32)
    two = np.zeros((1, 1, size, size), np.float32); two[0, 0, 10, 20] = 0.5; two[0, 0, 200, 101] = 0.5
    target = np.concatenate([gt, single, flat, two], 0)
    x = rs.uniform(0.02, 0.98, target.shape).astype(np.float32)
    x[0, 0, 0, :4] = [0.0, 1.0, 1e-30, 1 - 1e-7]
    assert np.array_equal(x[[0, 3]], gold["x"])
    w = O.floss_weights(target)
    assert np.array_equal(w[:, 0, ::37, :], gold["weights_rows"])           # bit-exact
    assert np.allclose(w.astype(np.float64).sum(axis=(1, 2, 3)), gold["weights_sum"], rtol=0, atol=0)
    xin = torch.from_numpy(x).requires_grad_(True)
    loss = O.floss_forward(xin, torch.from_numpy(target))
    loss.backward()
    assert abs(loss.item() - gold["loss"]) <= 1e-6 * abs(gold["loss"])
    assert rel(xin.grad[0, 0].numpy(), gold["grad_b0"]) < 1e-6
    assert rel(xin.grad[3, 0].numpy(), gold["grad_b3"]) < 1e-6
    assert rel(xin.grad.double().sum(dim=(1, 2, 3)).numpy(), gold["grad_sum"]) < 1e-6


def test_lstmnet():
    gold = g("lstmnet.npz")
    sd = synth.synth_state_dict(O.lstm_shapes(), seed=2)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    inp, tgt = synth.synth_at_batch(3, 2, seed=3)
    out, (hn, cn) = O.lstmnet_forward(leaves, inp, (torch.zeros(2, 2, 512), torch.zeros(2, 2, 512)))
    loss = O.mse(out, torch.tanh(tgt))
    loss.backward()
    assert rel(out.detach().numpy(), gold["t3b2_out"]) < 2e-6
    assert rel(hn.detach().numpy(), gold["t3b2_hn"]) < 2e-6
    assert rel(cn.detach().numpy(), gold["t3b2_cn"]) < 2e-6
    assert abs(loss.item() - gold["t3b2_loss"]) < 1e-6 * abs(gold["t3b2_loss"])
    for k in sd:
        assert rel(gsum(leaves[k].grad), gold["t3b2_gsum/" + k]) < 1e-4, k
    assert rel(leaves["lin.bias"].grad.numpy(), gold["t3b2_grad/lin.bias"]) < 1e-5
    assert rel(leaves["lstm.bias_ih_l1"].grad.numpy(), gold["t3b2_grad/lstm.bias_ih_l1"]) < 1e-5
    inp1, _ = synth.synth_at_batch(1, 1, seed=4)
    out1, (h1, _) = O.lstmnet_forward(sd, inp1, None)
    assert rel(out1.numpy(), gold["t1b1_out"]) < 2e-6
    assert rel(h1.numpy(), gold["t1b1_hn"]) < 2e-6
    assert int(gold["b2_none_raises"]) == 1
    with pytest.raises(RuntimeError):
        O.lstmnet_forward(sd, inp, None)


def test_at_train_replay():
    gold = g("lstmnet.npz")
    sd = synth.synth_state_dict(O.lstm_shapes(), seed=2)
    ins, tgts = synth.synth_at_batch(5, 1, seed=6)
    losses = O.at_train_replay(sd, {}, ins, tgts, [1, 1, 1, 0, 1], lr=1e-4)
    assert np.allclose(losses, gold["replay_losses"], rtol=2e-5, atol=0)
    assert rel(sd["lin.bias"].numpy(), gold["replay_lin_bias"]) < 1e-5
    order = ["lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0",
             "lstm.weight_ih_l1", "lstm.weight_hh_l1", "lstm.bias_ih_l1", "lstm.bias_hh_l1",
             "lin.weight", "lin.bias"]
    ws = np.array([sd[k].double().sum().item() for k in order])
    assert np.allclose(ws, gold["replay_w_sum"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag,size", [("s32", 32), ("s224", 224)])
def test_late_fusion(tag, size):
    gold = g("late_fusion.npz")
    sd = synth.synth_state_dict(O.lf_shapes(), seed=3, head_gain=0.5)
    im, feat, gt = synth.synth_lf_batch(2, size, seed=7)
    with torch.no_grad():
        ev = O.late_fusion_forward(dict(sd), feat, im, training=False)
        sw = O.late_fusion_forward(dict(sd), im, feat, training=False)
    assert rel(ev.numpy(), gold[f"{tag}_eval_out"]) < 2e-6
    assert abs(sw.double().sum().item() - gold[f"{tag}_eval_out_swapped_sum"]) < 1e-5 * abs(gold[f"{tag}_eval_out_swapped_sum"])
    work = {k: v.clone() for k, v in sd.items()}
    keys = O.trainable_keys(work)
    for k in keys:
        work[k].requires_grad_(True)
    out = O.late_fusion_forward(work, feat, im, training=True)
    loss = O.floss_forward(out, gt)
    loss.backward()
    assert rel(out.detach().numpy(), gold[f"{tag}_train_out"]) < 5e-6
    assert abs(loss.item() - gold[f"{tag}_loss"]) < 1e-5 * abs(gold[f"{tag}_loss"])
    for k in keys:
        assert rel(work[k].grad.numpy(), gold[f"{tag}_grad/{k}"]) < 2e-3, k
    for k in work:
        if "running_" in k:
            assert rel(work[k].numpy(), gold[f"{tag}_after/{k}"]) < 1e-5, k


@pytest.mark.parametrize("tag,size", [("s32", 32), ("s224", 224)])
def test_model_sp(tag, size):
    gold = g(f"model_sp_{tag}.npz")
    shapes = O.sp_shapes()
    assert len(shapes) == 215
    sd = synth.synth_state_dict(shapes, seed=1, head_gain=0.25)
    x_s, x_t, gt, _ = synth.synth_sp_batch(2, size, seed=0)
    with torch.no_grad():
        ev, aux = O.sp_forward({k: v.clone() for k, v in sd.items()}, x_s, x_t, training=False)
    assert rel(ev.numpy(), gold["eval_out"]) < 1e-5
    assert rel(aux["features_s"].double().sum(dim=(2, 3)).numpy(), gold["eval_features_s_sum"]) < 1e-5
    assert rel(aux["features_s"][0, 0].numpy(), gold["eval_features_s_b0c0"]) < 1e-5
    work = {k: v.clone() for k, v in sd.items()}
    opt = {}
    lr = float(gold["lr"])
    before = {k: v.clone() for k, v in work.items()}
    loss, out, grads = O.sp_train_step(work, opt, 1, x_s, x_t, gt, lr)
    assert rel(out.numpy(), gold["train_out"]) < 1e-5
    assert abs(loss.item() - gold["train_loss"]) < 1e-5 * abs(gold["train_loss"])
    gold_keys = {k[5:] for k in gold.files if k.startswith("gsum/")}
    assert gold_keys == set(O.trainable_keys(sd))
    # conv biases in front of a train-mode BN have an analytically ZERO gradient (what the
    # reference reports for them is round-off noise), so the floor is relative to the model.
    floor = 1e-6 * max(gold["gsum/" + k][0] for k in gold_keys)
    for k in gold_keys:
        got, want = gsum(grads[k]), gold["gsum/" + k]
        assert abs(got[0] - want[0]) <= 2e-3 * want[0] + floor, (k, got, want)
    for k in [f[5:] for f in gold.files if f.startswith("grad/")]:
        if gold["gsum/" + k][0] > 100 * floor:
            assert rel(grads[k].numpy(), gold["grad/" + k]) < 2e-3, k
    for f in gold.files:
        if f.startswith("after/"):
            assert rel(work[f[6:]].numpy(), gold[f]) < 1e-5, f
        elif f.startswith("after_sum/"):
            v = work[f[10:]].double()
            assert np.allclose([v.sum().item(), v.norm().item()], gold[f], rtol=1e-5), f
        elif f.startswith("delta/") and gold["gsum/" + f[6:]][0] > 100 * floor:
            d = (work[f[6:]] - before[f[6:]]).double()
            assert abs(d.abs().max().item() - gold[f][1]) < 1e-3 * lr + 1e-9, f
    assert int(work["bn.num_batches_tracked"]) == 1


def test_metrics_and_glue():
    gold = g("metrics_glue.npz")
    rs = np.random.RandomState(11)
    gt = synth.synth_gt(3, 224, rs)[:, 0]
    pred = synth.synth_gt(3, 224, rs)[:, 0] * 0.8 + rs.uniform(0, 0.05, (3, 224, 224)).astype(np.float32)
    aae, auc, gp = O.compute_aae_auc(pred, gt)
    assert np.allclose([aae, auc], gold["batch_aae_auc"], rtol=1e-12)
    assert np.array_equal(np.array(gp), gold["batch_gp"])
    aae1, auc1, gp1 = O.compute_aae_auc(pred[1], gt[1])
    assert np.allclose([aae1, auc1], gold["single_aae_auc"], rtol=1e-12)
    assert np.array_equal(np.array(gp1), gold["single_gp"])
    aae2, auc2, _ = O.compute_aae_auc(np.uint8(255 * np.clip(pred[2], 0, 1)), gt[2])
    assert np.allclose([aae2, auc2], gold["u8_aae_auc"], rtol=1e-12)
    od = collections.OrderedDict()
    krs = np.random.RandomState(12)
    od["features.0.weight"] = torch.from_numpy(krs.standard_normal((64, 3, 3, 3)).astype(np.float32))
    for n in range(1, 30):
        od[f"features.k{n}"] = torch.from_numpy(krs.standard_normal((4,)).astype(np.float32))
    new = O.change_key_names(od, 20)
    assert list(new.keys()) == list(gold["ckn_keys"])
    assert np.array_equal(new["features.0.weight"].numpy(), gold["ckn_w0"])
    feat = torch.from_numpy(np.abs(krs.standard_normal((2, 512, 14, 14))).astype(np.float32))
    cf = O.crop_feature(feat, [[5, 220], [117, 60]], 3)
    assert np.array_equal(cf.numpy(), gold["crop_feature"])
    w = cf.contiguous().view(2, 512, -1).mean(2)
    assert rel(O.get_weighted(w[0], feat[0:1]).numpy(), gold["get_weighted"]) < 1e-6


def test_config1_run_spatialstream_plumbing():
    """BASELINE config 1 (run_spatialstream.py on CPU): every stage of the plumbing against the reference."""
    gold = g("config1.npz")
    shapes = O.spatial_vgg_shapes()
    assert sum(int(np.prod(v)) for k, v in shapes.items()
               if not ("running" in k or "tracked" in k)) == int(gold["n_params"]) == 31795457
    sd = synth.synth_state_dict(shapes, seed=4, head_gain=0.25)
    sd_lf = synth.synth_state_dict(O.lf_shapes(), seed=3, head_gain=0.5)
    im_u8 = np.random.RandomState(21).randint(0, 256, (224, 224, 3)).astype(np.uint8)
    with torch.no_grad():
        r = O.config1_pipeline(sd, sd_lf, im_u8)
    assert rel(r["out"].numpy(), gold["out"]) < 1e-5
    assert rel(r["feat"].double().sum(dim=(2, 3)).numpy(), gold["feat_sum"]) < 1e-5
    assert np.array_equal(r["imq"], gold["imq"])
    assert np.allclose(r["predicted"], gold["predicted"], rtol=1e-12)
    assert rel(r["vec"].numpy(), gold["vec"]) < 1e-5
    assert rel(r["weighted"].numpy(), gold["weighted"]) < 1e-5
    assert rel(r["fin"].numpy(), gold["fin"]) < 1e-5
