"""GPU parity of the AT path (lstmnet + MSE + Adam through the C-ABI) against the reference's golden
vectors (tests/golden/lstmnet.npz) and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import egaze_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def build():
    from egaze_amd.models.LSTMnet import lstmnet
    net = lstmnet()
    net.load_state_dict(synth.synth_state_dict(O.lstm_shapes(), seed=2))
    return net.to(DEV)


def test_gemm_variants():
    import egaze_amd.hipops as h
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(70, 130, generator=g), torch.randn(50, 130, generator=g)
    b = torch.randn(50, generator=g)
    xd, wd = x.to(DEV), w.to(DEV)
    assert rel(h.linear_fwd(xd, wd, bias=b.to(DEV)).cpu(), x @ w.t() + b) < 1e-5
    assert rel(h.linear_fwd(xd, wd, relu=True).cpu(), torch.relu(x @ w.t())) < 1e-5
    dy = torch.randn(70, 50, generator=g)
    assert rel(h.matmul_nn(dy.to(DEV), wd).cpu(), dy @ w) < 1e-5
    assert rel(h.matmul_tn(dy.to(DEV), xd).cpu(), dy.t() @ x) < 1e-5
    acc = torch.randn(70, 50, generator=g)
    out = acc.clone().to(DEV)
    h.linear_fwd(xd, wd, out=out, accumulate=True)
    assert rel(out.cpu(), acc + x @ w.t()) < 1e-5


def test_lstmnet_golden_t3b2():
    gold = np.load(os.path.join(GOLDEN, "lstmnet.npz"))
    from egaze_amd.functions import MSELoss
    net = build()
    inp, tgt = synth.synth_at_batch(3, 2, seed=3)
    h0 = torch.zeros(2, 2, 512, device=DEV)
    c0 = torch.zeros(2, 2, 512, device=DEV)
    out, (hn, cn) = net(inp.to(DEV), (h0, c0))
    loss = MSELoss.apply(out, torch.tanh(tgt).to(DEV))
    loss.backward()
    assert rel(out.detach().cpu().numpy(), gold["t3b2_out"]) < 1e-5
    assert rel(hn.detach().cpu().numpy(), gold["t3b2_hn"]) < 1e-5
    assert rel(cn.detach().cpu().numpy(), gold["t3b2_cn"]) < 1e-5
    assert abs(loss.item() - float(gold["t3b2_loss"])) < 1e-5 * abs(float(gold["t3b2_loss"]))
    for k, p in net.named_parameters():
        g = p.grad.double().cpu()
        got = np.array([g.norm().item(), g.sum().item(), g.abs().max().item()])
        assert rel(got, gold["t3b2_gsum/" + k]) < 1e-3, (k, got, gold["t3b2_gsum/" + k])
    assert rel(net.lin.bias.grad.cpu().numpy(), gold["t3b2_grad/lin.bias"]) < 1e-4
    assert rel(net.lstm.bias_ih_l1.grad.cpu().numpy(), gold["t3b2_grad/lstm.bias_ih_l1"]) < 1e-4


def test_lstmnet_hidden_none():
    gold = np.load(os.path.join(GOLDEN, "lstmnet.npz"))
    net = build()
    inp1, _ = synth.synth_at_batch(1, 1, seed=4)
    with torch.no_grad():
        out1, (h1, c1) = net(inp1.to(DEV), None)
    assert rel(out1.cpu().numpy(), gold["t1b1_out"]) < 1e-5
    assert rel(h1.cpu().numpy(), gold["t1b1_hn"]) < 1e-5
    inp, _ = synth.synth_at_batch(3, 2, seed=3)
    with pytest.raises(RuntimeError):                  # batch 2 with hidden=None: the reference raises too
        net(inp.to(DEV), None)


def test_at_train_replay_golden():
    """AT.trainLSTM (AT.py:118-147) replay: batch 1 / seq 1, detach hidden, off-by-one loss, Adam per sample."""
    from egaze_amd.functions import MSELoss
    from egaze_amd.optim import FusedAdam
    from egaze_amd.utils import repackage_hidden
    gold = np.load(os.path.join(GOLDEN, "lstmnet.npz"))
    net = build()
    opt = FusedAdam(net.parameters(), lr=1e-4)
    ins, tgts = synth.synth_at_batch(5, 1, seed=6)
    same = [1, 1, 1, 0, 1]
    hidden, pred, losses = None, None, []
    for i in range(5):
        if int(same[i]) == 0:
            hidden = None
        a = ins[i].unsqueeze(0).to(DEV)
        t = tgts[i].unsqueeze(0).to(DEV)
        if pred is not None:
            l = MSELoss.apply(pred, torch.tanh(t))
            opt.zero_grad()
            l.backward()
            opt.step()
            losses.append(l.item())
        hidden = repackage_hidden(hidden)
        pred, hidden = net(a, hidden)
    assert np.allclose(losses, gold["replay_losses"], rtol=1e-4, atol=0)
    assert rel(pred.detach().cpu().numpy(), gold["replay_final_pred"]) < 1e-4
    assert rel(net.lin.bias.detach().cpu().numpy(), gold["replay_lin_bias"]) < 1e-4
    ws = np.array([p.detach().double().sum().item() for p in net.parameters()])
    assert np.allclose(ws, gold["replay_w_sum"], rtol=1e-4, atol=1e-5)


def test_lstmnet_t16_b32_vs_oracle():
    """BASELINE config 4 shape: T=16, B=32 with explicit (h, c); forward + full gradients vs the oracle."""
    from egaze_amd.functions import MSELoss
    net = build()
    sd = synth.synth_state_dict(O.lstm_shapes(), seed=2)
    inp, tgt = synth.synth_at_batch(16, 32, seed=9)
    g = torch.Generator().manual_seed(3)
    h0, c0 = torch.randn(2, 32, 512, generator=g) * 0.1, torch.randn(2, 32, 512, generator=g) * 0.1
    out, (hn, cn) = net(inp.to(DEV), (h0.to(DEV), c0.to(DEV)))
    MSELoss.apply(out, torch.tanh(tgt).to(DEV)).backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oout, (ohn, ocn) = O.lstmnet_forward(leaves, inp, (h0, c0))
    O.mse(oout, torch.tanh(tgt)).backward()
    assert rel(out.detach().cpu().numpy(), oout.detach().numpy()) < 2e-5
    assert rel(hn.detach().cpu().numpy(), ohn.detach().numpy()) < 2e-5
    assert rel(cn.detach().cpu().numpy(), ocn.detach().numpy()) < 2e-5
    for k, p in net.named_parameters():
        assert rel(p.grad.cpu().numpy(), leaves[k].grad.numpy()) < 2e-4, k


@pytest.mark.parametrize("through_state", [False, True])
def test_lstmnet_single_step_fused_vs_oracle(through_state, monkeypatch):
    """T = 1, B = 1 (the reference's own stepping, AT.py:127-145) runs on the fused one-call path (csrc/lstm_b1.hip): forward,
    state and EVERY parameter gradient against the oracle -- with the loss on the output only (the AT loop) and with
    gradients also arriving through the returned (h, c) -- and against the sequence path of the same build."""
    import egaze_amd.models.LSTMnet as M
    from egaze_amd.functions import MSELoss
    sd = synth.synth_state_dict(O.lstm_shapes(), seed=2)
    inp, tgt = synth.synth_at_batch(1, 1, seed=11)
    g = torch.Generator().manual_seed(5)
    h0, c0 = torch.randn(2, 1, 512, generator=g) * 0.3, torch.randn(2, 1, 512, generator=g) * 0.3
    wh, wc = torch.randn(2, 1, 512, generator=g), torch.randn(2, 1, 512, generator=g)

    def run(fused):
        monkeypatch.setattr(M, "B1_FUSED", fused)
        net = build()
        out, (hn, cn) = net(inp.to(DEV), (h0.to(DEV), c0.to(DEV)))
        assert type(out.grad_fn).__name__.startswith("_LSTMNetB1Fn" if fused else "_LSTMNetFn")
        loss = MSELoss.apply(out, torch.tanh(tgt).to(DEV))
        if through_state:
            loss = loss + (hn * wh.to(DEV)).sum() * 1e-3 + (cn * wc.to(DEV)).sum() * 1e-3
        loss.backward()
        return out.detach().cpu(), hn.detach().cpu(), cn.detach().cpu(), {k: p.grad.cpu() for k, p in net.named_parameters()}

    out, hn, cn, grads = run(True)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oout, (ohn, ocn) = O.lstmnet_forward(leaves, inp, (h0, c0))
    oloss = O.mse(oout, torch.tanh(tgt))
    if through_state:
        oloss = oloss + (ohn * wh).sum() * 1e-3 + (ocn * wc).sum() * 1e-3
    oloss.backward()
    assert rel(out.numpy(), oout.detach().numpy()) < 1e-5
    assert rel(hn.numpy(), ohn.detach().numpy()) < 1e-5 and rel(cn.numpy(), ocn.detach().numpy()) < 1e-5
    for k in grads:
        assert rel(grads[k].numpy(), leaves[k].grad.numpy()) < 2e-5, k
    out2, hn2, cn2, grads2 = run(False)
    assert rel(out.numpy(), out2.numpy()) < 1e-5 and rel(hn.numpy(), hn2.numpy()) < 1e-5
    for k in grads:
        assert rel(grads[k].numpy(), grads2[k].numpy()) < 2e-5, k


def test_lstmnet_single_step_routes():
    """The fused path is taken only when nothing upstream wants a gradient; an input / state that requires grad goes through
    the sequence path (which produces those gradients), as does any T > 1 or B > 1 call."""
    net = build()
    inp, _ = synth.synth_at_batch(1, 1, seed=12)
    x = inp.to(DEV).requires_grad_(True)
    out, _ = net(x, None)
    assert type(out.grad_fn).__name__.startswith("_LSTMNetFn")
    out.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    out, _ = net(inp.to(DEV), None)
    assert type(out.grad_fn).__name__.startswith("_LSTMNetB1Fn")
    with torch.no_grad():
        o1, (h1, c1) = net(inp.to(DEV), None)
    assert rel(o1.cpu().numpy(), out.detach().cpu().numpy()) == 0.0


def _at_shell(net, opt):
    """An AT driver object with just what _epoch needs (the constructor wants a trained SP checkpoint and dataset folders)."""
    from egaze_amd.AT import AT
    from egaze_amd.functions import MSELoss
    at = AT.__new__(AT)
    at.lstm, at.optimizer_lstm, at.criterion_lstm, at.device = net, opt, MSELoss.apply, torch.device(DEV)
    return at


def test_at_epoch_graphed_matches_golden_and_eager(monkeypatch):
    """AT.trainLSTM's loop with the per-sample step replayed from a captured hipGraph (AT._epoch_graphed: deferred forward,
    static buffers, device-side Adam counter): the reference's own 5-sample replay fixture (video break included), and a
    24-sample run against the launch-by-launch loop of the same build -- same mean loss, same final parameters, same
    optimizer step count."""
    import egaze_amd.AT as at_mod
    from egaze_amd.optim import FusedAdam
    gold = np.load(os.path.join(GOLDEN, "lstmnet.npz"))
    ins, tgts = synth.synth_at_batch(5, 1, seed=6)
    same = [1, 1, 1, 0, 1]
    loader = [{"input": ins[i], "gt": tgts[i], "same": torch.tensor([same[i]])} for i in range(5)]
    monkeypatch.setattr(at_mod, "AT_GRAPH", True)
    net = build()
    opt = FusedAdam(net.parameters(), lr=1e-4)
    mean_loss = _at_shell(net, opt)._epoch(loader, True)
    assert abs(mean_loss - float(np.mean(gold["replay_losses"]))) < 1e-4 * abs(float(np.mean(gold["replay_losses"])))
    assert rel(net.lin.bias.detach().cpu().numpy(), gold["replay_lin_bias"]) < 1e-4
    ws = np.array([p.detach().double().sum().item() for p in net.parameters()])
    assert np.allclose(ws, gold["replay_w_sum"], rtol=1e-4, atol=1e-5)
    assert opt.step_count == 4 and not opt.capturable

    g = torch.Generator().manual_seed(8)
    n = 24
    flags = [1] * n
    flags[9] = flags[17] = 0
    loader = [{"input": torch.randn(1, 512, generator=g), "gt": torch.rand(1, 512, generator=g), "same": torch.tensor([flags[i]])}
              for i in range(n)]
    res = {}
    for graphed in (True, False):
        monkeypatch.setattr(at_mod, "AT_GRAPH", graphed)
        net = build()
        opt = FusedAdam(net.parameters(), lr=1e-4)
        loss = _at_shell(net, opt)._epoch(loader, True)
        res[graphed] = (loss, {k: p.detach().cpu().clone() for k, p in net.named_parameters()}, opt.step_count)
    assert res[True][2] == res[False][2] == n - 1
    assert abs(res[True][0] - res[False][0]) < 1e-5 * abs(res[False][0])
    for k in res[True][1]:
        assert rel(res[True][1][k].numpy(), res[False][1][k].numpy()) < 1e-5, k


def test_fused_adam_device_step_counter():
    """egz_adam_step_dev (counter on the device, for captured steps) against the host-counter form over several steps."""
    from egaze_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(3)
    w0 = torch.randn(1000, generator=g)
    grads = [torch.randn(1000, generator=g) for _ in range(6)]
    out = {}
    for cap in (False, True):
        p = torch.nn.Parameter(w0.clone().to(DEV))
        opt = FusedAdam([p], lr=1e-2)
        if cap:
            opt.set_capturable(True)
        for gr in grads:
            opt.zero_grad()
            p.grad.copy_(gr.to(DEV))
            opt.step()
        if cap:
            assert int(opt.step_dev[0].item()) == 6 and int(opt.step_dev[1].item()) == 0
            opt.set_capturable(False)
        assert opt.step_count == 6
        out[cap] = p.detach().cpu()
    assert rel(out[True].numpy(), out[False].numpy()) < 1e-6


def test_mse_fwd_grad_matches_two_calls():
    """egz_mse_fwd_grad (loss + its gradient for a unit seed in one single-block launch, the AT per-sample step) against egz_mse_fwd +
    egz_mse_bwd(grad_out = 1): bit-identical; the loss is parked in ring[counter % len(ring)]."""
    import egaze_amd.hipops as H
    g = torch.Generator().manual_seed(4)
    for n, tanh_t in ((512, True), (512, False), (37, True), (4096, False)):
        a = torch.randn(n, generator=g).to(DEV)
        b = torch.rand(n, generator=g).to(DEV)
        one = torch.ones((), device=DEV)
        l0, d0 = H.mse_fwd(a, b, tanh_t), H.mse_bwd(a, b, one, tanh_t)
        ring = torch.full((8,), -1.0, device=DEV)
        counter = torch.tensor([21, 0], dtype=torch.int32, device=DEV)
        l1, d1 = H.mse_fwd_grad(a, b, tanh_t, ring, counter)
        assert torch.equal(l0, l1) and torch.equal(d0, d1)
        want = torch.full((8,), -1.0)
        want[21 % 8] = l0.item()
        assert torch.equal(ring.cpu(), want)
        l2, d2 = H.mse_fwd_grad(a, b, tanh_t)
        assert torch.equal(l0, l2) and torch.equal(d0, d2)


@pytest.mark.parametrize("L,T,B", [(1, 1, 2), (1, 5, 20), (2, 2, 33), (3, 4, 7), (2, 16, 1)])
def test_lstm_wavefront_vs_torch(L, T, B):
    """egz_lstm_wave_fwd / _bwd (the stacked recurrence as a wavefront over (layer, step)) against torch's nn.LSTM on the CPU, fp64:
    one to three layers, a single step, ragged batch tiles (B = 20, 33: rows past the batch in the last 16 / 32-row tile), gradients
    flowing in through the outputs AND the returned state, out through the input and the initial state."""
    from egaze_amd.models.LSTMnet import lstmnet
    torch.manual_seed(L * 100 + T * 10 + B)
    net = lstmnet(num_layer=L).to(DEV)
    ref = torch.nn.LSTM(512, 512, L).double()
    lin = torch.nn.Linear(512, 512).double()
    with torch.no_grad():
        for name, p in ref.named_parameters():
            p.copy_(getattr(net.lstm, name).detach().cpu().double())
        lin.weight.copy_(net.lin.weight.detach().cpu().double())
        lin.bias.copy_(net.lin.bias.detach().cpu().double())
    g = torch.Generator().manual_seed(7)
    x = torch.randn(T, B, 512, generator=g)
    h0, c0 = torch.randn(L, B, 512, generator=g) * 0.3, torch.randn(L, B, 512, generator=g) * 0.3
    wo, wh, wc = torch.randn(T, B, 512, generator=g), torch.randn(L, B, 512, generator=g), torch.randn(L, B, 512, generator=g)
    xd, hd, cd = (t.to(DEV).requires_grad_(True) for t in (x, h0, c0))
    out, (hn, cn) = net(xd, (hd, cd))
    ((out * wo.to(DEV)).sum() + (hn * wh.to(DEV)).sum() + (cn * wc.to(DEV)).sum()).backward()
    xr, hr, cr = (t.double().requires_grad_(True) for t in (x, h0, c0))
    o, (rh, rc) = ref(torch.tanh(xr), (hr, cr))
    ro = torch.relu(lin(o))
    ((ro * wo.double()).sum() + (rh * wh.double()).sum() + (rc * wc.double()).sum()).backward()
    assert rel(out.detach().cpu().numpy(), ro.detach().numpy()) < 2e-5
    assert rel(hn.detach().cpu().numpy(), rh.detach().numpy()) < 2e-5 and rel(cn.detach().cpu().numpy(), rc.detach().numpy()) < 2e-5
    assert rel(xd.grad.cpu().numpy(), xr.grad.numpy()) < 1e-4
    assert rel(hd.grad.cpu().numpy(), hr.grad.numpy()) < 1e-4 and rel(cd.grad.cpu().numpy(), cr.grad.numpy()) < 1e-4
    for name, p in ref.named_parameters():
        assert rel(getattr(net.lstm, name).grad.cpu().numpy(), p.grad.numpy()) < 2e-4, name


def _persist_weights(g):
    w_ih = [None, (torch.randn(2048, 512, generator=g) * 0.05).to(DEV)]
    w_hh = [(torch.randn(2048, 512, generator=g) * 0.05).to(DEV) for _ in range(2)]
    b_ih = [(torch.randn(2048, generator=g) * 0.1).to(DEV) for _ in range(2)]
    b_hh = [(torch.randn(2048, generator=g) * 0.1).to(DEV) for _ in range(2)]
    return w_ih, w_hh, b_ih, b_hh


@pytest.mark.parametrize("T,B", [(16, 32), (1, 1), (3, 16), (5, 20), (16, 7)])
def test_lstm_persistent_forward_matches_wavefront(T, B):
    """egz_lstm_persist_fwd (ONE weight-stationary launch, blocks hand h_t to each other inside it, biases summed in the kernel)
    against egz_lstm_wave_fwd (T + 1 launches, bias sum folded into the projection) on the AT network's geometry
    (models/LSTMnet.py:18: nn.LSTM(512, 512, 2)): every output -- all h_t and c_t of both layers, the gate activations the backward
    pass reads, the returned state.  Same products in another summation order (8 K-slices of 64 instead of 4 of 128): equal to a
    few ulp.  40 calls on recycled buffers with fresh inputs: a hand-off that read a stale line (the previous call's h at the same
    address) would show up as a mismatch; the status word must stay 0."""
    from egaze_amd import hipops as H
    g = torch.Generator().manual_seed(11)
    w_ih, w_hh, b_ih, b_hh = _persist_weights(g)
    bsum = [a + b for a, b in zip(b_ih, b_hh)]
    for it in range(40 if (T, B) == (16, 32) else 3):
        gx0 = torch.randn(T, B, 2048, generator=g).to(DEV)
        h0, c0 = (torch.randn(2, B, 512, generator=g) * 0.5).to(DEV), (torch.randn(2, B, 512, generator=g) * 0.5).to(DEV)
        want = H.lstm_wave_fwd(gx0 + bsum[0], w_ih, w_hh, bsum, h0, c0)
        got = H.lstm_persist_fwd(gx0, w_ih, w_hh, b_ih, b_hh, h0, c0)
        assert H.lstm_persist_status() == 0
        for name, a, b in zip(("hs", "cs", "acts", "hn", "cn"), got, want):
            assert torch.allclose(a, b, rtol=0, atol=3e-6), (it, name, float((a - b).abs().max()))
        del want, got
    # no-grad form (no gate activations kept)
    got = H.lstm_persist_fwd(gx0, w_ih, w_hh, b_ih, b_hh, h0, c0, want_acts=False)
    assert got[2] is None and H.lstm_persist_status() == 0
    want = H.lstm_wave_fwd(gx0 + bsum[0], w_ih, w_hh, bsum, h0, c0, want_acts=False)
    assert torch.allclose(got[0], want[0], rtol=0, atol=3e-6) and torch.allclose(got[3], want[3], rtol=0, atol=3e-6)
    # any other geometry is refused, nothing launched
    with pytest.raises(RuntimeError, match="built for L = 2"):
        H.lstm_persist_fwd(torch.zeros(2, 33, 2048, device=DEV), w_ih, w_hh, b_ih, b_hh, torch.zeros(2, 33, 512, device=DEV),
                           torch.zeros(2, 33, 512, device=DEV))


@pytest.mark.parametrize("T,B,top,state", [(16, 32, True, True), (16, 32, True, False), (1, 1, True, True), (3, 16, False, True),
                                           (5, 20, True, True), (16, 7, True, False), (2, 9, True, True)])
def test_lstm_persistent_backward_matches_wavefront(T, B, top, state):
    """egz_lstm_persist_bwd (one launch: 8-row batch tiles, v_mfma_f32_4x4x1 products, dgates handed over inside the launch, weights
    read untransposed, bias gradients folded by the last-arriving tile) against egz_lstm_wave_bwd (T + 3 launches on transposed
    weight copies) + egz_colsum on activations of a real forward pass: dgates of every step and layer, dh0, dc0, the four bias
    gradients.  With / without a gradient through the outputs and through the returned state (models/LSTMnet.py:35 returns both),
    ragged 8-row tiles (B = 20, 7, 9, 1), T = 1.  Repeated on recycled buffers like the forward test; bitwise repeatable."""
    from egaze_amd import hipops as H
    g = torch.Generator().manual_seed(13 + T + B)
    w_ih, w_hh, b_ih, b_hh = _persist_weights(g)
    bsum = [a + b for a, b in zip(b_ih, b_hh)]
    w_hh_t = [w.t().contiguous() for w in w_hh]
    w_ih_t = [None, w_ih[1].t().contiguous()]
    for it in range(30 if (T, B) == (16, 32) else 3):
        gx0 = torch.randn(T, B, 2048, generator=g).to(DEV)
        h0, c0 = (torch.randn(2, B, 512, generator=g) * 0.5).to(DEV), (torch.randn(2, B, 512, generator=g) * 0.5).to(DEV)
        hs, cs, acts, hn, cn = H.lstm_wave_fwd(gx0, w_ih, w_hh, bsum, h0, c0)
        dh_top = torch.randn(T, B, 512, generator=g).to(DEV) if top else None
        dhn = torch.randn(2, B, 512, generator=g).to(DEV) if state else None
        dcn = torch.randn(2, B, 512, generator=g).to(DEV) if state else None
        want = H.lstm_wave_bwd(dh_top, dhn, dcn, acts, cs, c0, w_hh_t, w_ih_t)
        db = [torch.full((2048,), 7.0, device=DEV) for _ in range(4)]
        db[1] = None                                             # a bias whose gradient is not wanted
        got = H.lstm_persist_bwd(dh_top, dhn, dcn, acts, cs, c0, w_hh, w_ih, db)
        assert H.lstm_persist_status() == 0
        for name, a, b in zip(("dgates", "dh0", "dc0"), got, want):
            scale = float(b.abs().max()) + 1e-30
            assert float((a - b).abs().max()) <= 2e-5 * scale, (it, name, float((a - b).abs().max()), scale)
        for i in (0, 2, 3):
            ref = H.colsum(want[0][i // 2].view(T * B, 2048))
            assert float((db[i] - ref).abs().max()) <= 2e-5 * (float(ref.abs().max()) + 1e-30), (it, "db", i)
        if it == 0:
            db2 = [torch.empty(2048, device=DEV) for _ in range(4)]
            again = H.lstm_persist_bwd(dh_top, dhn, dcn, acts, cs, c0, w_hh, w_ih, db2)
            assert all(torch.equal(a, b) for a, b in zip(got, again)) and all(torch.equal(db[i], db2[i]) for i in (0, 2, 3))
            assert torch.equal(db2[0], db2[1])
        del want, got


def test_at_batched_step_graphed_matches_eager(monkeypatch):
    """The T = 16 / B = 32 AT step (lstmnet forward + MSE + backward + Adam; AT.py:138-145 at BASELINE config 4's shape) captured
    into one hipGraph (graphs.GraphedTrainStep: the two persistent recurrence launches and their counter memsets become graph nodes)
    against the same steps issued eagerly: identical parameters after 5 steps, status word clean."""
    from egaze_amd import hipops as H
    from egaze_amd.functions import MSELoss
    from egaze_amd.graphs import GraphedTrainStep
    from egaze_amd.models.LSTMnet import lstmnet
    from egaze_amd.optim import FusedAdam
    monkeypatch.setattr(H, "LSTM_PERSIST", True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 32, 512, generator=g).to(DEV)
    tgt = torch.tanh(torch.randn(16, 32, 512, generator=g)).to(DEV)
    h0, c0 = torch.zeros(2, 32, 512, device=DEV), torch.zeros(2, 32, 512, device=DEV)
    torch.manual_seed(3)
    nets = [lstmnet().to(DEV) for _ in range(2)]
    nets[1].load_state_dict(nets[0].state_dict())
    opts = [FusedAdam(n.parameters(), lr=1e-3) for n in nets]

    def make(net):
        def forward_loss(a, b):
            pred, _ = net(a, (h0, c0))
            return MSELoss.apply(pred, b), pred
        return forward_loss
    losses = []
    for _ in range(5):
        opts[0].zero_grad()
        l, _ = make(nets[0])(x, tgt)
        l.backward()
        opts[0].step()
        losses.append(float(l))
    gs = GraphedTrainStep(make(nets[1]), opts[1], (x, tgt))
    glosses = [float(gs(x, tgt)[0]) for _ in range(5)]
    gs.close()
    assert H.lstm_persist_status() == 0
    assert glosses == losses, (glosses, losses)
    for (n, a), b in zip(nets[0].state_dict().items(), nets[1].state_dict().values()):
        assert torch.equal(a, b), n


@pytest.mark.gpu
def test_lstm_persistent_failure_words_are_sticky_and_checked():
    """a persistent launch that loses a hand-off raises a STICKY word in the tail of its (device, stream) scratch that
    no later launch clears (forward and backward kept apart), hipops.lstm_persist_check() turns it into an exception at the
    loops' synchronisation points and switches the process to the wavefront launches.  The time-out itself cannot be provoked
    on an idle device, so the test plants the words where the kernels would, and checks that real launches leave them alone."""
    from egaze_amd import hipops as H
    if not H.lstm_persist_ok(2, 32, 512):
        pytest.skip("persistent LSTM launches not available on this device")
    T, B = 4, 32
    g = torch.Generator().manual_seed(3)
    w_ih = [None, (torch.randn(2048, 512, generator=g) * 0.04).to(DEV)]
    w_hh = [(torch.randn(2048, 512, generator=g) * 0.04).to(DEV) for _ in range(2)]
    b_ih = [torch.zeros(2048, device=DEV) for _ in range(2)]
    b_hh = [torch.zeros(2048, device=DEV) for _ in range(2)]
    gx0 = (torch.randn(T, B, 2048, generator=g) * 0.1).to(DEV)
    h0, c0 = torch.zeros(2, B, 512, device=DEV), torch.zeros(2, B, 512, device=DEV)
    hs, cs, acts, hn, cn = H.lstm_persist_fwd(gx0, w_ih, w_hh, b_ih, b_hh, h0, c0)
    H.lstm_persist_check()                                   # clean
    assert H.lstm_persist_errors() == {"fwd": 0, "bwd": 0}
    sync = next(iter(H._PERSIST_SYNC.values()))
    assert sync.numel() == H.LIB.egz_lstm_persist_sync_words()
    sync[-32] = 7                                            # "a forward launch gave up in global step 6"
    H.lstm_persist_fwd(gx0, w_ih, w_hh, b_ih, b_hh, h0, c0)  # the launches that follow (the counters ARE zeroed by each call) ...
    dh = torch.randn(T, B, 512, generator=g).to(DEV)
    H.lstm_persist_bwd(dh, None, None, acts, cs, c0, w_hh, w_ih, None)
    assert H.lstm_persist_errors() == {"fwd": 7, "bwd": 0}  # ... do not clear it, and the backward word is its own
    assert int(sync[1024].item()) == 0                       # per-launch word of the (clean) last launch
    was = H.LSTM_PERSIST
    try:
        with pytest.raises(RuntimeError, match="hand-off"):
            H.lstm_persist_check()
        assert H.LSTM_PERSIST is False and not H.lstm_persist_ok(2, 32, 512)      # wavefront launches from here on
        H.lstm_persist_check()                               # reset by the failed check
    finally:
        H.LSTM_PERSIST = was
    # a sequence whose extent would overflow the 32-bit offsets of the persistent form is refused, not launched
    tabs = [H._ptr_table(t) for t in ([None, w_ih[1]], w_hh, b_ih, b_hh)]
    rc = H._RAW_LIB.egz_lstm_persist_fwd(gx0.data_ptr(), *tabs, h0.data_ptr(), c0.data_ptr(), hs.data_ptr(),
                                         cs.data_ptr(), None, hn.data_ptr(), cn.data_ptr(), sync.data_ptr(), 2, 1 << 14, 32, 512, None)
    assert rc == 801, rc                                     # hipErrorNotSupported: the caller launches the wavefront
