"""Mirror of the reference's ``models/LSTMnet.py``: the AT module's network
``tanh -> nn.LSTM(512, 512, num_layers=2) -> Linear(512, 512) -> ReLU`` (models/LSTMnet.py:15-37).

Same constructor, attributes (lstm, tanh, lin, relu, num_channel, num_layer) and state-dict keys
(``lstm.weight_ih_l0`` ... ``lin.bias``); ``forward`` runs one autograd node whose forward and BPTT backward
are sequences of C-ABI launches (f32-MFMA GEMMs + fused LSTM-cell kernels).  Quirk kept: ``hidden is None``
means zeros of batch **1** (module-global ``batch_size``, LSTMnet.py:13,30-31), so a batch > 1 without an
explicit ``(h, c)`` raises exactly like the reference.
"""
import os

import torch
import torch.nn as nn

from .. import hipops as H

batch_size = 1
B1_FUSED = True      # T = 1, B = 1 steps on the fused single-step kernels (False: always the sequence path; test_hip_at flips it)


class _LSTMNetFn(torch.autograd.Function):
    """input (T,B,C), h0/c0 (L,B,H), then per layer (w_ih, w_hh, b_ih, b_hh), then lin.weight, lin.bias.

    ONE batched GEMM for the input projections of layer 0, then the stacked recurrence as a WAVEFRONT over (layer, step): launch
    s runs step s - l of every layer l (csrc/lstm_seq.hip, egz_lstm_wave_fwd: T + L - 1 fused [[h_below | h] [W_ih | W_hh]^T +
    cell] launches issued by one C-ABI call instead of T x L step launches and L - 1 more GEMMs); backward: T + L fused
    [[dgates_next | dgates_above] [W_hh | W_ih_above] + cell backward] launches, then batched GEMMs for dW_ih, dW_hh, db and the
    gradient into the tanh'd input.  Parameter gradients go straight into the optimizer's flat buffer when it offers a sink
    (hipops.GradSink)."""

    @staticmethod
    def forward(ctx, inp, h0, c0, *params):
        L = (len(params) - 2) // 4
        T, B, C = inp.shape
        Hd = params[1].shape[1]
        inp_c = H._req(inp.detach().contiguous(), "input")
        h0c, c0c = H._req(h0.detach().contiguous(), "h0"), H._req(c0.detach().contiguous(), "c0")
        train = any(ctx.needs_input_grad)          # no-grad runs (AT.testLSTM, extract_late) skip the saved gate activations
        x = H.tanh_fwd(inp_c)
        w_ih = [params[4 * l].detach() for l in range(L)]
        w_hh = [H._req(params[4 * l + 1].detach(), "w_hh") for l in range(L)]
        persist = C == Hd and H.lstm_persist_ok(L, B, Hd)
        # hs: (L, T + 1, B, H) -- slot 0 of a layer holds its h0, slots 1 .. T the outputs
        if persist:
            # ONE persistent weight-stationary launch (csrc/lstm_seq.hip): it adds b_ih + b_hh itself, so the projection of all time
            # steps carries no bias and no bias-sum launches precede it
            gx0 = H.linear_fwd(x.view(T * B, C), w_ih[0]).view(T, B, 4 * Hd)
            hs, cs, acts, hn, cn = H.lstm_persist_fwd(gx0, w_ih, w_hh, [params[4 * l + 2].detach() for l in range(L)],
                                                      [params[4 * l + 3].detach() for l in range(L)], h0c, c0c, want_acts=train)
        else:
            bsum = [H.add(params[4 * l + 2].detach(), params[4 * l + 3].detach()) for l in range(L)]
            gx0 = H.linear_fwd(x.view(T * B, C), w_ih[0], bias=bsum[0]).view(T, B, 4 * Hd)      # all time steps at once
            hs, cs, acts, hn, cn = H.lstm_wave_fwd(gx0, w_ih, w_hh, bsum, h0c, c0c, want_acts=train)
        lin_w, lin_b = params[-2].detach(), params[-1].detach()
        out2d = H.linear_fwd(hs[L - 1, 1:].view(T * B, Hd), lin_w, bias=lin_b, relu=True)
        # the node keeps the 2-D base and hands out a VIEW: the returned tensor (whose grad_fn is this node) must not be
        # stored on the node itself -- that reference cycle runs through C++ and is never collected, leaking the
        # saved T*B*4H activations of every step and keeping the step's AccumulateGrad nodes alive into the next one
        ctx.saved = (x, h0c, c0c, hs, cs, acts, out2d)
        ctx.params = list(params)
        ctx.dims = (T, B, C, Hd, L)
        ctx.persist = persist
        ctx.set_materialize_grads(False)
        return out2d.view(T, B, lin_w.shape[0]), hn, cn

    @staticmethod
    def backward(ctx, dout, dhn, dcn):
        if ctx.saved is None:
            raise RuntimeError("lstmnet: backward a second time through the same forward pass (activations were freed)")
        x, h0c, c0c, hs, cs, acts, out = ctx.saved
        params = ctx.params
        T, B, C, Hd, L = ctx.dims
        ng = ctx.needs_input_grad
        if acts is None:
            raise RuntimeError("lstmnet: backward through a forward pass that ran without gradient tracking")
        lin_w = params[-2].detach()
        if dout is None:
            dpre = torch.empty((T * B, out.shape[-1]), dtype=torch.float32, device=out.device)
            H.fill_zero(dpre)
        else:
            dpre = H.relu_bwd(out.view(T * B, -1), H._req(dout.contiguous().view(T * B, -1), "grad"))
        grads = [None] * len(params)
        sinks = [H.grad_sink(p, ng[3 + i]) for i, p in enumerate(params)]
        h_top = hs[L - 1, 1:].view(T * B, Hd)
        if ng[3 + len(params) - 2]:
            grads[-2] = H.matmul_tn(dpre, h_top, out=sinks[-2])                  # d lin.weight = dpre^T h
        if ng[3 + len(params) - 1]:
            grads[-1] = H.colsum(dpre, out=sinks[-1])
        dh_top = H.matmul_nn(dpre, lin_w).view(T, B, Hd)          # gradient into the top layer's outputs
        bias_done = False
        if ctx.persist:
            # one persistent launch: reads the weights as they are (no transposed copies) and forms the bias gradients -- the column
            # sums of dgates -- on the way, straight into their sinks
            db = []
            for l in range(L):
                for i in (4 * l + 2, 4 * l + 3):
                    if ng[3 + i]:
                        grads[i] = H._out(sinks[i], (4 * Hd,), out.device)
                        db.append(grads[i])
                    else:
                        db.append(None)
            dgates, dh0, dc0 = H.lstm_persist_bwd(dh_top, dhn.contiguous() if dhn is not None else None,
                                                  dcn.contiguous() if dcn is not None else None, acts, cs, c0c,
                                                  [params[4 * l + 1].detach() for l in range(L)],
                                                  [params[4 * l].detach() for l in range(L)], db)
            bias_done = True
        else:
            w_hh_t = [H.transpose2d(H._req(params[4 * l + 1].detach(), "w_hh")) for l in range(L)]
            w_ih_t = [None] + [H.transpose2d(H._req(params[4 * l].detach(), "w_ih")) for l in range(1, L)]
            dgates, dh0, dc0 = H.lstm_wave_bwd(dh_top, dhn.contiguous() if dhn is not None else None,
                                               dcn.contiguous() if dcn is not None else None, acts, cs, c0c, w_hh_t, w_ih_t)
        # d W_ih = dgates^T layer_in, d W_hh = sum_t dgates_t^T h_{t-1} (slots 0 .. T - 1 of hs are (h0, h_1 .. h_{T-1}): one product
        # each) -- all 2 L products of one shape in ONE launch when C == H (egz_gemm_batched)
        prods = []
        for l in range(L):
            layer_in = x.view(T * B, C) if l == 0 else hs[l - 1, 1:].view(T * B, Hd)
            dg2 = dgates[l].view(T * B, 4 * Hd)
            if ng[3 + 4 * l]:
                prods.append((4 * l, dg2, layer_in))
            if ng[3 + 4 * l + 1]:
                prods.append((4 * l + 1, dg2, hs[l, :T].view(T * B, Hd)))
        if len(prods) > 1 and len({tuple(b.shape) for _, _, b in prods}) == 1 and len(prods) <= 8:
            got = H.matmul_tn_batched([a for _, a, _ in prods], [b for _, _, b in prods], [sinks[i] for i, _, _ in prods])
            for (i, _, _), g in zip(prods, got):
                grads[i] = g
        else:
            for i, a, b in prods:
                grads[i] = H.matmul_tn(a, b, out=sinks[i])
        for l in range(0 if not bias_done else L, L):
            dg2 = dgates[l].view(T * B, 4 * Hd)
            # d b_ih = d b_hh = the column sums of dgates: summed once, the second one is a copy of the first
            gb = None
            for i in (4 * l + 2, 4 * l + 3):
                if not ng[3 + i]:
                    continue
                if gb is None:
                    gb = grads[i] = H.colsum(dg2, out=sinks[i])
                else:
                    grads[i] = H._out(sinks[i], tuple(gb.shape), gb.device)
                    H.copy_into(grads[i], gb)
        dinp = None
        if ng[0]:
            dh_in = H.matmul_nn(dgates[0].view(T * B, 4 * Hd), params[0].detach())        # into the tanh'd input
            dinp = H.tanh_bwd(x, dh_in).view(T, B, C)
        for i, p in enumerate(params):
            if sinks[i] is not None:
                H.grad_done(p)
                grads[i] = None
        ctx.saved = None              # activations are dead after one backward pass (as with save_for_backward)
        return (dinp, dh0 if ng[1] else None, dc0 if ng[2] else None, *grads)


class _LSTMNetB1Fn(torch.autograd.Function):
    """The same network at T = 1, B = 1 -- how the reference itself steps it (AT.py:127-145 per fixation sample, AT.py:246 per
    frame): ONE C-ABI call per direction (csrc/lstm_b1.hip: matrix-vector kernels, rank-1 weight gradients written straight
    into the optimizer's buffer) instead of ~25 launches each with a host round trip.  No gradient flows to the input or
    the initial state on this path (lstmnet.forward routes such calls to the sequence path)."""

    @staticmethod
    def forward(ctx, inp, h0, c0, *params):
        train = any(ctx.needs_input_grad)
        ps = [p.detach() for p in params]
        h0c = H._req(h0.detach().contiguous(), "h0").view(h0.shape[0], -1)
        c0c = H._req(c0.detach().contiguous(), "c0").view(c0.shape[0], -1)
        xt, acts, hn, cn, out = H.lstm_b1_fwd(ps, inp.detach().contiguous().view(-1), h0c, c0c, want_acts=train)
        ctx.saved = (xt, acts, h0c, c0c, hn, cn, out)
        ctx.params = list(params)
        ctx.set_materialize_grads(False)
        return out.view(1, 1, -1), hn.view(hn.shape[0], 1, -1), cn.view(cn.shape[0], 1, -1)

    @staticmethod
    def backward(ctx, dout, dhn, dcn):
        xt, acts, h0c, c0c, hn, cn, out = ctx.saved
        params = ctx.params
        if acts is None:
            raise RuntimeError("lstmnet: backward through a forward pass that ran without gradient tracking")
        ng = ctx.needs_input_grad
        if dout is None:
            dout = torch.empty_like(out)
            H.fill_zero(dout)
        sinks = [H.grad_sink(p, ng[3 + i]) for i, p in enumerate(params)]
        grads = [sinks[i] if sinks[i] is not None else
                 (torch.empty(p.shape, dtype=torch.float32, device=p.device) if ng[3 + i] else None)
                 for i, p in enumerate(params)]
        H.lstm_b1_bwd([p.detach() for p in params], grads, dout.contiguous().view(-1),
                      None if dhn is None else dhn.contiguous().view(dhn.shape[0], -1),
                      None if dcn is None else dcn.contiguous().view(dcn.shape[0], -1), xt, acts, h0c, c0c, hn, cn, out)
        for i, p in enumerate(params):
            if sinks[i] is not None:
                H.grad_done(p)
                grads[i] = None
        return (None, None, None, *grads)


class lstmnet(nn.Module):
    def __init__(self, num_channel=512, num_layer=2):
        super(lstmnet, self).__init__()
        self.lstm = nn.LSTM(num_channel, num_channel, num_layer)      # parameter container (keys / init as torch)
        self.tanh = nn.Tanh()
        self.num_channel = num_channel
        self.num_layer = num_layer
        self.lin = nn.Linear(512, 512)
        self.relu = nn.ReLU()

    def forward(self, input, hidden):
        # this hidden should be (h, c)
        T, B, _ = input.shape
        if hidden is None:
            if B != batch_size:
                raise RuntimeError(f"Expected hidden[0] size ({self.num_layer}, {B}, {self.num_channel}), "
                                   f"got [{self.num_layer}, {batch_size}, {self.num_channel}]")
            h0 = torch.zeros(self.num_layer, batch_size, self.num_channel, device=input.device)
            c0 = torch.zeros(self.num_layer, batch_size, self.num_channel, device=input.device)
        else:
            h0, c0 = hidden
        params = []
        for l in range(self.num_layer):
            params += [getattr(self.lstm, f"weight_ih_l{l}"), getattr(self.lstm, f"weight_hh_l{l}"),
                       getattr(self.lstm, f"bias_ih_l{l}"), getattr(self.lstm, f"bias_hh_l{l}")]
        params += [self.lin.weight, self.lin.bias]
        # one sample, one step, no gradient wanted for the input / state (the reference's own stepping): the fused path
        single = (T == 1 and B == 1 and B1_FUSED
                  and not (torch.is_grad_enabled() and (input.requires_grad or h0.requires_grad or c0.requires_grad)))
        out, hn, cn = (_LSTMNetB1Fn if single else _LSTMNetFn).apply(input, h0, c0, *params)
        return (out, (hn, cn))
