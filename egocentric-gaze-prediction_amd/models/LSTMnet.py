"""Mirror of the reference's ``models/LSTMnet.py``: the AT module's network
``tanh -> nn.LSTM(512, 512, num_layers=2) -> Linear(512, 512) -> ReLU`` (models/LSTMnet.py:15-37).

Same constructor, attributes (lstm, tanh, lin, relu, num_channel, num_layer) and state-dict keys
(``lstm.weight_ih_l0`` ... ``lin.bias``); ``forward`` runs one autograd node whose forward and BPTT backward
are sequences of C-ABI launches (f32-MFMA GEMMs + fused LSTM-cell kernels).  Quirk kept: ``hidden is None``
means zeros of batch **1** (module-global ``batch_size``, LSTMnet.py:13,30-31), so a batch > 1 without an
explicit ``(h, c)`` raises exactly like the reference.
"""
import torch
import torch.nn as nn

from .. import hipops as H

batch_size = 1


class _LSTMNetFn(torch.autograd.Function):
    """input (T,B,C), h0/c0 (L,B,H), then per layer (w_ih, w_hh, b_ih, b_hh), then lin.weight, lin.bias."""

    @staticmethod
    def forward(ctx, inp, h0, c0, *params):
        L = (len(params) - 2) // 4
        T, B, C = inp.shape
        Hd = params[1].shape[1]
        inp_c = H._req(inp.detach().contiguous(), "input")
        h0c, c0c = H._req(h0.detach().contiguous(), "h0"), H._req(c0.detach().contiguous(), "c0")
        x = H.tanh_fwd(inp_c)
        saved_layers = []
        layer_in = x.view(T * B, C)
        hn, cn = [], []
        for l in range(L):
            w_ih, w_hh, b_ih, b_hh = (p.detach() for p in params[4 * l:4 * l + 4])
            bsum = H.add(b_ih, b_hh)
            gx = H.linear_fwd(layer_in, w_ih, bias=bsum).view(T, B, 4 * Hd)      # all time steps at once
            hs = torch.empty((T, B, Hd), dtype=torch.float32, device=inp.device)
            cs = torch.empty_like(hs)
            acts = torch.empty((T, B, 4 * Hd), dtype=torch.float32, device=inp.device)
            h, c = h0c[l], c0c[l]
            for t in range(T):
                H.linear_fwd(h, w_hh, out=gx[t], accumulate=True)                 # gates += h W_hh^T
                H.lstm_cell_fwd(gx[t], c, hs[t], cs[t], acts[t])
                h, c = hs[t], cs[t]
            saved_layers.append((layer_in, hs, cs, acts))
            layer_in = hs.view(T * B, Hd)
            hn.append(h)
            cn.append(c)
        lin_w, lin_b = params[-2].detach(), params[-1].detach()
        out = H.linear_fwd(layer_in, lin_w, bias=lin_b, relu=True).view(T, B, lin_w.shape[0])
        ctx.saved = (x, h0c, c0c, saved_layers, out)
        ctx.params = [p.detach() for p in params]
        ctx.dims = (T, B, C, Hd, L)
        ctx.set_materialize_grads(False)
        return out, torch.stack(hn, 0), torch.stack(cn, 0)

    @staticmethod
    def backward(ctx, dout, dhn, dcn):
        x, h0c, c0c, saved_layers, out = ctx.saved
        params = ctx.params
        T, B, C, Hd, L = ctx.dims
        dev = out.device
        lin_w = params[-2]
        if dout is None:
            dout = torch.zeros_like(out)
        dpre = H.relu_bwd(out.view(T * B, -1), H._req(dout.contiguous().view(T * B, -1), "grad"))
        grads = [None] * len(params)
        h_top = saved_layers[-1][1].view(T * B, Hd)
        grads[-2] = H.matmul_tn(dpre, h_top)                      # d lin.weight = dpre^T h
        grads[-1] = H.colsum(dpre)
        dh_all = H.matmul_nn(dpre, lin_w).view(T, B, Hd)          # gradient into the top layer's outputs
        dh0 = torch.zeros((L, B, Hd), dtype=torch.float32, device=dev)
        dc0 = torch.zeros((L, B, Hd), dtype=torch.float32, device=dev)
        for l in reversed(range(L)):
            layer_in, hs, cs, acts = saved_layers[l]
            w_ih, w_hh = params[4 * l], params[4 * l + 1]
            dgates = torch.empty((T, B, 4 * Hd), dtype=torch.float32, device=dev)
            dh_next = dhn[l].contiguous() if dhn is not None else None
            dc_next = dcn[l].contiguous() if dcn is not None else None
            for t in reversed(range(T)):
                dh = dh_all[t] if dh_next is None else H.add(dh_all[t], dh_next)
                c_prev = cs[t - 1] if t > 0 else c0c[l]
                dc_prev = torch.empty((B, Hd), dtype=torch.float32, device=dev)
                H.lstm_cell_bwd(acts[t], cs[t], c_prev, dh, dc_next, dgates[t], dc_prev)
                dh_next = H.matmul_nn(dgates[t], w_hh)           # [B,4H] @ [4H,H]
                dc_next = dc_prev
            dh0[l], dc0[l] = dh_next, dc_next
            dg2 = dgates.view(T * B, 4 * Hd)
            h_prev_all = torch.cat((h0c[l:l + 1], hs[:-1]), 0).view(T * B, Hd)
            grads[4 * l] = H.matmul_tn(dg2, layer_in)             # d W_ih
            grads[4 * l + 1] = H.matmul_tn(dg2, h_prev_all)       # d W_hh
            db = H.colsum(dg2)
            grads[4 * l + 2] = db
            grads[4 * l + 3] = db.clone()
            dh_all = H.matmul_nn(dg2, w_ih).view(T, B, -1)        # into the layer below / the tanh'd input
        dinp = H.tanh_bwd(x, dh_all.contiguous()).view(T, B, C)
        ng = ctx.needs_input_grad
        return (dinp if ng[0] else None, dh0 if ng[1] else None, dc0 if ng[2] else None,
                *[g if ng[3 + i] else None for i, g in enumerate(grads)])


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
        out, hn, cn = _LSTMNetFn.apply(input, h0, c0, *params)
        return (out, (hn, cn))
