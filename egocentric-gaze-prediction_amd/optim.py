"""Fused Adam over flat fp32 buffers -- the optimizer the drivers use where the reference calls
``torch.optim.Adam(params, lr=...)`` (SP.py:110-113, AT.py:84, LF.py:77; torch defaults: betas
(0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).

On construction every trainable parameter is re-homed into ONE flat buffer (``p.data`` becomes a view)
and gets a persistent ``.grad`` view into a flat gradient buffer, so that
  * one ``egz_adam_step`` launch updates the whole model (7 fp32 streams, HBM-bound),
  * ``zero_grad`` is one memset,
  * data-parallel training all-reduces contiguous buckets of the flat gradient in place (dp.py).
``state_dict()`` / ``load_state_dict()`` use torch.optim.Adam's format so SP checkpoints
({'optimizer': ...}, SP.py:205-208, resume mode '2' SP.py:40-50,114-115) stay interchangeable.
"""
from __future__ import annotations

from typing import Iterable, List

import os

import torch

from . import hipops as H
from . import streams


# (Rounds 2 / 3 carried a bucketed "overlapped tail" -- Adam slices and weight repacking issued under the backward pass as
# gradient buckets became final -- and one-launch multi-tensor repacks.  Bit-identical, and worth nothing measurable any more
# (30.56 vs 30.60 ms per step with / without, profiles/r04_dp_world1.txt; the repack variants: profiles/r03_ab_notes.txt):
# removed in round 4.  The step is one Adam launch; packed weights are rebuilt lazily by the first launch that needs them.)


class FusedAdam:
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        params = list(params)
        if params and isinstance(params[0], dict):           # torch-style param groups (SP.py:110)
            if len(params) != 1:
                raise NotImplementedError("the reference uses a single parameter group")
            group = params[0]
            lr = group.get("lr", lr)
            params = list(group["params"])
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam needs parameters on the HIP device (move the model first)")
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False)
        self.lr, self.betas, self.eps = lr, tuple(betas), eps
        self.step_count = 0
        self.grad_scale = 1.0
        # 16-byte aligned slots so every parameter starts on a float4 boundary
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.zero_gen = 0                 # zero_grad() generation: hipops.GradSink lets ONE in-place write through per generation
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                p.grad = self.flat_g[o:o + n].view(p.shape)
                # backward kernels write this view directly instead of returning a tensor for AccumulateGrad to add
                p._egz_sink = H.GradSink(self.flat_g[o:o + n], self)
        H.bump_weight_epoch()
        self.pre_step_hooks = []          # dp.GradReducer registers its wait() here
        self.capturable = False
        self.step_dev = None
        # bit 0 set by the Adam kernel when it skipped an element whose gradient was NaN / inf (csrc/adam.hip): check_finite()
        self.nonfinite = torch.zeros(1, dtype=torch.int32, device=dev)

    def set_capturable(self, on: bool = True):
        """Keep the step counter on the device (egz_adam_step_dev), so that step() can sit inside a captured hipGraph.  Every
        replay advances the device counter; the owner of the graph adds the replays to ``step_count`` (note_replays)."""
        if on:
            # {completed steps, reserved}
            self.step_dev = torch.tensor([self.step_count, 0], dtype=torch.int32, device=self.flat_p.device)
        elif self.capturable:
            self.step_count = int(self.step_dev[0].item())
        self.capturable = on

    def check_finite(self, reset: bool = True) -> None:
        """Raise if a step since the last check met NaN / inf gradients (synchronises: call where the loop reads back anyway).
        The kernel skipped those elements -- the weights are the ones from before the poisoned step(s), not NaN."""
        if int(self.nonfinite.item()):
            if reset:
                self.nonfinite.zero_()
            raise FloatingPointError("FusedAdam: a step since the last check saw NaN / inf gradient elements and skipped them "
                                     "(parameters untouched there); the backward pass that produced them is broken -- e.g. a "
                                     "persistent-LSTM hand-off that timed out (hipops.lstm_persist_check)")

    def note_replays(self, n: int = 1):
        """A captured graph containing step() was replayed n times: the host-side count follows the device counter."""
        self.step_count += n

    # -- torch.optim.Optimizer surface used by the drivers
    def zero_grad(self, set_to_none: bool = False, all_overwritten: bool = False):
        """``all_overwritten``: the caller guarantees that the coming backward pass writes EVERY parameter's gradient in full
        through the gradient sinks (the AT single-sample step: rank-1 weight gradients and bias gradients written by
        csrc/lstm_b1.hip) -- the flat gradient buffer then needs no zero fill, only a new generation."""
        from . import functions as _F
        _F._JOIN_QUEUED[0] = False      # a backward pass that raised never ran its end-of-backward callback: re-arm it
        if all_overwritten:
            pass
        else:
            H.fill_zero(self.flat_g)
        self.zero_gen += 1
        for p, o in zip(self.params, self.offsets):         # re-attach if something dropped the views
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    @torch.no_grad()
    def step(self):
        from . import functions as _F
        _F._JOIN_QUEUED[0] = False
        for hook in self.pre_step_hooks:
            hook()
        # gradients are written in place by kernels on several HIP streams (encoder / wgrad helper streams, streams.py);
        # autograd only orders the streams its own AccumulateGrad nodes ran on, so order all of them here
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:             # (a capture holds exactly the streams that forked from it; nothing else may be joined)
            streams.join_all_into_current()
        self.step_count += 1
        if self.capturable:
            # the step counter lives on the device so that a captured step can be replayed (set_capturable)
            H.adam_step_dev(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.lr, self.betas[0], self.betas[1],
                            self.eps, self.step_dev, self.grad_scale, nonfinite=self.nonfinite)
        else:
            if capturing:
                raise RuntimeError("FusedAdam.step() inside a hipGraph capture needs set_capturable(True)")
            H.adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.lr, self.betas[0], self.betas[1],
                        self.eps, self.step_count, self.grad_scale, nonfinite=self.nonfinite)
        H.touch_params(self.params)
        H.refresh_packings(self.params)     # one launch for every packing this step made stale (the next forward finds them fresh)

    @property
    def param_groups(self):
        # the keys torch.optim.Adam groups carry (so that the state dict loads into the reference's optimizer unchanged)
        return [dict(self.defaults, lr=self.lr, maximize=False, foreach=None, capturable=False, differentiable=False,
                     fused=None, decoupled_weight_decay=False, params=list(range(len(self.params))))]

    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.flat_m[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.flat_v[o:o + n].view(p.shape).clone()}
        return {"state": state, "param_groups": self.param_groups}

    def load_state_dict(self, sd):
        groups = sd.get("param_groups", [])
        if groups and "params" in groups[0] and len(groups[0]["params"]) != len(self.params):
            # same failure torch.optim.Optimizer.load_state_dict reports (e.g. an all-parameter checkpoint of
            # `--sp_resume 0` loaded into the fusion+bn+decoder optimizer of resume mode '2', SP.py:109-115)
            raise ValueError("loaded state dict contains a parameter group that doesn't match the size of "
                             "optimizer's group")
        if groups:
            self.lr = groups[0].get("lr", self.lr)
            self.betas = tuple(groups[0].get("betas", self.betas))
            self.eps = groups[0].get("eps", self.eps)
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                st = sd["state"].get(i)
                if st is None:
                    continue
                n = p.numel()
                self.flat_m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self.flat_v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_count = int(float(st["step"]))
        if self.capturable:
            self.step_dev[0] = self.step_count


Adam = FusedAdam
