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


# Bucketed optimizer tail (EGAZE_OVERLAP_ADAM, default on; single process, host step counter): the Adam kernel of a contiguous
# bucket of parameters and the rebuild of their packed weights are issued on a side stream as soon as the bucket's gradients
# are final (gradient-sink hooks, as dp.GradReducer does for the all-reduce) instead of as one 0.22 ms HBM-bound kernel plus
# ~70 pack launches between the backward pass and the next forward.  Element-wise identical to the single launch.
_OVERLAP = os.environ.get("EGAZE_OVERLAP_ADAM", "1") != "0"
# The overlapped tail under data parallelism (each bucket stepped on the side stream right behind its own all-reduce,
# _OverlappedTail.adopt).  Bit-identical to reduce-then-step (tests/test_hip_dp.py) but measured slightly SLOWER at world size 1
# over RCCL (33.33 vs 32.94 ms per step, no reducer: 31.26 -- the cost of the data-parallel path there is the collectives'
# own kernels and stream joins, not the optimizer tail), and not measurable here at N > 1: opt-in, EGAZE_DP_TAIL=1.
_DP_TAIL = os.environ.get("EGAZE_DP_TAIL", "0") != "0"


class _OverlappedTail:
    BUCKET_BYTES = 24 * 1024 * 1024

    def __init__(self, opt):
        self.opt = opt
        self.stream = None
        self.enabled = False
        self.reducer = None
        self.stats = {"in_backward": 0, "in_finish": 0}       # buckets stepped under the backward pass / in step()
        self._own_buckets()
        for i, p in enumerate(opt.params):
            p._egz_sink.hooks.append(self._make_hook(i))
        if getattr(opt, "_reducer", None) is not None:
            self.adopt(opt._reducer)

    def _own_buckets(self):
        opt = self.opt
        order = sorted(range(len(opt.params)), key=lambda i: opt.offsets[i], reverse=True)      # backward order
        self.buckets, self.bucket_of = [], {}
        cur, cur_end, cur_start = [], None, None
        for i in order:
            start, end = opt.offsets[i], opt.offsets[i] + (opt.params[i].numel() + 3) // 4 * 4
            if cur_end is None:
                cur_end = end
            cur_start = start
            cur.append(i)
            self.bucket_of[i] = len(self.buckets)
            if (cur_end - cur_start) * 4 >= self.BUCKET_BYTES:
                self.buckets.append((cur_start, cur_end, cur))
                cur, cur_end = [], None
        if cur:
            self.buckets.append((cur_start, cur_end, cur))
        self._reset()

    def adopt(self, reducer):
        """Data parallel: take over the gradient reducer's buckets (dp.GradReducer cuts them the same way, in backward order),
        so that a bucket is stepped right behind ITS all-reduce: the tail makes the reducer issue the bucket's collective as
        soon as the bucket's gradients are final, lets the side stream wait for that collective's handle, and runs the Adam
        slice (1 / world folded in) and the repacking there -- under the rest of the backward pass, as at world size 1.
        ``None``: back to the tail's own buckets."""
        self.reducer = reducer
        if reducer is None:
            self._own_buckets()
            return
        idx = [[] for _ in reducer.buckets]
        for i, b in reducer.bucket_of.items():
            idx[b].append(i)
        self.buckets = [(reducer.buckets[b][0], reducer.buckets[b][1], idx[b]) for b in range(len(reducer.buckets))]
        self.bucket_of = dict(reducer.bucket_of)
        self._reset()

    def _reset(self):
        self.fired = [False] * len(self.opt.params)
        self.pending = [len(b[2]) for b in self.buckets]
        self.done = [False] * len(self.buckets)

    def usable(self) -> bool:
        o = self.opt
        red = self.reducer
        # the only pre-step hook the tail can live with is the wait() of the reducer whose buckets it has adopted
        hooks_ok = ((_DP_TAIL and all(getattr(h, "__self__", None) is red for h in o.pre_step_hooks)) if red is not None
                    else not o.pre_step_hooks)
        return (self.enabled and hooks_ok and not o.capturable and streams.ENABLED and o.flat_p.is_cuda
                and not torch.cuda.is_current_stream_capturing())

    def _make_hook(self, i):
        def hook(_param, _producer=None):
            if self.fired[i] or not self.usable():
                return
            b = self.bucket_of[i]
            self.fired[i] = True
            self.pending[b] -= 1
            if self.pending[b] == 0 and not self.done[b]:
                self._launch(b, self.opt.step_count + 1, in_backward=True)
        return hook

    def _launch(self, b, step, in_backward=False):
        o = self.opt
        start, end, idx = self.buckets[b]
        self.done[b] = True
        self.stats["in_backward" if in_backward else "in_finish"] += 1
        if self.stream is None:
            self.stream = streams.side_stream("adam")
        handle = None
        if self.reducer is not None and in_backward:
            # (from finish() the reducer's wait() has already run: every collective is joined into the stepping stream)
            handle = self.reducer.ensure_launched(b)
        streams.join_all_into(self.stream, include_comm=False)      # every producer of the bucket's gradients, every reader of its weights
        with torch.cuda.stream(self.stream):
            if handle is not None:
                handle.wait()                                       # the side stream continues behind the bucket's all-reduce
            H.adam_step(o.flat_p, o.flat_g, o.flat_m, o.flat_v, o.lr, o.betas[0], o.betas[1], o.eps, step, o.grad_scale,
                        lo=start, hi=end)
            ps = [o.params[i] for i in idx]
            H.touch_params(ps)
            if _TAIL_MULTIPACK and not torch.cuda.is_current_stream_capturing():
                H.refresh_packings_multi(ps)         # one launch for the bucket's fragment-ordered packings
            else:
                H.refresh_packings(ps)

    def finish(self, step):
        for b in range(len(self.buckets)):
            if not self.done[b]:
                self._launch(b, step)
        streams.join_all_into_current()          # (includes the tail stream)
        self._reset()


# One-launch refresh of all split packings after the step (hipops.repack_params).  Measured SLOWER than the lazy
# per-layer packs (41.2 vs 40.3 ms/step): the single kernel sits alone at the end of the step, the 74 small lazy
# launches hide behind the other streams' kernels.  Opt-in for A/B runs.
_MULTIPACK = os.environ.get("EGAZE_MULTIPACK", "0") != "0"
# The bucketed tail's packings in one launch per bucket (hipops.refresh_packings_multi: 8 launches instead of ~80 per step).
# Bit-identical, and no faster (31.80 vs 31.80 ms per step, profiles/r03_ab_notes.txt): the tail stream's small launches
# already hide under the backward pass.  Opt-in: EGAZE_TAIL_MULTIPACK=1.
_TAIL_MULTIPACK = os.environ.get("EGAZE_TAIL_MULTIPACK", "0") != "0"


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
        self._tail = None                 # overlap_with_backward()

    def set_capturable(self, on: bool = True):
        """Keep the step counter on the device (egz_adam_step_dev), so that step() can sit inside a captured hipGraph.  Every
        replay advances the device counter; the owner of the graph adds the replays to ``step_count`` (note_replays)."""
        if on:
            # {completed steps, reserved}
            self.step_dev = torch.tensor([self.step_count, 0], dtype=torch.int32, device=self.flat_p.device)
        elif self.capturable:
            self.step_count = int(self.step_dev[0].item())
        self.capturable = on

    def overlap_with_backward(self, on: bool = True):
        """Opt-in for loops that call ``step()`` right after ``backward()`` (SP.trainSP, LF.trainLate, bench.py -- the
        reference's loops, SP.py:136-137): the Adam update of a bucket of parameters and the rebuild of their packed weights are
        issued on a side stream as soon as the bucket's gradients are final, i.e. DURING the backward pass (_OverlappedTail).
        Element-wise the same update, but parameters start changing before ``backward()`` returns -- so code that inspects
        parameters between ``backward()`` and ``step()``, or accumulates gradients over several backward passes, must not
        enable it.  With a gradient reducer attached (dp.attach) it is ignored unless EGAZE_DP_TAIL=1, in which case the tail
        adopts the reducer's buckets and steps each bucket behind its own all-reduce (_OverlappedTail.adopt); ignored while
        capturable."""
        if on and _OVERLAP:
            if self._tail is None:
                self._tail = _OverlappedTail(self)
            self._tail.enabled = True
        elif self._tail is not None:
            self._tail.enabled = False

    def note_replays(self, n: int = 1):
        """A captured graph containing step() was replayed n times: the host-side count follows the device counter."""
        self.step_count += n

    # -- torch.optim.Optimizer surface used by the drivers
    def zero_grad(self, set_to_none: bool = False, all_overwritten: bool = False):
        """``all_overwritten``: the caller guarantees that the coming backward pass writes EVERY parameter's gradient in full
        through the gradient sinks (the AT single-sample step: rank-1 weight gradients and bias gradients written by
        csrc/lstm_b1.hip) -- the flat gradient buffer then needs no zero fill, only a new generation."""
        if all_overwritten:
            pass
        elif os.environ.get('EGAZE_TORCH_ZERO') == '1':
            self.flat_g.zero_()
        else:
            H.fill_zero(self.flat_g)
        self.zero_gen += 1
        for p, o in zip(self.params, self.offsets):         # re-attach if something dropped the views
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    @torch.no_grad()
    def step(self):
        for hook in self.pre_step_hooks:
            hook()
        # gradients are written in place by kernels on several HIP streams (encoder / wgrad helper streams, streams.py);
        # autograd only orders the streams its own AccumulateGrad nodes ran on, so order all of them here
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:             # (a capture holds exactly the streams that forked from it; nothing else may be joined)
            streams.join_all_into_current()
        if self._tail is not None and self._tail.usable():
            # the buckets whose gradients were complete were already stepped (and their packings rebuilt) under the backward
            # pass; finish the rest and make this stream wait for the tail stream
            self.step_count += 1
            self._tail.finish(self.step_count)
            return
        self.step_count += 1
        if self.capturable:
            # the step counter lives on the device so that a captured step can be replayed (set_capturable)
            H.adam_step_dev(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.lr, self.betas[0], self.betas[1],
                            self.eps, self.step_dev, self.grad_scale)
        else:
            if capturing:
                raise RuntimeError("FusedAdam.step() inside a hipGraph capture needs set_capturable(True)")
            H.adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.lr, self.betas[0], self.betas[1],
                        self.eps, self.step_count, self.grad_scale)
        H.touch_params(self.params)
        if _MULTIPACK:
            H.repack_params(self.params)  # every cached split-half weight packing, one launch

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
