"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI, overlapped with backward.

The reference is single-GPU (gaze_full.py:37); BASELINE.json's north_star adds exactly one exchange step per
iteration -- the gradient all-reduce.  Design for MI355X's point-to-point xGMI fabric:
  * gradients already live in ONE flat fp32 buffer (optim.FusedAdam), so a bucket is a contiguous slice that
    is all-reduced IN PLACE -- no gather/scatter copies, few large messages (ring collectives are per-link
    bound: ~25 MB buckets keep each of the 7 links busy without serialising the tail);
  * buckets are cut walking the parameters in reverse registration order (decoder -> bn -> fusion -> encoders),
    i.e. the order backward produces them; a post-accumulate-grad hook counts a bucket down and launches its
    all-reduce asynchronously (RCCL runs on its own stream, ordered after the producing kernels), so
    communication hides behind the remaining backward kernels;
  * ``wait()`` (run by the optimizer right before the Adam kernel) joins the outstanding collectives; the
    1/world_size averaging is folded into the Adam kernel's ``grad_scale``.
BatchNorm statistics stay per rank (the reference has no SyncBN; the per-rank batch is the parity unit).
Works with any torch.distributed backend ('nccl' = RCCL on ROCm; 'gloo' in the CPU tests).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


_raw_stream_fn = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _raw_current_stream(dev_index):
    if _raw_stream_fn is not None:
        return _raw_stream_fn(dev_index if dev_index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


# (Round 5 measured the hand-over variants with environment switches -- equal / halving buckets, one stream wait per bucket /
# for the last bucket only, synchronous collectives on the comm stream, hooks without a collective: profiles/r05_ab_notes.txt --
# and resolved them: halving buckets, one wait under RCCL.  The switches are gone; ``geometric=False`` keeps equal buckets.)      # experiment switch of round 5 (tools/dp_world1.py); resolved below


class GradReducer:
    def __init__(self, flat_grad: torch.Tensor, params: Sequence[torch.nn.Parameter], offsets: Sequence[int],
                 bucket_bytes: int = 25 * 1024 * 1024, group=None, flat_param: Optional[torch.Tensor] = None,
                 force: Optional[bool] = None, record_events: bool = False, geometric: bool = True):
        """``geometric``: bucket sizes halve along the backward pass (see below; False: equal buckets of ``bucket_bytes``).
        ``force``: install the hooks and issue the collectives even at world size 1 (default: EGAZE_DP_FORCE=1) -- a
        1-rank RCCL all-reduce is legal and runs the same code path (comm stream, async handles, the joins in front of
        the optimizer step), which is how the ``nccl`` path is exercised on a one-GPU box (tests/test_hip_rccl.py,
        ``bench.py`` extra.rccl_world1).  ``record_events``: keep a HIP event per bucket launch (tests)."""
        self.flat_grad = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if force is None:
            force = os.environ.get("EGAZE_DP_FORCE", "0") == "1"
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        self.record_events = record_events
        self.events = []                            # (bucket, event recorded on the comm stream behind its collective)
        self.stats = {"launched_in_backward": 0, "launched_in_wait": 0, "steps": 0}
        self._in_wait = False
        self.buckets: List[List[int]] = []          # [start, end, n_params]
        self.bucket_of = {}
        self._pending: List[int] = []
        self._handles = []
        self._launched: List[bool] = []
        order = sorted(range(len(params)), key=lambda i: offsets[i], reverse=True)
        cur_end, cur_start, cur_n = None, None, 0
        # Bucket sizes halve along the backward pass: a bucket closes when it holds max(bucket_bytes, half of the gradient bytes
        # still to come).  Every bucket hand-over costs event records / stream waits on the compute streams (~35 of each per SP
        # step with seven 25 MB buckets: +0.5 ms per step at world size 1, profiles/r05_dp_world1.txt), so few buckets are
        # cheaper -- but what the step waits for at its end is the LAST bucket's collective, so that one stays small: 186 MB of
        # SP gradients become 92 + 47 + 27 + 11 MiB (four buckets; tests/test_dp_gloo.py pins the cut) instead of 7 x 25 + 11.
        remaining = sum((p.numel() + 3) // 4 * 4 for p in params) * 4
        target = max(bucket_bytes, remaining // 2) if geometric else bucket_bytes
        for i in order:
            start, end = offsets[i], offsets[i] + (params[i].numel() + 3) // 4 * 4
            if cur_end is None:
                cur_end = end
            cur_start = start
            cur_n += 1
            self.bucket_of[i] = len(self.buckets)
            if (cur_end - cur_start) * 4 >= target:
                self.buckets.append([cur_start, cur_end, cur_n])
                remaining -= (cur_end - cur_start) * 4
                target = max(bucket_bytes, remaining // 2) if geometric else bucket_bytes
                cur_end, cur_n = None, 0
        if cur_n:
            self.buckets.append([cur_start, cur_end, cur_n])
        self._stream_objs = {}
        self._dev_index = flat_grad.device.index if flat_grad.is_cuda else None
        self._reset()
        self._hooks = []
        if self.active:
            if flat_param is not None:
                dist.broadcast(flat_param, src=0, group=group)       # identical replicas to start from
            for i, p in enumerate(params):
                hook = self._make_hook(i)
                self._hooks.append(p.register_post_accumulate_grad_hook(hook))      # gradient came through autograd
                sink = getattr(p, "_egz_sink", None)
                if sink is not None:
                    sink.hooks.append(hook)                                         # ... or was written in place

    def _reset(self):
        self._fired = [False] * len(self.bucket_of)
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._producers = [dict() for _ in self.buckets]      # bucket -> {stream id: stream} that wrote gradients of the bucket
        self._handles = []
        if getattr(self, "events", None):
            self.last_events, self.events = self.events, []      # the finished step's bucket events (record_events)

    def _launch(self, b: int):
        start, end, _ = self.buckets[b]
        self._launched[b] = True
        self.stats["launched_in_wait" if self._in_wait else "launched_in_backward"] += 1
        if self.flat_grad.is_cuda:
            # A bucket holds gradients written on several HIP streams (the two encoders, the detached weight-gradient
            # streams: streams.py).  The collective is issued from a dedicated comm stream that waits for ALL of them; the
            # compute streams themselves are not held up (joining them here would serialise the weight-gradient streams
            # with the data-gradient chain once per bucket).  The library orders the collective after the stream it is
            # issued from; wait() orders the optimizer step after the collective.
            # Only the streams that WROTE this bucket are joined (each parameter's hook runs on the stream its gradient kernels
            # were issued on): round 3 joined every helper stream of the process, which tied an unrelated model training on
            # its own stream (bench.py's AT step, a second optimizer) into every bucket and cost +1.1 ms per step at world 1
            # before a single collective ran (profiles/r04_dp_world1.txt).
            from . import streams
            comm = streams.comm_stream(self.flat_grad.device)
            cur = torch.cuda.current_stream()
            producers = dict(self._producers[b])
            producers[cur.cuda_stream] = cur
            for sid, st in producers.items():
                if sid == comm.cuda_stream:
                    continue
                if st is None:                    # noted by its raw handle only: the Stream object, once per distinct stream
                    st = self._stream_objs.get(sid)
                    if st is None:
                        st = self._stream_objs[sid] = torch.cuda.ExternalStream(sid, device=self.flat_grad.device)
                comm.wait_stream(st)
            with torch.cuda.stream(comm):
                self._handles.append(dist.all_reduce(self.flat_grad[start:end], op=dist.ReduceOp.SUM, group=self.group,
                                                     async_op=True))
                if self.record_events:
                    # the library runs the collective on its own stream, ordered after `comm`; the handle's wait() orders a
                    # stream after the collective -- make `comm` wait and mark that point
                    self._handles[-1].wait()
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(comm)
                    self.events.append((b, ev))
            return
        self._handles.append(dist.all_reduce(self.flat_grad[start:end], op=dist.ReduceOp.SUM, group=self.group,
                                             async_op=True))

    def _make_hook(self, i: int):
        b = self.bucket_of[i]

        def hook(_param, producer=None):
            # a parameter can report twice per backward: from its gradient sink (in-place write, hipops.GradSink) and
            # again from autograd, which runs the post-accumulate hook even when the node returned None for it
            # (observed on torch 2.10) -- the first report is the one that follows the gradient kernels
            if self._fired[i]:
                return
            self._fired[i] = True
            if self.flat_grad.is_cuda:
                # The producing streams are noted by their RAW handles: torch.cuda.current_stream() builds a Stream object
                # through several Python layers (~10 us per call; 215 parameters report per SP step = 2 ms of host time inside
                # backward -- what the data-parallel path cost at world 1 in round 4, profiles/r05_dp_world1.txt).  A Stream
                # object is looked up (once per distinct stream) only when a bucket is launched.
                prod = self._producers[b]
                raw = _raw_current_stream(self._dev_index)
                if raw not in prod:
                    prod[raw] = None
                if producer is not None and producer.cuda_stream not in prod:
                    prod[producer.cuda_stream] = producer
            self._pending[b] -= 1
            if self._pending[b] == 0 and not self._launched[b]:
                self._launch(b)
        hook._egz_reducer = self
        return hook

    def wait(self):
        """Join all bucket all-reduces of this step (launching any whose parameters got no gradient)."""
        if not self.active:
            return
        self._in_wait = True
        try:
            for b in range(len(self.buckets)):
                if not self._launched[b]:
                    self._launch(b)
        finally:
            self._in_wait = False
        if self.flat_grad.is_cuda and dist.get_backend(self.group) == "nccl":
            # RCCL runs a process group's collectives of one device in issue order on ONE internal stream: ordering the
            # optimizer's stream behind the LAST bucket orders it behind all of them (one stream wait instead of one per bucket)
            if self._handles:
                self._handles[-1].wait()
        else:
            for h in self._handles:
                h.wait()
        self.stats["steps"] += 1
        self._reset()

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def detach(self, optimizer=None):
        """Remove the hooks again (bench.py's untimed RCCL leg attaches a reducer to the live optimizer and takes it off)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in getattr(optimizer, "params", []):
            sink = getattr(p, "_egz_sink", None)
            if sink is not None:
                sink.hooks = [k for k in sink.hooks if getattr(k, "_egz_reducer", None) is not self]
        if optimizer is not None and self.wait in optimizer.pre_step_hooks:
            optimizer.pre_step_hooks.remove(self.wait)
            optimizer.grad_scale = 1.0
        if optimizer is not None and getattr(optimizer, "_reducer", None) is self:
            optimizer._reducer = None
        self.active = False


def attach(optimizer, bucket_bytes: int = 25 * 1024 * 1024, group=None, force: Optional[bool] = None,
           record_events: bool = False, geometric: bool = True) -> GradReducer:
    """Wire a GradReducer to a FusedAdam: reduce before the step, average inside the Adam kernel."""
    red = GradReducer(optimizer.flat_g, optimizer.params, optimizer.offsets, bucket_bytes, group, optimizer.flat_p,
                      force=force, record_events=record_events, geometric=geometric)
    optimizer.pre_step_hooks.append(red.wait)
    optimizer.grad_scale = red.grad_scale
    if red.active:
        from . import hipops
        hipops.bump_weight_epoch()          # parameters were overwritten by the broadcast
        optimizer._reducer = red
    return red


# ----------------------------------------------------------------------------- rank helpers for the drivers
def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_main() -> bool:
    """Rank 0 writes checkpoints / plots / extracted files (reference sites SP.py:205-208, LF.py:144-153, AT.py:189-195)."""
    return rank() == 0


def barrier():
    if world_size() > 1:
        dist.barrier()


class RankShardSampler(torch.utils.data.Sampler):
    """Shards a dataset over the ranks: rank r takes indices r, r+world, ... of a (per-epoch seeded) permutation.

    The reference's loaders are ``DataLoader(shuffle=True)`` (SP.py:34, LF.py:69) on one GPU; with one process per GPU
    every rank must see a disjoint 1/world share and -- because the gradient all-reduce is a collective -- the SAME
    number of minibatches, so the permutation is padded by wrapping around to a multiple of world * batch_size
    (``drop_last=False`` semantics; at world == 1 nothing is padded and the order is a plain shuffle).  Validation
    loops hold no collective, so they shard with ``pad=False``: no sample is counted twice."""

    def __init__(self, dataset, shuffle: bool, batch_size: int = 1, seed: int = 0, pad: bool = True,
                 world: Optional[int] = None, rank_: Optional[int] = None):
        self.n = len(dataset)
        self.shuffle, self.seed, self.epoch = shuffle, seed, 0
        self.world = world_size() if world is None else world
        self.rank = rank() if rank_ is None else rank_
        # pad=False (validation: no collective inside the loop): plain stride, ranks may differ by one sample
        unit = self.world * max(1, batch_size) if (self.world > 1 and pad) else 1
        self.total = (self.n + unit - 1) // unit * unit
        self.num_samples = len(range(self.rank, self.total, self.world))

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        if self.n and self.total > self.n:
            order = (order * (self.total // self.n + 1))[:self.total]
        return iter(order[self.rank:self.total:self.world])


def reduce_meters(*pairs):
    """[(sum, count), ...] summed over the ranks -> list of global averages: every rank then takes the same
    'is this the best epoch' decision (reference: SP.py:203-208, LF.py:141-153)."""
    vals = [float(v) for pr in pairs for v in pr]
    if world_size() > 1:
        t = torch.tensor(vals, dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t)
        vals = t.tolist()
    return [vals[2 * i] / max(vals[2 * i + 1], 1e-30) for i in range(len(pairs))]


def sync_buffers(module: torch.nn.Module):
    """Rank 0's BatchNorm running statistics to every rank (training keeps them per rank like the single-GPU reference
    would; validation and checkpoints use rank 0's, so all ranks evaluate the same model)."""
    if world_size() > 1:
        for b in module.buffers():
            if b.is_cuda and dist.get_backend() != "nccl":
                h = b.detach().cpu()
                dist.broadcast(h, src=0)
                b.copy_(h)
            else:
                dist.broadcast(b, src=0)
