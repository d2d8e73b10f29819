"""HIP-stream level concurrency for the SP step (instead of a tracing compiler / graph rewriter).

The layer chain is serial, but two kinds of work are independent and are issued on separate HIP streams so that the
workgroup tail of one kernel (e.g. 784 tiles over 512 resident slots) is filled by the next kernel's blocks:
  * forward: the RGB encoder and the flow encoder (models/model_SP.py:36-37) -- and, since autograd replays a node on
    the stream its forward ran on, their backward passes too;
  * backward: the weight gradient and the data gradient of the same convolution (both consume dy).
Ordering is expressed with stream waits only (no host synchronisation).  EGAZE_STREAMS=0 disables it (A/B runs).
"""
import os

import torch

ENABLED = os.environ.get("EGAZE_STREAMS", "1") != "0"
_SIDE = {}
# Helper-stream kinds that exist ONCE per device instead of once per parent stream.  HIP multiplexes
# streams onto 4 hardware queues (GPU_MAX_HW_QUEUES; more is 17 % slower, profiles/r02_hw_queues_ab.txt) and streams that
# share a queue run FIFO.  One weight-gradient stream for both encoders keeps the step at main + encoder_t + wgrad + one
# more (the AT stream, the H2D copy stream or the RCCL comm stream): the step time is unchanged (35.4 vs 35.3 ms) and the
# prefetched H2D copy overlaps fully (fp32 loader: 36.2 vs 38.0 ms per step; profiles/r02_hw_queues_ab.txt).
_SHARED = {"wgrad"}


# (A low-priority helper stream and a high-priority main stream were tried in rounds 2 / 3: priority serialises instead of
# interleaving -- 42.7 vs 35.6 ms -- and the switches were removed.)


# torch.cuda.current_stream() / torch.cuda.stream(...) go through several Python layers (~10 / ~25 us per use); the SP step forks
# ~45 times and asks for the current stream ~150 times, and at the reference's default batch (8, gaze_full.py:45) the step is
# bound by the host issuing it (tools/host_issue.py).  The raw accessors cost ~0.3 us: Stream objects are looked up by raw
# handle (torch's streams come from a fixed pool: a handle always names the same stream), the current stream is switched with
# the C call the context manager ends up in.
_raw_get = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_set = getattr(torch._C, "_cuda_setStream", None)
_OBJS = {}


def current() -> torch.cuda.Stream:
    """torch.cuda.current_stream() of the current device, without the Python layers."""
    if _raw_get is None:
        return torch.cuda.current_stream()
    dev = torch._C._cuda_getDevice()
    key = (dev, _raw_get(dev))
    st = _OBJS.get(key)
    if st is None:
        st = _OBJS[key] = torch.cuda.current_stream()
    return st


def _set_current(st: torch.cuda.Stream) -> None:
    if _raw_set is not None:
        _raw_set(stream_id=st.stream_id, device_index=st.device_index, device_type=st.device_type)
    else:
        torch.cuda.set_stream(st)


def side_stream(kind: str) -> torch.cuda.Stream:
    """A persistent helper stream per (current stream, kind)."""
    cur = current()
    key = (cur.device.index, 0 if kind in _SHARED else cur.cuda_stream, kind)
    st = _SIDE.get(key)
    if st is None:
        st = torch.cuda.Stream(device=cur.device)
        _SIDE[key] = st
    return st


class fork:
    """``with fork('wgrad') as f: ...`` runs the body on a side stream that first waits for the current stream;
    ``f.join(*tensors)`` makes the current stream wait for it and hands the tensors over to the current stream."""

    def __init__(self, kind: str):
        self.enabled = ENABLED and torch.cuda.is_available()
        self.kind = kind

    def __enter__(self):
        if self.enabled:
            self.cur = current()
            self.side = side_stream(self.kind)
            self.side.wait_stream(self.cur)
            _set_current(self.side)
        return self

    def __exit__(self, *exc):
        if self.enabled:
            _set_current(self.cur)
        return False

    def join(self, *tensors):
        if self.enabled:
            self.cur.wait_stream(self.side)
            for t in tensors:
                if t is not None:
                    t.record_stream(self.cur)

    def detach(self, *inputs):
        """Leave the side stream running instead of joining it: its kernels only produce results nobody on the current
        stream reads (a weight gradient written straight into the optimizer's buffer -- the optimizer step and the
        gradient all-reduce order themselves after every helper stream).  ``inputs`` are the tensors the side-stream
        kernels read: the caching allocator must not hand their memory out again before those kernels have run."""
        if self.enabled:
            for t in inputs:
                if t is not None:
                    t.record_stream(self.side)
                    am = getattr(t, "_egz_absmax", None)         # the abs-max scalar a gradient tensor carries (hipops)
                    if am is not None:
                        am.record_stream(self.side)


def join_all_into(target: torch.cuda.Stream, include_comm: bool = True):
    """Make ``target`` wait for every helper stream created so far, the current stream and the default stream."""
    cur = current()
    seen = {target.cuda_stream}
    side = [st for key, st in _SIDE.items() if include_comm or key[2] != "comm"]
    for st in side + [cur, torch.cuda.default_stream(target.device)]:
        if st.device == target.device and st.cuda_stream not in seen:
            target.wait_stream(st)
            seen.add(st.cuda_stream)


def join_all_into_current(include_comm: bool = True):
    """Make the current stream wait for every helper stream created so far (the optimizer step: gradients are written in
    place by kernels on several streams).  ``include_comm=False`` leaves the collective stream out (the end-of-backward
    join: the gradient buckets still in flight are joined by the reducer's wait() in front of the optimizer step)."""
    if torch.cuda.is_available():
        join_all_into(current(), include_comm)


_COMM = {}


def comm_stream(device=None) -> torch.cuda.Stream:
    """One stream per device on which gradient buckets are handed to the collective library: it waits for the producers of
    a bucket, the compute streams do NOT wait for it (dp.GradReducer)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    st = _COMM.get(dev.index)
    if st is None:
        st = torch.cuda.Stream(device=dev)
        _COMM[dev.index] = st
        _SIDE[(dev.index, 0, "comm")] = st            # joined by join_all_* like any helper stream
    return st
