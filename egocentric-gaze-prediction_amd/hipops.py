"""Tensor-level wrappers over the C-ABI (``include/egaze_hip.h``).

PyTorch is plumbing here: it owns device memory and the HIP stream; every wrapper passes raw
``data_ptr()``s, explicit shapes and ``torch.cuda.current_stream()`` to ``libegaze_hip.so``.
Activations are NHWC fp32 (``(B, H, W, C)`` contiguous tensors); weights keep the reference layout
``(Cout, Cin, 3, 3)`` and are re-packed on the device when they change.
All wrappers raise on CPU tensors -- there is no CPU fallback.
"""
from __future__ import annotations

import weakref
from typing import Optional, Tuple

import torch

from ._lib import LIB as _RAW_LIB, check

EPI_BIAS, EPI_BIAS_RELU, EPI_BIAS_STATS = 0, 1, 2
import os as _os


class _Profiler:
    """Optional per-entry-point timing with HIP events recorded on the launch stream (the stream every
    C-ABI call is issued on is torch's current stream, so torch.cuda.Event brackets exactly those kernels).
    Used by bench.py's roofline leg; off by default (zero overhead beyond one attribute test)."""

    def __init__(self):
        self.enabled = False
        self.events = {}
        self.flops = {}

    def start(self):
        self.events, self.flops, self.enabled = {}, {}, True

    def stop(self):
        """-> {entry point: {'calls', 'ms', 'flops'}}"""
        self.enabled = False
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.events.items():
            out[name] = {"calls": len(evs), "ms": sum(a.elapsed_time(b) for a, b in evs),
                         "flops": self.flops.get(name, 0.0)}
        for name, v in self.flops.items():      # FLOPs noted under a family name none of whose own launches ran in this step
            out.setdefault(name, {"calls": 0, "ms": 0.0, "flops": v})
        return out

    def note_flops(self, name, v):
        if self.enabled:
            self.flops[name] = self.flops.get(name, 0.0) + v


PROF = _Profiler()


class _LibProxy:
    def __init__(self, raw):
        self._raw = raw

    def __getattr__(self, name):
        fn = getattr(self._raw, name)

        def call(*a):
            if not PROF.enabled or name.endswith(("_bytes", "_rows", "_elems", "_ok", "_splits")):   # host-side queries launch nothing
                return fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            PROF.events.setdefault(name, []).append((e0, e1))
            return rc
        setattr(self, name, call)
        return call


LIB = _LibProxy(_RAW_LIB)


# Every C-ABI launch asks for the current stream: torch.cuda.current_stream() builds a Stream object through several Python
# layers (~10 us per call, measured: 7 calls = 67 us of a 436 us AT sample step); the raw-handle accessor costs ~0.3 us.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a HIP ('cuda') tensor -- this package has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous buffer, got strides {t.stride()} for {tuple(t.shape)}")
    return t


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _out(out: Optional[torch.Tensor], shape, device) -> torch.Tensor:
    """A kernel output: a fresh tensor, or the caller's destination (e.g. a gradient sink) viewed with that shape."""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    n = 1
    for d in shape:
        n *= d
    if out.numel() != n or out.dtype != torch.float32 or not out.is_contiguous() or not out.is_cuda:
        raise RuntimeError(f"output buffer {tuple(out.shape)} / {out.dtype} does not fit {tuple(shape)} fp32 contiguous")
    return out.view(shape)


# ----------------------------------------------------------------------------- workspace
_WS = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Stream-ordered scratch (one buffer per device AND stream, grown on demand; the C-ABI never allocates).
    Keyed by stream because kernels on different HIP streams run concurrently (streams.py)."""
    key = (torch.device(device).index or 0, _stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# ----------------------------------------------------------------------------- packed weights
_WEIGHT_EPOCH = [0]
_PACKED = {}
_USED = set()                     # packings requested by a launch since their last rebuild (refresh_packings)


def bump_weight_epoch():
    """Invalidate every packing (parameters were overwritten wholesale behind torch's back, e.g. a DP broadcast)."""
    _WEIGHT_EPOCH[0] += 1


def touch_params(params):
    """Called by the fused optimizer after it rewrote THESE parameters through its flat buffer (no torch version bump):
    only their packings go stale -- a second optimizer (the AT module's) must not invalidate the SP model's."""
    for p in params:
        p._egz_epoch = getattr(p, "_egz_epoch", 0) + 1


def _tag(w):
    return (_WEIGHT_EPOCH[0], getattr(w, "_egz_epoch", 0), w._version, w.data_ptr())


_UID = [0]


def _uid(w: torch.Tensor) -> int:
    """Identity of a weight tensor that cannot be recycled the way a data_ptr can: a counter stored on the
    tensor object (torch preserves the Python object of a live tensor); the cache entry dies with the tensor."""
    uid = getattr(w, "_egz_uid", None)
    if uid is None:
        _UID[0] += 1
        uid = _UID[0]
        w._egz_uid = uid
        weakref.finalize(w, _drop_packed, uid)
    return uid


_PACK_FN = {"fwd": ("egz_pack_w3x3_fwd", 0, 0), "dgrad": ("egz_pack_w3x3_dgrad", 0, 1),
            "ups_fwd": ("egz_pack_w3x3_ups_fwd", 1, 2), "ups_dgrad": ("egz_pack_w3x3_ups_dgrad", 1, 3),
            # MFMA-fragment-ordered split packings of the streamed-weight kernel (split dtypes only)
            "fwd_frag": (None, 0, 4), "dgrad_frag": (None, 0, 5), "ups_dgrad_frag": (None, 1, 6),
            "ups_fwd_frag": (None, 1, 7)}

# Arithmetic of the wide convolutions (GEMM output channels % 128 == 0):
#   "split" (default) error-compensated split-half operands on the 16-bit MFMA path: f16 x3 (22 significant bits;
#           measured err 5e-7 of max|ref| per layer vs 8.6e-7 for the exact-f32 MFMA kernel) in the forward pass, the
#           data gradient and the weight gradient (GRAD_SPLIT below) -- conv3x3_igemm_x3.hip, conv3x3_wgrad.hip.
#   "f32"   exact-f32 MFMA everywhere (v_mfma_f32_32x32x2_f32); EGAZE_PRECISION=f32 selects it.
PRECISION = _os.environ.get("EGAZE_PRECISION", "split")


def precision_banner() -> str:
    """One line for the drivers' logs: which arithmetic the convolutions run in (an environment default, so say it)."""
    if PRECISION != "split":
        return "egaze-hip arithmetic: exact-f32 MFMA (EGAZE_PRECISION=f32)"
    g = "f16x3, 22 bits, abs-max scaled" if GRAD_SPLIT == "f16" else "bf16x3, 16 bits"
    return (f"egaze-hip arithmetic: split-half MFMA -- forward f16x3 (22 significant bits per operand), gradients {g}; fp32 "
            f"accumulate; EGAZE_PRECISION=f32 selects exact-f32 MFMA")
# Operand type of the split-half GRADIENT kernels (data and weight gradients):
#   "f16" (default) f16 x3, 22 significant bits (3e-7 per layer, the exact-f32 kernels' own level -- the reference's
#          arithmetic class): the gradient producers (BN / ReLU / fusion backward) also emit max |dy| and the conv backward
#          scales dy by the matching power of two so that gradients of 1e-3 ... 1e-13 stay representable;
#   "bf16" bf16 x3, 16 significant bits (5e-6 per layer), fp32 exponent range, no scaling: ~3 % faster per step, opt-in.
GRAD_SPLIT = _os.environ.get("EGAZE_GRAD_SPLIT", "f16")
F32, F16X3, BF16X3 = 0, 1, 2


_SPLIT_MAX_BYTES = (1 << 32) - (1 << 24)      # the split kernels fetch through 32-bit buffer offsets


def conv_dtype(role: str, gemm_out: int, gemm_in: int, operand: Optional[torch.Tensor] = None) -> int:
    """dtype code for a conv launch under the current PRECISION policy (role: 'fwd' | 'dgrad').  ``operand`` = the gathered
    activation / gradient tensor: one of 4 GiB or more (B >= 320 at 224 x 224 x 64) stays on the 64-bit-addressed f32 kernels."""
    if PRECISION != "split":
        return F32
    if operand is not None and operand.numel() * 4 >= _SPLIT_MAX_BYTES:
        return F32
    if gemm_out % 64 != 0 or gemm_in % 32 != 0:
        # narrow layers (the late-fusion stack: 32 -> 32 -> 8 channels): only the streamed-weight kernel's 32-column tile
        # takes them, and only for plain convs over an operand whose geometry it covers
        if not (STREAMED and operand is not None and operand.dim() == 4
                and LIB.egz_conv3x3_streamed_ok(operand.shape[0], operand.shape[1], operand.shape[2], gemm_in, gemm_out, 0)):
            return F32
    return F16X3 if (role == "fwd" or GRAD_SPLIT == "f16") else BF16X3


# Module-level switches of this file are plain constants unless they read the environment: the tests flip them with
# monkeypatch (each one names the test that exercises its off position); round 4 removed the environment knobs of everything
# that was measured and decided.
STREAMED = True           # plain convs on the streamed-weight halo kernel (False: the round-1 LDS-DMA halo / gather kernels)
UPSF_STREAMED = True      # forward of the four upsample-fused decoder convs on the streamed kernel (MODE UPSF)
# Split-K form of the streamed kernel when a launch has too few pixel tiles to fill the chip (egz_conv3x3_streamed_splits):
# batch-1 inference runs its 28 x 28 / 14 x 14 layers on 8-28 of 512 block slots otherwise.  (tests flip the constant)
# (tests: test_conv3x3_streamed_splitk, the whole-model gradient tests pin the summation order with it).
SPLITK = True


def conv_weight(w: torch.Tensor, role: str, dtype: int, x: torch.Tensor, gemm_out: int):
    """The packed weight a plain 3x3 conv launch over the NHWC operand ``x`` needs -> (packed buffer, streamed flag).
    Split-half launches whose geometry the streamed-weight kernel covers (egz_conv3x3_streamed_ok) use the MFMA-fragment-
    ordered packing; everything else the plane-ordered one.  role: 'fwd' | 'dgrad' | 'ups_dgrad' (x = the hi-res dy) |
    'ups_fwd' (x = the low-res input of [upsample x2 -> conv]; four phase convolutions, f16 x3 only)."""
    if dtype and STREAMED:
        B, H, W, C = x.shape
        if role == "ups_fwd":
            if UPSF_STREAMED and dtype == F16X3 and LIB.egz_conv3x3_streamed_ok(B, 2 * H, 2 * W, C, gemm_out, 2):
                return packed_weight(w, "ups_fwd_frag", dtype), True
        elif LIB.egz_conv3x3_streamed_ok(B, H, W, C, gemm_out, 1 if role == "ups_dgrad" else 0):
            return packed_weight(w, role + "_frag", dtype), True
    return packed_weight(w, role, dtype), False


def _drop_packed(uid: int):
    for kind in _PACK_FN:
        for dt in (0, 1, 2):
            _PACKED.pop((uid, kind, dt), None)


def packed_weight(w: torch.Tensor, kind: str, dtype: int = 0) -> torch.Tensor:
    """(Cout, Cin, [1,] 3, 3) -> the kernels' private packed layout: 'fwd' / 'dgrad' (9 taps) or, for a conv that
    follows a nearest x2 upsample, 'ups_fwd' (4 phases x 2x2 pre-summed taps) / 'ups_dgrad' (4x4 stride-2 taps).
    Pass the parameter object itself (not a detached alias) so the cache can follow its identity."""
    _req(w, "weight")
    K, C = w.shape[0], w.shape[1]
    key = (_uid(w), kind, dtype)
    tag = _tag(w)
    hit = _PACKED.get(key)
    if hit is not None and hit[0] == tag:
        _USED.add(key)
        return hit[1]
    fname, ekind, kidx = _PACK_FN[kind]
    buf = hit[1] if hit is not None else torch.empty(LIB.egz_pack_w3x3_elems(C, K, ekind), dtype=torch.float32,
                                                     device=w.device)
    if kidx >= 4:
        if not dtype:
            raise RuntimeError("fragment-ordered packings exist for the split-half dtypes only")
        check(LIB.egz_pack_w3x3_split_frag(w.data_ptr(), buf.data_ptr(), C, K, kidx, dtype, _stream()),
              "egz_pack_w3x3_split_frag")
    elif dtype:      # hi / lo 16-bit planes (same byte count as the fp32 packing)
        check(LIB.egz_pack_w3x3_split(w.data_ptr(), buf.data_ptr(), C, K, kidx, dtype, _stream()), "egz_pack_w3x3_split")
    else:
        check(getattr(LIB, fname)(w.data_ptr(), buf.data_ptr(), C, K, _stream()), fname)
    _PACKED[key] = (tag, buf, weakref.ref(w))
    _USED.add(key)
    return buf


BATCH_REPACK = True       # FusedAdam.step() rebuilds the stale fragment-ordered packings of its parameters in ONE launch
#                           (False: each packing is rebuilt lazily in front of the first launch that needs it;
#                           test_pack_frag_batch_matches_per_layer flips it)
_BATCH_TABLES = {}
_BATCH_PINNED = set()          # tables whose launch was captured into a hipGraph: never evicted


def refresh_packings(params, force: bool = False) -> int:
    """Rebuild, in one launch on the current stream, every fragment-ordered split packing of ``params`` that went stale and
    that some launch has used since its last rebuild (egz_pack_w3x3_frag_batch; bit-identical to the lazy per-layer rebuilds).
    Returns the number of packings rebuilt.  Packings nobody asked for since the last rebuild (eval-only kinds, a layer whose
    geometry changed) stay stale and are rebuilt lazily by packed_weight().  ``force``: every existing fragment-ordered packing
    of ``params``, stale by its tag or not (a captured training step whose parameters were overwritten between two replays)."""
    if not BATCH_REPACK or not (_USED or force):
        return 0
    ids = {getattr(p, "_egz_uid", None) for p in params}
    todo = []
    for key in sorted(k for k in (_PACKED if force else _USED) if k[0] in ids and k[2] and _PACK_FN[k[1]][2] >= 4):
        hit = _PACKED.get(key)
        w = hit[2]() if hit is not None else None
        if w is None:
            _USED.discard(key)
            continue
        if force or hit[0] != _tag(w):
            todo.append((key, w, hit[1]))
    if not todo:
        return 0
    sig = tuple((key, w.data_ptr(), buf.data_ptr()) for key, w, buf in todo)
    ent = _BATCH_TABLES.get(sig)
    capturing = torch.cuda.is_current_stream_capturing()
    if ent is None and capturing:
        # No table for THIS set inside a hipGraph capture (the warm-up step before the capture touched another set: an eval
        # pass in between, a table eviction): building one needs a host -> device copy, which a capture cannot hold.  The
        # per-key pack launches need no table and are capturable -- the captured step then repacks exactly this set on every
        # replay (returning 0 here left every replay after the first on stale packed weights).
        for key, w, buf in todo:
            check(LIB.egz_pack_w3x3_split_frag(w.data_ptr(), buf.data_ptr(), w.shape[1], w.shape[0], _PACK_FN[key[1]][2], key[2],
                                               _stream()), "egz_pack_w3x3_split_frag")
    else:
        if ent is None:
            rows, nb = [], 0
            for key, w, buf in todo:
                K, C = w.shape[0], w.shape[1]
                rows.append([w.data_ptr(), buf.data_ptr(), C, K, _PACK_FN[key[1]][2], key[2], nb, 0])
                nb += LIB.egz_pack_w3x3_frag_blocks(C, K)
            if len(_BATCH_TABLES) > 64:
                # (a table a captured graph replays from must stay alive: its launch baked the table's address in)
                for k_ in [k_ for k_ in _BATCH_TABLES if k_ not in _BATCH_PINNED]:
                    del _BATCH_TABLES[k_]
            ent = (torch.tensor(rows, dtype=torch.int64).to(todo[0][1].device), nb)
            _BATCH_TABLES[sig] = ent
        if capturing:
            _BATCH_PINNED.add(sig)
        check(LIB.egz_pack_w3x3_frag_batch(ent[0].data_ptr(), len(todo), ent[1], _stream()), "egz_pack_w3x3_frag_batch")
    for key, w, buf in todo:
        _PACKED[key] = (_tag(w), buf, weakref.ref(w))
        _USED.discard(key)
    return len(todo)


# ----------------------------------------------------------------------------- gradient sinks
class GradSink:
    """Where a parameter's gradient lands when the parameter lives in a fused optimizer's flat buffers (optim.FusedAdam):
    the backward kernels write ``p.grad`` (a view of the flat gradient buffer) DIRECTLY and the autograd node returns
    ``None`` for that input, so no stock ``AccumulateGrad`` add (one launch + 3 HBM passes per parameter, reference
    SP.py:136-138) runs in the step.  torch's accumulation semantics are kept: only the FIRST gradient of a
    ``zero_grad()`` generation is written in place; a second backward without ``zero_grad`` (or a weight used twice in one
    graph) finds the sink taken and goes through autograd's ordinary accumulation.  ``hooks`` fire after the in-place
    write (dp.GradReducer counts its buckets down there, as its post-accumulate-grad hook does on the autograd route)."""
    __slots__ = ("buf", "owner", "gen_written", "hooks")

    def __init__(self, buf, owner):
        self.buf, self.owner, self.gen_written, self.hooks = buf, owner, -1, []


def grad_sink(param, wanted: bool = True) -> Optional[torch.Tensor]:
    """The in-place gradient destination of ``param`` for this backward pass, or None (-> return the gradient to autograd).
    Taking the sink marks it used for the current zero_grad generation."""
    if not wanted:
        return None
    sk = getattr(param, "_egz_sink", None)
    if sk is None or not DIRECT_GRADS:
        return None
    gen = sk.owner.zero_gen
    if sk.gen_written == gen:
        return None
    sk.gen_written = gen
    return sk.buf


PENDING_PRODUCER = [None]       # helper stream a detached weight-gradient fork just left running (functions._close_fork)


def grad_done(param):
    """Run the sink hooks of ``param``: its gradient is final on the current stream -- or on the helper stream a detached
    weight-gradient fork left running (the hooks get that stream as their second argument)."""
    st, PENDING_PRODUCER[0] = PENDING_PRODUCER[0], None
    sk = getattr(param, "_egz_sink", None)
    if sk is not None:
        for h in sk.hooks:
            h(param, st)


DIRECT_GRADS = True      # gradient sinks on (False: every gradient goes back through autograd's AccumulateGrad; AT._GraphedSampleStep honours it)


# ----------------------------------------------------------------------------- abs-max of gradient tensors


class _AbsmaxArena:
    """Zero-filled abs-max buffers (egz_common.h: the producers fold their maxima in with atomic max, so a buffer must start at
    zero).  Buffers are cut from chunks that ONE fill zeroes (a chunk serves ~250 buffers = about one SP step); a chunk is
    filled on the stream that was current when it was created and every other stream that takes a buffer from it waits for
    that fill once.  A hipGraph capture gets ONE chunk of its own (CAPTURE_CHUNK buffers), created by ``capture()`` below on
    the capture's origin stream before any side stream forks: the captured fill re-zeroes it on every replay, buffers handed
    out before / after the capture are never touched by it, and a capture that needs more buffers than the chunk holds raises
    (a second chunk created mid-capture on a forked stream would not be ordered before the other streams' atomic maxima --
    ADVICE r4).  Captures must go through ``capture()``: take() inside a capture that did not raises."""
    CHUNK = 256
    CAPTURE_CHUNK = 1024

    def __init__(self):
        self.elems = 0
        self.chunk = None
        self.size = 0
        self.used = 0
        self.tag = None
        self.event = None
        self.waited = set()
        self.capture_seq = 0
        self.in_capture = False

    def capture_begin(self, device, eager: bool = True):
        """First statement inside a hipGraph capture (``capture()``): the capture's buffers come from a chunk of its own whose
        fill is captured HERE, on the capture's origin stream, before any side stream forks.  ``eager=False``: a capture that
        stays on one stream and may not take a buffer at all (the AT per-sample step: its replay would zero the chunk for
        nothing, one launch per sample) -- the chunk is then created by the first take() inside the capture."""
        self.capture_seq += 1
        self.chunk = None
        self.in_capture = True
        if eager:
            self.take(device)

    def capture_end(self):
        self.in_capture = False
        self.chunk = None                 # the next eager take() starts a chunk of its own

    def take(self, device) -> torch.Tensor:
        if not self.elems:
            self.elems = int(LIB.egz_absmax_elems())
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing and not self.in_capture:
            raise RuntimeError("abs-max arena: a hipGraph capture that did not go through hipops.capture() asked for a buffer "
                               "(it would share the previous capture's chunk)")
        tag = (capturing, self.capture_seq if capturing else 0, str(device))
        if self.chunk is None or self.used == self.size or tag != self.tag:
            if capturing and self.chunk is not None and tag == self.tag:
                raise RuntimeError(f"abs-max arena: the capture needs more than {self.size} abs-max buffers "
                                   "(raise _AbsmaxArena.CAPTURE_CHUNK)")
            self.size = self.CAPTURE_CHUNK if capturing else self.CHUNK
            self.chunk = torch.zeros(self.size * self.elems, dtype=torch.int32, device=device)
            self.used, self.tag = 0, tag
            if capturing:
                self.event = None                    # inside a capture the forked streams are ordered by the capture graph itself
            else:
                self.event = torch.cuda.Event()
                self.event.record()
            self.waited = {_stream()}
        elif self.event is not None:
            raw = _stream()                           # (the raw handle: ~0.3 us; a Stream object only when this stream has to wait)
            if raw not in self.waited:
                torch.cuda.current_stream().wait_event(self.event)
                self.waited.add(raw)
        buf = self.chunk[self.used * self.elems:(self.used + 1) * self.elems]
        self.used += 1
        return buf


ABSMAX_ARENA = _AbsmaxArena()


class capture:
    """``with hipops.capture(graph[, eager_arena=False]): ...`` = ``torch.cuda.graph(graph)`` plus what this package's launches
    need inside a capture: the abs-max arena's per-capture chunk (see _AbsmaxArena).  Every capture site of the package goes
    through here (graphs.GraphedModule / GraphedTrainStep, AT._GraphedSampleStep)."""

    def __init__(self, graph, eager_arena: bool = True, **kw):
        self.ctx = torch.cuda.graph(graph, **kw)
        self.eager_arena = eager_arena

    def __enter__(self):
        # the persistent LSTM launches keep their hand-off scratch per (device, stream), zeroed ONCE when it is allocated (its
        # sticky error words must survive replays): a buffer allocated inside the capture would come from the graph's pool and
        # be re-zeroed by nothing / by every replay -- so the capture stream's scratch is created here, eagerly, before the
        # capture begins (torch.cuda.graph synchronises the device on entry: the zero fill has landed by then)
        st = getattr(self.ctx, "capture_stream", None)
        if st is not None and LSTM_PERSIST:
            _persist_sync(torch.device("cuda", torch.cuda.current_device()), st.cuda_stream)
        self.ctx.__enter__()
        try:
            ABSMAX_ARENA.capture_begin(torch.device("cuda", torch.cuda.current_device()), eager=self.eager_arena)
        except BaseException:
            ABSMAX_ARENA.capture_end()
            self.ctx.__exit__(None, None, None)
            raise
        return self

    def __exit__(self, *exc):
        ABSMAX_ARENA.capture_end()
        return self.ctx.__exit__(*exc)


def _new_absmax(device) -> torch.Tensor:
    """A zero-filled abs-max buffer (32 slots, one per 128-byte line; see egz_common.h and _AbsmaxArena)."""
    return ABSMAX_ARENA.take(device)


def absmax_value(am: torch.Tensor) -> torch.Tensor:
    """max |x| held by an abs-max buffer, as a 1-element float32 tensor (the maximum over its slots)."""
    return am.view(32, -1)[:, 0].max().reshape(1).view(torch.float32)


def _want_absmax() -> bool:
    return PRECISION == "split" and GRAD_SPLIT == "f16"


# Forward activations get the same treatment as gradients: the pass that WRITES a post-ReLU activation (BN apply + ReLU
# [+ pool], the bias + ReLU epilogue of a decoder conv) also emits max |a|, and the consuming convolution (forward operand)
# and weight gradient (x operand) multiply by the matching power of two before the f16 split -- without it the pair's 22 bits
# only hold for max |a| in [2^-3, 65504] (lo goes subnormal below, hi saturates above).  (test_forward_scaling_is_needed flips the constant).
FWD_SCALE = True
ABSMAX_STATS = {"standalone": 0}         # standalone egz_absmax passes (an operand arrived without its producer's abs-max)


def _want_fwd_absmax() -> bool:
    return PRECISION == "split" and FWD_SCALE


def carry_absmax(dst: torch.Tensor, src) -> torch.Tensor:
    """Hand the abs-max scalar of ``src`` (a tensor or the scalar buffer itself) on to ``dst`` -- a view / re-layout of the same
    values (permutes and detaches create new tensor objects, which drop Python attributes)."""
    am = getattr(src, "_egz_absmax", None) if isinstance(src, torch.Tensor) and src.dtype != torch.int32 else src
    if am is not None:
        dst._egz_absmax = am
    if isinstance(src, torch.Tensor) and getattr(src, "_egz_presplit", False):
        dst._egz_presplit = True          # (the tensor holds f16 pairs, not fp32 values: see PRESPLIT)
    return dst


def absmax_of(x: torch.Tensor) -> torch.Tensor:
    """The abs-max scalar of a gradient tensor: the one its producer kernel attached (bn_relu_pool_bwd, relu_bwd_bias,
    pairmax_bwd compute it in their own pass), else a standalone reduction (egz_absmax)."""
    am = getattr(x, "_egz_absmax", None)
    if am is None:
        _req(x, "x")
        am = _new_absmax(x.device)
        ABSMAX_STATS["standalone"] += 1
        check(LIB.egz_absmax(x.data_ptr(), x.numel(), am.data_ptr(), _stream()), "egz_absmax")
        x._egz_absmax = am
    return am


# Pre-split activations (round 5).  Every split-half launch re-derives the f16 hi / lo pair of each staged
# activation from fp32 on the vector ALU -- ~3.5 instructions per staged float inside kernels that sit at the chip's power limit
# (timing-only builds without the split: conv fwd / dgrad -4 ... -5 %, weight gradient -7.5 % on the microbenchmark, more inside
# the step: profiles/r05_presplit_gonogo.txt).  Where the producer of an activation is a streaming pass that knows the
# tensor's EXACT abs-max before it writes the first element, it stores the pair instead of the fp32 value (same 4 bytes per
# element): the [BatchNorm -> ReLU (-> pool)] pass of an encoder block whose consumer is the next block's convolution.  The
# abs-max comes from the per-channel max / min of the conv output (atomic max in the conv's statistics epilogue) pushed through
# the BatchNorm's monotonic map by egz_bn_finalize_bound.  Consumers (the next conv's forward and weight gradient) then stage
# the pair without touching the vector ALU; the pair is bit-identical to the one they would have formed, so every result is.
# hipops.PRESPLIT = False keeps fp32 activations everywhere (test_presplit_activations_bit_identical flips the constant).
# Backward arithmetic.  Default (3): THREE MFMA products per MAC in the data and weight gradients too -- a_hi b_hi + a_hi b_lo +
# a_lo b_hi, both operands with 22 significant bits, fp32-class gradients (2e-7 per op): the arithmetic class of the reference's
# fp32 autograd (SP.py:132-138), and what every headline number is timed on.
# EGAZE_BWD_PRODUCTS=2 is an OPT-IN (like EGAZE_GRAD_SPLIT=bf16): the backward convolutions of the wide layers issue TWO products
# per MAC -- the operand that goes through LDS as a halo image (dy in a data gradient, x in a weight gradient) enters with its
# f16 hi half only, rounded to nearest (11 significant bits); the other keeps its 22 (csrc/egz_common.h, egz_f16p2).  Gradients
# move by ~2e-4 relative L2 per convolution, <= 1e-3 through the whole backward chain; the forward pass and with it every
# predicted map is untouched.  tests: test_backward_two_products, test_two_product_backward_whole_model, the `two_products`
# fixture; bench.py reports it as extra.bwd2, never as `value`.
BWD_PRODUCTS = int(_os.environ.get("EGAZE_BWD_PRODUCTS", "3"))
if BWD_PRODUCTS not in (2, 3):
    raise ValueError(f"EGAZE_BWD_PRODUCTS={BWD_PRODUCTS}: 3 (default, fp32-class) or 2 (opt-in)")
P2_DTYPE = 0x10          # egz_conv3x3_fwd_streamed: dtype | 0x10
P2_WGRAD = 0x20000       # egz_conv3x3_wgrad: flags | 0x20000


def _note_p2(dtype: int, flops: float) -> None:
    """bench.py's roofline leg: algorithmic FLOPs of the conv launches that ran on two products (to price the family's MFMA work)."""
    if _p2(dtype) != dtype:
        PROF.note_flops("two_product_conv", flops)


def _p2(dtype: int) -> int:
    """dtype of a data-gradient launch on the streamed kernel: the two-product bit where the knob and the type allow it."""
    return dtype | P2_DTYPE if (BWD_PRODUCTS == 2 and dtype == F16X3) else dtype


PRESPLIT = True          # (A/B decided in round 5: -0.24 ms, bit-identical; test_presplit_activations_bit_identical flips it)
PRESPLIT_STATS = {"produced": 0, "fwd": 0, "wgrad": 0, "grad_produced": 0, "dgrad": 0, "wgrad_dy": 0}


def presplit_ok(B: int, H: int, W: int, C: int, K_next: int) -> bool:
    """Can the (B, H, W, C) output of a train-mode [BN -> ReLU (-> pool)] block be stored pre-split for a consuming 3x3 conv
    with K_next filters?  Needs the f16 x3 policy with forward scaling, the streamed kernel's 64- / 128-column tiles without
    split-K for the consumer's forward and the split-half 9-tap kernel for its weight gradient."""
    if not (PRESPLIT and PRECISION == "split" and GRAD_SPLIT == "f16" and FWD_SCALE and STREAMED):
        return False
    if C % 64 != 0 or K_next % 64 != 0 or C > 512 or 4 * B * H * W * C >= _SPLIT_MAX_BYTES:
        return False
    if not LIB.egz_conv3x3_streamed_ok(B, H, W, C, K_next, 0):
        return False
    if SPLITK and LIB.egz_conv3x3_streamed_splits(B, H, W, C, K_next) > 1:
        return False
    return bool(LIB.egz_conv3x3_wgrad_presplit_ok(B, H, W, C, K_next))


# ... and the GRADIENT the BatchNorm backward of an encoder block writes (the dy operand of that block's data gradient and weight
# gradient).  Its maximum is not known before the apply pass runs, so the pairs are scaled by a BOUND instead: |dy| <= |sc| (max
# |dout| + |mean dz| + max |xhat| |mean dz xhat|) per channel, from the abs-max the data-gradient kernel above emits for dout, the
# per-channel max / min of y the forward conv left behind and the two backward sums (egz_bn_relu_pool_bwd_presplit).  A bound that
# is 2^b above the true maximum only moves the f16 window: elements more than 2^(16 - b) below the maximum lose low-order bits
# of their lo half (absolute error <= 2^-37 of the bound), nothing saturates.  Unlike the forward activations this is not
# bit-identical to the fp32-gradient path (another power-of-two scale rounds sub-window elements differently): the step's
# gradients move by ~1e-7 relative (test_presplit_gradients_match flips the constant).
PRESPLIT_GRAD = True     # (A/B decided in round 5: -0.19 ms; test_presplit_gradients_match flips it)


def presplit_grad_ok(B: int, H: int, W: int, C: int, K: int) -> bool:
    """Can the gradient w.r.t. the (B, H, W, K) output of a C -> K conv (written by that block's BatchNorm backward) be stored
    pre-split?  Its consumers: the conv's data gradient (reduction over K, C output columns, unsplit streamed launch on the
    64- / 128-column tiles) and its weight gradient (the split-half 9-tap kernel)."""
    if not (PRESPLIT_GRAD and PRESPLIT and PRECISION == "split" and GRAD_SPLIT == "f16" and FWD_SCALE and STREAMED):
        return False
    if C % 64 != 0 or K % 64 != 0 or K > 512 or 4 * B * H * W * K >= _SPLIT_MAX_BYTES:
        return False
    if not LIB.egz_conv3x3_streamed_ok(B, H, W, K, C, 0):
        return False
    if SPLITK and LIB.egz_conv3x3_streamed_splits(B, H, W, K, C) > 1:
        return False
    return bool(LIB.egz_conv3x3_wgrad_presplit_ok(B, H, W, C, K))


# FLOP accounting of a launch over a zero-PADDED operand (the 20-channel flow stack runs as 32 channels): the algorithmic count
# prices the real channels (the padded count over-stated the step's FLOPs by 0.3 %).  Set around the launch.
ALGO_CHANNELS = [None]


def conv3x3_fwd(x: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], K: int, ups=False,
                epi: int = EPI_BIAS, tile_flag: int = 0, dtype: int = 0, p2: bool = False, absmax: Optional[torch.Tensor] = None,
                streamed: bool = False, bn_in: Optional[torch.Tensor] = None, want_minmax: bool = False,
                pre_in: bool = False, want_bound: bool = False, want_amax: bool = False):
    """x: (B, Hin, Win, C) NHWC.  Returns (y (B,H,W,K), stat_partial or None); H,W = 2*Hin,2*Win if ups.
    ups: False | 'fold' (3x3 taps on the virtual upsampled image, wp = 'fwd' packing) | 'phase' or True (four 2x2
    phase convolutions on the low-res input, 4/9 of the MACs, wp = 'ups_fwd' packing).
    ``bn_in``: x is the PRE-BatchNorm output of the block below and bn_in its (4, C) coefficients -- the kernel normalises +
    ReLUs while staging x (deferred BatchNorm; narrow streamed geometry only, x must carry the abs-max of the normalised
    values).  ``want_minmax``: with EPI_BIAS_STATS on the narrow kernel, also return y's per-channel max / min rows as
    ``y._egz_minmax`` (what the deferred BatchNorm of THIS layer needs).
    ``pre_in``: x holds PRE-SPLIT activations (bn_relu_pool_fwd(presplit_am=...); x must carry the abs-max they were scaled
    with) -- plain streamed f16 x3 launch with the statistics epilogue only.  ``want_bound`` (EPI_BIAS_STATS on the streamed
    kernel's 64- / 128-column tiles, unsplit): also fold y's per-channel max / min into a 2 K-uint buffer, returned as
    ``y._egz_mm`` (input of bn_finalize(..., mm=...)); absent when the launch took another route."""
    _req(x, "x")
    if pre_in and not (dtype == F16X3 and streamed and not ups and epi in (EPI_BIAS_STATS, EPI_BIAS) and bn_in is None
                       and getattr(x, "_egz_absmax", None) is not None):
        raise RuntimeError("a pre-split activation reached a launch that cannot take it")
    if (bn_in is not None or want_minmax) and not (dtype and streamed and not ups):
        raise RuntimeError("deferred BatchNorm operands exist on the streamed narrow kernel only")
    B, Hin, Win, C = x.shape
    H, W = (2 * Hin, 2 * Win) if ups else (Hin, Win)
    y = torch.empty((B, H, W, K), dtype=torch.float32, device=x.device)
    stat = None
    if dtype == F16X3 and absmax is None and FWD_SCALE:
        absmax = absmax_of(x)               # forward operand: the producer's max |x| (or one standalone pass)
    amo = None                              # max |y| of a bias + ReLU launch, for the convolution that consumes y
    if dtype and epi == EPI_BIAS_RELU and _want_fwd_absmax() and K % 64 == 0:
        amo = _new_absmax(x.device)
        y._egz_absmax = amo
    # ... or of a data gradient (epi EPI_BIAS without a bias): max |dx| bounds the BatchNorm backward of the block below
    # (hipops.PRESPLIT_GRAD); only the unsplit streamed launch on the 64- / 128-column tiles emits it
    amax_dgrad = bool(want_amax and dtype and streamed and not ups and epi == EPI_BIAS and K % 64 == 0)
    uflag = 0 if not ups else (1 if ups == "fold" else 3)
    flags = uflag | (epi << 4) | (0x200 if dtype else tile_flag)      # split kernels: 128-row tiles
    if epi == EPI_BIAS_STATS and not (dtype and streamed):
        rows = LIB.egz_conv3x3_stat_rows(B, H, W, K, flags)
        stat = torch.empty((rows, 2, K), dtype=torch.float64, device=x.device)
    if dtype and streamed and ups:      # wp = the 'ups_fwd_frag' packing: four phase convolutions on the streamed-weight kernel
        if ups == "fold" or epi not in (EPI_BIAS, EPI_BIAS_RELU) or bn_in is not None:
            raise RuntimeError("the streamed upsample forward is the phase form with a bias / bias + ReLU epilogue")
        PROF.note_flops("egz_conv3x3_fwd_split", 2.0 * B * H * W * K * 9 * C)
        check(LIB.egz_conv3x3_fwd_streamed(x.data_ptr(), wp.data_ptr(), _p(bias), y.data_ptr(), None, B, H, W, C, K,
                                           epi, dtype, 2, _p(absmax), None, _p(amo), None, None, _stream()),
              "egz_conv3x3_fwd_split")
        return y, None
    if dtype and streamed:       # wp = fragment-ordered packing (conv_weight): weights L2 -> registers, halo through LDS
        PROF.note_flops("egz_conv3x3_fwd_split", 2.0 * B * H * W * K * 9 * (ALGO_CHANNELS[0] or C))
        ns = LIB.egz_conv3x3_streamed_splits(B, H, W, C, K) if (SPLITK and epi <= EPI_BIAS_STATS) else 1
        if ns > 1 and pre_in:
            raise RuntimeError("a pre-split activation reached a split-K launch (hipops.presplit_ok excludes them)")
        if ns > 1:      # few pixel tiles (batch-1 inference, 14 x 14 layers at small batches): split the channel blocks
            if epi == EPI_BIAS_STATS:   # the fix-up pass emits one partial row per 32 pixels
                stat = torch.empty((LIB.egz_conv3x3_fwd_streamed_splitk_stat_rows(B, H, W), 2, K), dtype=torch.float64,
                                   device=x.device)
            nb = LIB.egz_conv3x3_fwd_streamed_splitk_ws_bytes(B, H, W, K, ns)
            ws = workspace(nb, x.device)
            check(LIB.egz_conv3x3_fwd_streamed_splitk(x.data_ptr(), wp.data_ptr(), _p(bias), y.data_ptr(), _p(stat), B, H, W,
                                                      C, K, epi, dtype, _p(absmax), ws.data_ptr(), nb, ns, _p(amo), _stream()),
                  "egz_conv3x3_fwd_streamed_splitk")
            return y, stat
        mm = None
        if epi == EPI_BIAS_STATS:
            rows = LIB.egz_conv3x3_streamed_stat_rows(B, H, W, C, K)
            stat = torch.empty((rows, 2, K), dtype=torch.float64, device=x.device)
            if want_minmax:
                mm = torch.empty((rows, 2, K), dtype=torch.float32, device=x.device)
                y._egz_minmax = mm
            elif want_bound and K % 64 == 0 and K <= 512:
                mm = _new_absmax(x.device)              # 2 K <= 1024 zero-filled uints: per-channel max of y and of -y
                y._egz_mm = mm
        if pre_in:
            PRESPLIT_STATS["fwd" if epi == EPI_BIAS_STATS else "dgrad"] += 1
        if amax_dgrad:
            amo = _new_absmax(x.device)
            y._egz_absmax = amo
        if p2:
            _note_p2(dtype, 2.0 * B * H * W * K * 9 * C)
        check(LIB.egz_conv3x3_fwd_streamed(x.data_ptr(), wp.data_ptr(), _p(bias), y.data_ptr(), _p(stat), B, H, W, C, K,
                                           epi, _p2(dtype) if p2 else dtype, 0x100 if pre_in else 0, _p(absmax), None, _p(amo), _p(bn_in), _p(mm),
                                           _stream()), "egz_conv3x3_fwd_split")
        return y, stat
    if pre_in:
        raise RuntimeError("a pre-split activation reached a launch that is not on the streamed kernel")
    if dtype:
        PROF.note_flops("egz_conv3x3_fwd_split", 2.0 * B * H * W * K * 9 * C)
        sflags = (flags & 0x33) | (tile_flag & 0x6000)      # 0x4000: split-K tail schedule, 0x2000: no halo-tile kernel
        nb = LIB.egz_conv3x3_fwd_split_ws_bytes(B, H, W, C, K, sflags)
        ws = workspace(nb, x.device) if nb else None
        check(LIB.egz_conv3x3_fwd_split(x.data_ptr(), wp.data_ptr(), _p(bias), y.data_ptr(), _p(stat), B, H, W, C, K,
                                        sflags, dtype, _p(ws), nb, _p(absmax), _p(amo), _stream()), "egz_conv3x3_fwd_split")
        return y, stat
    PROF.note_flops("egz_conv3x3_fwd", 2.0 * B * H * W * K * 9 * C)
    check(LIB.egz_conv3x3_fwd(x.data_ptr(), wp.data_ptr(), _p(bias), y.data_ptr(), _p(stat), B, H, W, C, K, flags,
                              _stream()), "egz_conv3x3_fwd")
    return y, stat


def conv3x3_dgrad(dy: torch.Tensor, wp_dgrad: torch.Tensor, C: int, dtype: int = 0, streamed: bool = False,
                  pre_in: bool = False) -> torch.Tensor:
    """dy: (B,H,W,K) -> dx (B,H,W,C) (for an upsampled conv this is the gradient of the upsampled input).
    dtype F16X3: dy is scaled by a power of two derived from its abs-max so that the f16 halves carry it.
    ``pre_in``: dy holds pre-split pairs (bn_relu_pool_bwd(presplit=...)).  dx carries max |dx| (``_egz_absmax``) where the launch
    can emit it."""
    y, _ = conv3x3_fwd(dy, wp_dgrad, None, C, ups=False, epi=EPI_BIAS, dtype=dtype,
                       absmax=absmax_of(dy) if dtype == F16X3 else None, streamed=streamed, pre_in=pre_in,
                       want_amax=_want_absmax() and PRESPLIT_GRAD, p2=True)
    return y


def conv3x3_ups_dgrad(dy: torch.Tensor, wp_ups_dgrad: torch.Tensor, C: int, dtype: int = 0, streamed: bool = False) -> torch.Tensor:
    """Data gradient of [upsample x2 -> conv3x3] w.r.t. the LOW-res input: dy (B,H,W,K) -> dx (B,H/2,W/2,C).
    streamed: wp = the 'ups_dgrad_frag' packing (conv_weight(..., 'ups_dgrad', ...)), polyphase streamed-weight kernel."""
    _req(dy, "dy")
    B, H, W, K = dy.shape
    dx = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=dy.device)
    if dtype and streamed:
        PROF.note_flops("egz_conv3x3_fwd_split", 2.0 * B * H * W * K * 9 * C)
        am = absmax_of(dy) if dtype == F16X3 else None
        _note_p2(dtype, 2.0 * B * H * W * K * 9 * C)
        check(LIB.egz_conv3x3_fwd_streamed(dy.data_ptr(), wp_ups_dgrad.data_ptr(), None, dx.data_ptr(), None, B, H, W, K, C,
                                           0, _p2(dtype), 1, _p(am), None, None, None, None, _stream()),
              "egz_conv3x3_fwd_streamed(ups_dgrad)")
        return dx
    if dtype:      # GEMM roles: reduction over the conv's K, output channels = the conv's C
        PROF.note_flops("egz_conv3x3_fwd_split", 2.0 * B * H * W * K * 9 * C)
        nb = LIB.egz_conv3x3_fwd_split_ws_bytes(B, H, W, K, C, 4)
        ws = workspace(nb, dy.device) if nb else None
        am = absmax_of(dy) if dtype == F16X3 else None
        check(LIB.egz_conv3x3_fwd_split(dy.data_ptr(), wp_ups_dgrad.data_ptr(), None, dx.data_ptr(), None, B, H, W, K, C,
                                        4, dtype, _p(ws), nb, _p(am), None, _stream()), "egz_conv3x3_fwd_split(ups_dgrad)")
        return dx
    PROF.note_flops("egz_conv3x3_ups_dgrad", 2.0 * B * H * W * K * 9 * C)     # algorithmic (reference) FLOPs
    check(LIB.egz_conv3x3_ups_dgrad(dy.data_ptr(), wp_ups_dgrad.data_ptr(), dx.data_ptr(), B, H, W, C, K,
                                    0, _stream()), "egz_conv3x3_ups_dgrad")
    return dx


EPI_MASK_SUMS = 3
# ReLU backward of a decoder block folded into the epilogue of the dgrad kernel above it (functions.ConvReLU): it removes 11
# passes over the decoder gradients (relu_bwd_bias, 1.0 ms per step).  Round 2 measured it slower (35.56 vs 35.39 ms,
# profiles/r02_bench_ab_knobs.txt) -- with the mask load inside a per-lane branch every epilogue element waited for its own
# load.  With the branch-free epilogue (buffer loads / stores with out-of-range offsets for invalid rows, round 3) the step
# is 0.55 ms faster with it (32.6 vs 33.15 ms, profiles/r03_ab_notes.txt), so it is the default (off position: test_relu_backward_folded_into_dgrad_epilogue flips MASK_FUSE).
MASK_FUSE = True
HEAD_MASK_FUSE = True   # the same for the block under the 1x1 head (test_head_sigmoid_masked_backward covers the kernel)
MASK_FUSE_STATS = {"produced": 0, "consumed": 0}                 # how often the fused form ran / was picked up (tests)


def conv3x3_dgrad_masked(dy: torch.Tensor, wq: torch.Tensor, C: int, dtype: int, mask_src: torch.Tensor, ups: bool):
    """Data gradient of a plain (``ups`` False) or upsample-fused (True) 3x3 conv on the streamed-weight kernel, with the
    ReLU backward of the layer BELOW folded into the epilogue: ``mask_src`` (B, H', W', C) is that layer's post-ReLU output
    (= this conv's input).  Returns (dx masked, stat rows (rows, 2, C) fp64 whose plane 0 sums to the bias gradient of the
    layer below, abs-max buffer of dx)."""
    _req(dy, "dy"); _req(mask_src, "mask_src")
    B, H, W, K = dy.shape
    Ho, Wo = (H // 2, W // 2) if ups else (H, W)
    if tuple(mask_src.shape) != (B, Ho, Wo, C):
        raise RuntimeError(f"mask_src {tuple(mask_src.shape)} does not match the gradient {(B, Ho, Wo, C)}")
    dx = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=dy.device)
    rows = (B * Ho * Wo + 127) // 128
    stat = torch.empty((rows, 2, C), dtype=torch.float64, device=dy.device)
    MASK_FUSE_STATS["produced"] += 1
    amo = _new_absmax(dy.device)
    PROF.note_flops("egz_conv3x3_fwd_split", 2.0 * B * H * W * K * 9 * C)
    am = absmax_of(dy) if dtype == F16X3 else None
    _note_p2(dtype, 2.0 * B * H * W * K * 9 * C)
    check(LIB.egz_conv3x3_fwd_streamed(dy.data_ptr(), wq.data_ptr(), None, dx.data_ptr(), stat.data_ptr(), B, H, W, K, C,
                                       EPI_MASK_SUMS, _p2(dtype), 1 if ups else 0, _p(am), mask_src.data_ptr(), amo.data_ptr(),
                                       None, None, _stream()), "egz_conv3x3_fwd_streamed(masked dgrad)")
    return dx, stat, amo


EPI_BNSUMS = 5
# BatchNorm-backward sums of the layer below folded into the narrow data-gradient kernel (late_fusion.py:10-12 chain at
# 32 channels): removes that layer's reduce pass (two reads of its 205 MB tensors at B = 32, 224 x 224).  BNSUMS_FUSE = False
# keeps the separate pass (A/B runs, and the parity test compares the two).
BNSUMS_FUSE = True
BNSUMS_STATS = {"produced": 0, "consumed": 0}


# ... and for the wide layers (the VGG encoders: conv -> BN -> ReLU -> conv without a pool in between, utils.py:64-76): the same
# epilogue on the 128- / 64-column tiles of the streamed kernel.  BNSUMS_WIDE = False keeps the reduce pass there (test_bn_backward_sums_folded_into_encoder_dgrad compares the two).
BNSUMS_WIDE = True


def bnsums_ok(B: int, H: int, W: int, C: int, K: int, dtype: int) -> bool:
    """Can the data gradient of a K <- C channel conv (dy (B,H,W,K) -> dx (B,H,W,C)) also produce the BatchNorm-backward sums
    of the C-channel block below?  The persistent narrow kernel (C, K <= 32), or the streamed kernel's 64- / 128-column
    tiles (C % 64 == 0)."""
    if not (BNSUMS_FUSE and dtype and 4 * B * H * W * C < 2 ** 32 and LIB.egz_conv3x3_streamed_ok(B, H, W, K, C, 0)):
        return False
    if C <= 32 and K <= 32:
        return C % 4 == 0 and K % 4 == 0 and H % 16 == 0 and W % 16 == 0
    return bool(BNSUMS_WIDE and STREAMED and C % 64 == 0 and K % 32 == 0
                and (not SPLITK or LIB.egz_conv3x3_streamed_splits(B, H, W, K, C) <= 1))      # (few-tile launches stay split-K)


def conv3x3_dgrad_bnsums(dy: torch.Tensor, wq: torch.Tensor, C: int, dtype: int, bn_y: torch.Tensor, coef: torch.Tensor,
                         pre_in: bool = False):
    """Data gradient of a plain 3x3 conv (dy (B,H,W,K) -> dx (B,H,W,C)) whose input is the output of a train-mode
    [BatchNorm -> ReLU]: ``bn_y`` (B,H,W,C) is that layer's pre-BN conv output and ``coef`` its (4, C) batch coefficients
    (mean, 1/std, scale, shift).  Returns (dx, sums) with sums (rows, 2, C) fp64 = partial rows of (sum dz, sum dz * xhat),
    the ``sums`` argument of bn_relu_pool_bwd for that layer."""
    _req(dy, "dy"); _req(bn_y, "bn_y"); _req(coef, "coef")
    B, H, W, K = dy.shape
    if tuple(bn_y.shape) != (B, H, W, C) or tuple(coef.shape) != (4, C):
        raise RuntimeError(f"bn_y {tuple(bn_y.shape)} / coef {tuple(coef.shape)} do not match the gradient {(B, H, W, C)}")
    dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dy.device)
    stat = torch.empty((LIB.egz_conv3x3_streamed_stat_rows(B, H, W, K, C), 2, C), dtype=torch.float64, device=dy.device)
    BNSUMS_STATS["produced"] += 1
    PROF.note_flops("egz_conv3x3_fwd_split", 2.0 * B * H * W * K * 9 * C)
    am = absmax_of(dy) if dtype == F16X3 else None
    amo = None
    if _want_absmax() and PRESPLIT_GRAD and C % 64 == 0:       # max |dx|: bounds the BatchNorm backward of the block below
        amo = _new_absmax(dy.device)
        dx._egz_absmax = amo
    if pre_in:
        if not (dtype == F16X3 and C % 64 == 0 and getattr(dy, "_egz_absmax", None) is not None):
            raise RuntimeError("a pre-split gradient reached a data-gradient launch that cannot take it")
        PRESPLIT_STATS["dgrad"] += 1
    if C % 64 == 0:        # (the narrow late-fusion form of this launch stays three-product)
        _note_p2(dtype, 2.0 * B * H * W * K * 9 * C)
    check(LIB.egz_conv3x3_fwd_streamed(dy.data_ptr(), wq.data_ptr(), None, dx.data_ptr(), stat.data_ptr(), B, H, W, K, C,
                                       EPI_BNSUMS, _p2(dtype), 0x100 if pre_in else 0, _p(am), bn_y.data_ptr(), _p(amo), coef.data_ptr(),
                                       None, _stream()), "egz_conv3x3_fwd_streamed(dgrad + BN sums)")
    return dx, stat


def colsum_f64(stat: torch.Tensor, ncols_out: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """stat: (rows, cols...) fp64 partial rows -> fp32 sums of the first ``ncols_out`` columns (fixed order)."""
    rows = stat.shape[0]
    cols = stat.numel() // rows
    res = _out(out, (ncols_out,), stat.device)
    ws = workspace(64 * cols * 8, stat.device)
    check(LIB.egz_colsum_f64(stat.data_ptr(), rows, cols, ncols_out, res.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
          "egz_colsum_f64")
    return res


WGRAD_SPLIT = 0x2000       # egz_conv3x3_wgrad flag: split-half arithmetic (bf16 x3, or f16 x3 when dy_absmax is passed)


WGRAD_TAPPACK = True      # K <= 8 filters on the narrow kernel: (tap, k) pairs as GEMM columns (False: nine zero-padded 32-column tiles;
#                           test_conv3x3_wgrad_split flips it)


def conv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor, ups: bool = False, variant_flag: int = 0,
                  precision: Optional[str] = None, out: Optional[torch.Tensor] = None,
                  x_bn: Optional[torch.Tensor] = None, x_pre: bool = False, dy_pre: bool = False) -> torch.Tensor:
    """-> dw (K, C, 3, 3); ``out`` = a contiguous K*C*9 destination (a gradient sink) written instead of a fresh tensor.
    ``x_bn``: x is a pre-BatchNorm tensor, normalised + ReLU'd while it is staged (deferred BatchNorm, narrow kernel only)."""
    _req(x, "x"); _req(dy, "dy")
    B, H, W, K = dy.shape
    C = x.shape[3]
    dw = _out(out, (K, C, 3, 3), x.device)
    flags = (1 if ups else 0) | variant_flag | (0 if WGRAD_TAPPACK else 0x4000)
    am = xam = None
    prec = precision or PRECISION
    if prec in ("split", "split_bf16", "split_f16"):
        flags |= WGRAD_SPLIT
        if prec == "split_f16" or (prec == "split" and GRAD_SPLIT == "f16"):
            am = absmax_of(dy)             # f16 x3 with dy scaled by its abs-max; bf16 x3 otherwise
            if FWD_SCALE:
                xam = absmax_of(x)         # ... and the activation operand by its own (the forward pass left it on x)
    if x_pre:      # x holds pre-split activations (see PRESPLIT): the split-half 9-tap kernel copies the pairs into its LDS image
        if not (flags & WGRAD_SPLIT and am is not None and getattr(x, "_egz_absmax", None) is not None and not ups and x_bn is None
                and LIB.egz_conv3x3_wgrad_presplit_ok(B, H, W, C, K)):
            raise RuntimeError("a pre-split activation reached a weight-gradient launch that cannot take it")
        xam = x._egz_absmax
        flags |= 0x8000
        PRESPLIT_STATS["wgrad"] += 1
    if dy_pre:     # dy holds pre-split pairs scaled by the bound in dy._egz_absmax (bn_relu_pool_bwd(presplit=...))
        if not (flags & WGRAD_SPLIT and am is not None and xam is not None and not ups and x_bn is None
                and LIB.egz_conv3x3_wgrad_presplit_ok(B, H, W, C, K)):
            raise RuntimeError("a pre-split gradient reached a weight-gradient launch that cannot take it")
        flags |= 0x10000
        PRESPLIT_STATS["wgrad_dy"] += 1
    if BWD_PRODUCTS == 2 and am is not None:
        flags |= P2_WGRAD          # two products per MAC on the wide split-half kernels (ignored by the others)
    nb = LIB.egz_conv3x3_wgrad_ws_bytes(B, H, W, C, K, flags)
    ws = workspace(nb, x.device)
    PROF.note_flops("egz_conv3x3_wgrad", 2.0 * B * H * W * K * 9 * (ALGO_CHANNELS[0] or C))
    check(LIB.egz_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, C, K, flags, ws.data_ptr(),
                                ws.numel(), _p(am), _p(xam), _p(x_bn), _stream()), "egz_conv3x3_wgrad")
    return dw


def conv_first_fwd(x_nchw: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], stats: bool,
                   want_minmax: bool = False, want_bound: bool = False):
    """``want_minmax`` (direct kernel, C <= 3 -> 32 filters, with stats): y's per-channel max / min rows as y._egz_minmax.
    ``want_bound`` (direct kernel, with stats): the 2 K-uint atomic form as y._egz_mm (see conv3x3_fwd)."""
    _req(x_nchw, "x"); _req(w, "weight")
    B, C, H, W = x_nchw.shape
    K = w.shape[0]
    y = torch.empty((B, H, W, K), dtype=torch.float32, device=x_nchw.device)
    stat = mm = mmo = None
    if stats:
        rows = LIB.egz_conv_first_stat_rows_for(B, H, W, C, K)
        stat = torch.empty((rows, 2, K), dtype=torch.float64, device=y.device)
        if want_minmax:
            mm = torch.empty((rows, 2, K), dtype=torch.float32, device=y.device)
            y._egz_minmax = mm
        if want_bound and C <= 3 and K in (32, 64) and B * H * W * K < 2 ** 31:
            mmo = _new_absmax(y.device)
            y._egz_mm = mmo
    PROF.note_flops("egz_conv_first_fwd", 2.0 * B * H * W * K * 9 * C)
    check(LIB.egz_conv_first_fwd(x_nchw.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), _p(stat), B, H, W, C, K,
                                 _p(mm), _p(mmo), _stream()), "egz_conv_first_fwd")
    return y, stat


def conv_first_wgrad(x_nchw: torch.Tensor, dy: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(x_nchw, "x"); _req(dy, "dy")
    B, C, H, W = x_nchw.shape
    K = dy.shape[3]
    dw = _out(out, (K, C, 3, 3), dy.device)
    ws = workspace(LIB.egz_conv_first_wgrad_ws_bytes(B, H, W, C), dy.device)
    PROF.note_flops("egz_conv_first_wgrad", 2.0 * B * H * W * K * 9 * C)
    check(LIB.egz_conv_first_wgrad(x_nchw.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, C, K, ws.data_ptr(),
                                   ws.numel(), _stream()), "egz_conv_first_wgrad")
    return dw


# ----------------------------------------------------------------------------- BN / ReLU / pool
def bn_finalize(stat: torch.Tensor, count: float, gamma, beta, running_mean, running_var, momentum: float,
                eps: float, num_batches_tracked: Optional[torch.Tensor] = None, mm: Optional[torch.Tensor] = None):
    """``num_batches_tracked``: the BatchNorm's int64 counter on the device, incremented by the same launch.
    ``mm``: the per-channel max / min buffer of the conv output (conv3x3_fwd(want_bound=True)) -> returns (coef, abs-max buffer
    holding the EXACT max of relu(y * scale + shift)) instead of coef alone."""
    rows, _, K = stat.shape
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not num_batches_tracked.is_cuda):
        raise RuntimeError("num_batches_tracked: expected an int64 HIP tensor")
    dev = stat.device
    coef = torch.empty((4, K), dtype=torch.float32, device=dev)      # mean, invstd, scale, shift
    ws = workspace(LIB.egz_bn_ws_bytes(K), dev)
    if running_mean is not None:        # written behind torch's back: invalidate what is cached on it (bn_eval_coeffs)
        running_mean._egz_epoch = getattr(running_mean, "_egz_epoch", 0) + 1
    if mm is not None:
        am = _new_absmax(dev)
        check(LIB.egz_bn_finalize_bound(stat.data_ptr(), rows, K, float(count), _p(gamma), _p(beta), _p(running_mean),
                                        _p(running_var), momentum, eps, coef[0].data_ptr(), coef[1].data_ptr(),
                                        coef[2].data_ptr(), coef[3].data_ptr(), _p(num_batches_tracked), ws.data_ptr(),
                                        ws.numel(), mm.data_ptr(), am.data_ptr(), _stream()), "egz_bn_finalize_bound")
        return coef, am
    check(LIB.egz_bn_finalize(stat.data_ptr(), rows, K, float(count), _p(gamma), _p(beta), _p(running_mean),
                              _p(running_var), momentum, eps, coef[0].data_ptr(), coef[1].data_ptr(),
                              coef[2].data_ptr(), coef[3].data_ptr(), _p(num_batches_tracked), ws.data_ptr(), ws.numel(),
                              _stream()), "egz_bn_finalize")
    return coef


# Deferred BatchNorm (late-fusion stack, training): the [BN -> ReLU] of a narrow block is applied by the NEXT block's conv and
# weight-gradient kernels while they stage its pre-BN output, so the normalised tensor (205 MB at B = 32, 224 x 224 x 32) is
# neither written nor read.  BN_DEFER = False materialises it as before (test_deferred_batchnorm_matches_materialised compares the two).
BN_DEFER = True
BN_DEFER_STATS = {"deferred": 0}


def bn_defer_ok(B: int, H: int, W: int, K: int, K_next: int, first_direct: bool) -> bool:
    """Can the [BN -> ReLU] of a block with K output channels at (B, H, W) be left to a next 3x3 conv with K_next filters?
    Needs the split-half policy, the persistent narrow conv kernel and the narrow weight-gradient kernel for K -> K_next, and
    a producer that emits per-channel max / min rows (the narrow conv kernel itself, or the direct first-layer kernel)."""
    if not (BN_DEFER and BNSUMS_FUSE and PRECISION == "split" and GRAD_SPLIT == "f16" and FWD_SCALE and STREAMED):
        return False
    if not (K in (16, 32) and K_next <= 32 and K_next % 4 == 0 and H % 16 == 0 and W % 16 == 0):
        return False
    if 4 * B * H * W * max(K, K_next) >= 2 ** 32:
        return False
    return bool(LIB.egz_conv3x3_streamed_ok(B, H, W, K, K_next, 0) and LIB.egz_conv3x3_streamed_ok(B, H, W, K_next, K, 0)
                and LIB.egz_conv3x3_wgrad_narrow_ok(B, H, W, K, K_next))


def bn_finalize_deferred(stat: torch.Tensor, minmax: torch.Tensor, count: float, gamma, beta, running_mean, running_var,
                         momentum: float, eps: float, num_batches_tracked: Optional[torch.Tensor] = None):
    """bn_finalize for a BatchNorm whose output is never materialised -> (coef (4, K), abs-max buffer holding the exact max of
    relu(y * scale + shift), derived from y's per-channel max / min rows)."""
    rows, _, K = stat.shape
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not num_batches_tracked.is_cuda):
        raise RuntimeError("num_batches_tracked: expected an int64 HIP tensor")
    dev = stat.device
    coef = torch.empty((4, K), dtype=torch.float32, device=dev)
    am = _new_absmax(dev)
    if running_mean is not None:
        running_mean._egz_epoch = getattr(running_mean, "_egz_epoch", 0) + 1
    check(LIB.egz_bn_finalize_deferred(stat.data_ptr(), rows, K, float(count), _p(gamma), _p(beta), _p(running_mean),
                                       _p(running_var), momentum, eps, coef[0].data_ptr(), coef[1].data_ptr(),
                                       coef[2].data_ptr(), coef[3].data_ptr(), _p(num_batches_tracked), minmax.data_ptr(),
                                       minmax.shape[0], am.data_ptr(), _stream()), "egz_bn_finalize_deferred")
    BN_DEFER_STATS["deferred"] += 1
    return coef, am


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps: float):
    """Eval-mode (scale, shift) rows of a BatchNorm.  Pass the module's own parameter / buffer objects: the result is cached
    on ``running_mean`` and reused until any of the four tensors changed (torch version counters for in-place torch
    writes such as load_state_dict or a broadcast, the per-tensor epoch for writes behind torch's back: bn_finalize's
    running-statistics update, the fused optimizer's step) -- at batch 1 the two launches per layer were a quarter of the
    device time of a captured forward."""
    key = (_tag(running_mean), _tag(running_var), None if gamma is None else _tag(gamma), None if beta is None else _tag(beta),
           float(eps), _stream_device(running_mean))
    hit = getattr(running_mean, "_egz_evalcoef", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    K = running_mean.numel()
    coef = torch.zeros((4, K), dtype=torch.float32, device=running_mean.device)
    g = None if gamma is None else gamma.detach()
    b = None if beta is None else beta.detach()
    check(LIB.egz_bn_eval_coeffs(K, _p(g), _p(b), running_mean.data_ptr(), running_var.data_ptr(), eps,
                                 coef[2].data_ptr(), coef[3].data_ptr(), _stream()), "egz_bn_eval_coeffs")
    # the cached rows are read by later launches on whatever stream is current then: make them safe to read from any
    # stream by finishing the two small launches first (once per weight change, not per forward)
    if not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream().synchronize()
        running_mean._egz_evalcoef = (key, coef)
    return coef


def _stream_device(t):
    return t.device.index or 0


# Inference: a [Conv2d 3x3 -> BatchNorm2d (eval) -> ReLU] block without a pool runs as ONE launch -- the BatchNorm's eval-mode
# affine map is folded into the convolution (w' = w * scale[k], b' = b * scale[k] + shift[k], the bias + ReLU epilogue), so the
# normalise pass (one read + one write of the layer's output, 8 of the 13 blocks of a VGG16-BN encoder) disappears.  The folded
# tensors are cached on the weight and rebuilt when the weight, the bias or the BatchNorm's coefficients change.  Rounding: the
# product w * scale is rounded once per weight (relative 6e-8) -- eval outputs move by ~1e-7 relative.  (tests flip the constant).
EVAL_FOLD = True
EVAL_FOLD_STATS = {"folded": 0}
INFER_CALL = False      # set by utils.conv_bn_relu_pool right before ConvBNReLUPool.apply: the block runs under torch.no_grad()


def bn_fold_key(weight, bias, gamma, beta, running_mean, running_var, eps: float):
    return (_tag(weight), None if bias is None else _tag(bias), _tag(running_mean), _tag(running_var),
            None if gamma is None else _tag(gamma), None if beta is None else _tag(beta), float(eps))


def bn_fold_is_warm(weight, bias, gamma, beta, running_mean, running_var, eps: float) -> bool:
    """True when the folded tensors cached on ``weight`` match the current parameters / statistics (so that
    bn_folded_conv launches nothing): what a hipGraph capture needs -- a cold OR stale cache takes the unfolded path."""
    hit = getattr(weight, "_egz_fold", None)
    return hit is not None and hit[0] == bn_fold_key(weight, bias, gamma, beta, running_mean, running_var, eps)


def bn_folded_conv(weight, bias, gamma, beta, running_mean, running_var, eps: float):
    """-> (w', b') of the folded block, cached on ``weight`` (pass the module's own parameter / buffer objects).  Not to be
    called inside a hipGraph capture before the cache is warm (the fold itself is a few stock elementwise kernels, once per
    weight change)."""
    coef = bn_eval_coeffs(gamma, beta, running_mean, running_var, eps)
    key = bn_fold_key(weight, bias, gamma, beta, running_mean, running_var, eps)
    hit = getattr(weight, "_egz_fold", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("bn_folded_conv: cold or stale cache inside a hipGraph capture (callers check bn_fold_is_warm)")
    with torch.no_grad():
        K = weight.shape[0]
        wf = hit[1] if hit is not None else torch.empty_like(weight.detach())
        bf = hit[2] if hit is not None else torch.empty((K,), dtype=torch.float32, device=weight.device)
        torch.mul(weight.detach(), coef[2].view(K, 1, 1, 1), out=wf)
        if bias is not None:
            torch.addcmul(coef[3], bias.detach(), coef[2], out=bf)
        else:
            bf.copy_(coef[3])
    touch_params([wf])                       # its packings are stale
    torch.cuda.current_stream().synchronize()
    weight._egz_fold = (key, wf, bf)
    return wf, bf


def bn_relu_pool_fwd(y: torch.Tensor, coef: torch.Tensor, pool: bool, out: Optional[torch.Tensor] = None,
                     presplit_am: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``presplit_am``: the exact abs-max of the output (bn_finalize(..., mm=...)): the output is then stored PRE-SPLIT (f16
    hi / lo pairs in the fp32 footprint, see PRESPLIT) and tagged ``_egz_presplit``; only conv3x3_fwd(pre_in=True) and
    conv3x3_wgrad(x_pre=True) can read it."""
    _req(y, "y")
    B, H, W, K = y.shape
    out = _out(out, (B, H // 2, W // 2, K) if pool else (B, H, W, K), y.device)
    if presplit_am is not None:
        check(LIB.egz_bn_relu_pool_fwd_presplit(y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), out.data_ptr(), B, H, W, K,
                                                int(pool), presplit_am.data_ptr(), _stream()), "egz_bn_relu_pool_fwd_presplit")
        out._egz_absmax = presplit_am
        out._egz_presplit = True
        PRESPLIT_STATS["produced"] += 1
        return out
    am = _new_absmax(y.device) if _want_fwd_absmax() else None
    check(LIB.egz_bn_relu_pool_fwd(y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), out.data_ptr(), B, H, W, K,
                                   int(pool), _p(am), _stream()), "egz_bn_relu_pool_fwd")
    if am is not None:
        out._egz_absmax = am      # max |out|, folded into the same pass: scales the f16 split of the consuming convolution
    return out


def bn_relu_pool_bwd(y: torch.Tensor, dout: torch.Tensor, coef: torch.Tensor, pool: bool,
                     out_dgamma: Optional[torch.Tensor] = None, out_dbeta: Optional[torch.Tensor] = None,
                     sums: Optional[torch.Tensor] = None, presplit=None):
    """Returns (dy, dgamma, dbeta); ``out_dgamma`` / ``out_dbeta``: K-float destinations (gradient sinks).
    ``sums``: (rows, 2, K) fp64 partial rows from conv3x3_dgrad_bnsums (the producer of ``dout``): no reduce pass.
    ``presplit`` = (abs-max buffer of dout, max / min buffer of y): dy is stored as pre-split pairs scaled by a bound of its
    maximum (see PRESPLIT_GRAD), tagged ``_egz_presplit`` and carrying that bound as ``_egz_absmax``."""
    _req(y, "y"); _req(dout, "dout")
    B, H, W, K = y.shape
    dy = torch.empty_like(y)
    dg, db = _out(out_dgamma, (K,), y.device), _out(out_dbeta, (K,), y.device)
    ws = workspace(LIB.egz_bn_relu_pool_bwd_ws_bytes(K), y.device)
    if presplit is not None:
        dout_am, y_mm = presplit
        am = _new_absmax(y.device)
        check(LIB.egz_bn_relu_pool_bwd_presplit(y.data_ptr(), dout.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(),
                                                coef[0].data_ptr(), coef[1].data_ptr(), dy.data_ptr(), dg.data_ptr(),
                                                db.data_ptr(), B, H, W, K, int(pool), ws.data_ptr(), ws.numel(), am.data_ptr(),
                                                _p(sums), 0 if sums is None else sums.shape[0], y_mm.data_ptr(),
                                                dout_am.data_ptr(), _stream()), "egz_bn_relu_pool_bwd_presplit")
        if sums is not None:
            BNSUMS_STATS["consumed"] += 1
        dy._egz_absmax = am
        dy._egz_presplit = True
        PRESPLIT_STATS["grad_produced"] += 1
        return dy, dg, db
    am = _new_absmax(y.device) if _want_absmax() else None
    check(LIB.egz_bn_relu_pool_bwd(y.data_ptr(), dout.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(),
                                   coef[0].data_ptr(), coef[1].data_ptr(), dy.data_ptr(), dg.data_ptr(),
                                   db.data_ptr(), B, H, W, K, int(pool), ws.data_ptr(), ws.numel(), _p(am),
                                   _p(sums), 0 if sums is None else sums.shape[0], _stream()), "egz_bn_relu_pool_bwd")
    if sums is not None:
        BNSUMS_STATS["consumed"] += 1
    if am is not None:
        dy._egz_absmax = am      # max |dy|, folded into the same pass: scales the f16 split of the conv backward
    return dy, dg, db


FIRST_FUSE = True        # first block of a narrow stack: BatchNorm backward + weight gradient in one pass


def bn_bwd_first_wgrad_ok(C: int, K: int, pool: bool) -> bool:
    return bool(FIRST_FUSE and K in (32, 64) and 1 <= C <= 3 and not pool)


def bn_bwd_first_wgrad(y: torch.Tensor, dout: torch.Tensor, coef: torch.Tensor, x_nchw: torch.Tensor,
                       out_dgamma: Optional[torch.Tensor] = None, out_dbeta: Optional[torch.Tensor] = None,
                       out_dw: Optional[torch.Tensor] = None, sums: Optional[torch.Tensor] = None):
    """Backward of a first [Conv2d(C <= 3 -> 32 | 64) -> BN(train) -> ReLU] block (the late-fusion stack; the RGB encoder) in one
    pass over (y, dout): returns (dw (K, C, 3, 3), dgamma, dbeta); the gradient w.r.t. the conv output is never materialised."""
    _req(y, "y"); _req(dout, "dout"); _req(x_nchw, "x")
    B, H, W, K = y.shape
    C = x_nchw.shape[1]
    if tuple(x_nchw.shape) != (B, C, H, W) or tuple(dout.shape) != tuple(y.shape):
        raise RuntimeError(f"x {tuple(x_nchw.shape)} / dout {tuple(dout.shape)} do not match y {tuple(y.shape)}")
    dg, db = _out(out_dgamma, (K,), y.device), _out(out_dbeta, (K,), y.device)
    dw = _out(out_dw, (K, C, 3, 3), y.device)
    nb = LIB.egz_bn_bwd_first_wgrad_ws_bytes(C, K)
    ws = workspace(nb, y.device)
    check(LIB.egz_bn_bwd_first_wgrad(y.data_ptr(), dout.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), coef[0].data_ptr(),
                                     coef[1].data_ptr(), x_nchw.data_ptr(), dw.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                     B, H, W, C, K, ws.data_ptr(), ws.numel(), _p(sums), 0 if sums is None else sums.shape[0],
                                     _stream()), "egz_bn_bwd_first_wgrad")
    if sums is not None:
        BNSUMS_STATS["consumed"] += 1
    return dw, dg, db


def pairmax_fwd(y2: torch.Tensor) -> torch.Tensor:
    """y2: (2B, H, W, K) = stream s rows then stream t rows -> (B, H, W, K) element-wise max (s wins ties)."""
    _req(y2, "y2")
    B2 = y2.shape[0]
    z = torch.empty((B2 // 2,) + tuple(y2.shape[1:]), dtype=torch.float32, device=y2.device)
    check(LIB.egz_pairmax_fwd(y2.data_ptr(), z.data_ptr(), z.numel(), _stream()), "egz_pairmax_fwd")
    return z


def pairmax_bwd(y2: torch.Tensor, dz: torch.Tensor) -> torch.Tensor:
    _req(y2, "y2"); _req(dz, "dz")
    dy2 = torch.empty_like(y2)
    am = _new_absmax(y2.device) if _want_absmax() else None
    check(LIB.egz_pairmax_bwd(y2.data_ptr(), dz.data_ptr(), dy2.data_ptr(), dz.numel(), _p(am), _stream()),
          "egz_pairmax_bwd")
    if am is not None:
        dy2._egz_absmax = am
    return dy2


def stack2(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """(B, ...) + (B, ...) -> (2B, ...): the depth-2 stack of the fusion block folded into the batch dim, for callers
    whose two maps do not already share one buffer (two stream-ordered device copies, no torch.cat kernel)."""
    _req(a, "a"); _req(b, "b")
    out = torch.empty((2 * a.shape[0],) + tuple(a.shape[1:]), dtype=torch.float32, device=a.device)
    n = a.numel()
    check(LIB.egz_copy(a.data_ptr(), out.data_ptr(), n, _stream()), "egz_copy")
    check(LIB.egz_copy(b.data_ptr(), out.data_ptr() + 4 * n, n, _stream()), "egz_copy")
    return out


def cat2_planes(f: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """torch.cat((f, g), dim=1) of two (B, 1, H, W) maps in one kernel (models/late_fusion.py:19)."""
    _req(f, "f"); _req(g, "g")
    B, _, Hh, Ww = f.shape
    out = torch.empty((B, 2, Hh, Ww), dtype=torch.float32, device=f.device)
    check(LIB.egz_cat2_planes(f.data_ptr(), g.data_ptr(), out.data_ptr(), B, Hh * Ww, _stream()), "egz_cat2_planes")
    return out


def fill_zero(t: torch.Tensor):
    """hipMemsetAsync on the current stream (zero_grad of the flat gradient buffer)."""
    check(LIB.egz_fill_zero(t.data_ptr(), t.numel() * t.element_size(), _stream()), "egz_fill_zero")


def channel_stats(x: torch.Tensor) -> torch.Tensor:
    _req(x, "x")
    K = x.shape[-1]
    rows = x.numel() // K
    stat = torch.empty((LIB.egz_channel_stats_rows(), 2, K), dtype=torch.float64, device=x.device)
    check(LIB.egz_channel_stats(x.data_ptr(), rows, K, stat.data_ptr(), _stream()), "egz_channel_stats")
    return stat


def relu_bwd(out: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
    _req(out, "out"); _req(dout, "dout")
    dy = torch.empty_like(dout)
    check(LIB.egz_relu_bwd(out.data_ptr(), dout.data_ptr(), dy.data_ptr(), out.numel(), _stream()), "egz_relu_bwd")
    return dy


def relu_bwd_bias(out: torch.Tensor, dout: torch.Tensor, out_db: Optional[torch.Tensor] = None):
    """ReLU backward fused with the producing conv's bias gradient: (dy, db)."""
    _req(out, "out"); _req(dout, "dout")
    K = out.shape[-1]
    rows = out.numel() // K
    dy = torch.empty_like(dout)
    db = _out(out_db, (K,), out.device)
    ws = workspace(LIB.egz_relu_bwd_bias_ws_bytes(K), out.device)
    am = _new_absmax(out.device) if _want_absmax() else None
    check(LIB.egz_relu_bwd_bias(out.data_ptr(), dout.data_ptr(), dy.data_ptr(), db.data_ptr(), rows, K,
                                ws.data_ptr(), ws.numel(), _p(am), _stream()), "egz_relu_bwd_bias")
    if am is not None:
        dy._egz_absmax = am
    return dy, db


def upsample2x_bwd(dxu: torch.Tensor) -> torch.Tensor:
    _req(dxu, "dxu")
    B, H2, W2, C = dxu.shape
    dx = torch.empty((B, H2 // 2, W2 // 2, C), dtype=torch.float32, device=dxu.device)
    check(LIB.egz_upsample2x_bwd(dxu.data_ptr(), dx.data_ptr(), B, H2 // 2, W2 // 2, C, _stream()),
          "egz_upsample2x_bwd")
    return dx


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Sum over all leading dims of an NHWC tensor -> (K,) (conv bias gradient)."""
    _req(x, "x")
    K = x.shape[-1]
    rows = x.numel() // K
    out = _out(out, (K,), x.device)
    ws = workspace(64 * K * 8, x.device)
    check(LIB.egz_colsum(x.data_ptr(), rows, K, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "egz_colsum")
    return out


def nchw_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    _req(x, "x")
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, C), dtype=torch.float32, device=x.device)
    check(LIB.egz_nchw_to_nhwc(x.data_ptr(), out.data_ptr(), B, C, H, W, _stream()), "egz_nchw_to_nhwc")
    return out


def nchw_to_nhwc_pad(x: torch.Tensor, Cp: int) -> torch.Tensor:
    """(B, C, H, W) -> (B, H, W, Cp) with zero channels C .. Cp-1."""
    _req(x, "x")
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, Cp), dtype=torch.float32, device=x.device)
    check(LIB.egz_nchw_to_nhwc_pad(x.data_ptr(), out.data_ptr(), B, C, H, W, Cp, _stream()), "egz_nchw_to_nhwc_pad")
    return out


def prepare_network_input(x: torch.Tensor, Cp: int = 32):
    """Input-side work of the NEXT forward pass, issued NOW on the current stream (a copy / helper stream): the re-layout of a
    16 ... 32-channel NCHW network input (the 20-channel flow stack, SP.py:128) into the zero-padded NHWC-32 form its first
    convolution reads, and the abs-max of the result (the scale of its f16 split).  Neither depends on the weights, so a driver
    that has batch k + 1 on the device while step k computes (data.STdatas.staged_batches, bench.py) moves these two HBM passes
    (~0.14 ms at batch 32) out of the serial head of the forward pass and under the MFMA-bound kernels of the step before.  The
    prepared tensor rides on ``x`` and is consumed ONCE by functions.ConvBNReLUPool (the tensor's version and shape are
    checked; anything else falls back to the in-step conversion)."""
    if not (x.is_cuda and x.dim() == 4 and 16 <= x.shape[1] <= Cp and PRECISION == "split" and x.is_contiguous()):
        return None
    xin = nchw_to_nhwc_pad(x.detach(), Cp)
    if _want_fwd_absmax():
        absmax_of(xin)
    ev = torch.cuda.Event()
    ev.record()
    x._egz_prepared = (xin, x._version, ev)
    return xin


def take_prepared_input(x: torch.Tensor, Cp: int = 32):
    """The tensor prepare_network_input left on ``x`` (or None), handed over to the current stream; one-shot."""
    prep = getattr(x, "_egz_prepared", None)
    if prep is None:
        return None
    del x._egz_prepared
    xin, version, ev = prep
    if version != x._version or xin.shape[0] != x.shape[0] or xin.shape[3] != Cp or tuple(xin.shape[1:3]) != tuple(x.shape[2:]):
        return None
    cur = torch.cuda.current_stream()
    cur.wait_event(ev)
    xin.record_stream(cur)
    return xin


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    _req(x, "x")
    B, H, W, C = x.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    check(LIB.egz_nhwc_to_nchw(x.data_ptr(), out.data_ptr(), B, C, H, W, _stream()), "egz_nhwc_to_nchw")
    return out


# ----------------------------------------------------------------------------- head + losses
def conv1x1_sigmoid_fwd(x: torch.Tensor, w: torch.Tensor, bias, want_logits: bool = False):
    _req(x, "x"); _req(w, "weight")
    C = x.shape[-1]
    M = x.numel() // C
    out = torch.empty(tuple(x.shape[:-1]), dtype=torch.float32, device=x.device)
    logits = torch.empty_like(out) if want_logits else None
    check(LIB.egz_conv1x1_sigmoid_fwd(x.data_ptr(), w.data_ptr(), _p(bias), out.data_ptr(), _p(logits), M, C,
                                      _stream()), "egz_conv1x1_sigmoid_fwd")
    return out, logits


def conv1x1_sigmoid_bwd(x, w, out, dout, need_dx: bool = True, out_dw=None, out_db=None):
    _req(x, "x"); _req(out, "out"); _req(dout, "dout")
    C = x.shape[-1]
    M = x.numel() // C
    dx = torch.empty_like(x) if need_dx else None
    dw = _out(out_dw, (1, C, 1, 1), x.device)
    db = _out(out_db, (1,), x.device)
    ws = workspace(LIB.egz_conv1x1_sigmoid_bwd_ws_bytes(C), x.device)
    check(LIB.egz_conv1x1_sigmoid_bwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), dout.data_ptr(), _p(dx),
                                      dw.data_ptr(), db.data_ptr(), M, C, ws.data_ptr(), ws.numel(), _stream()),
          "egz_conv1x1_sigmoid_bwd")
    return dx, dw, db


def conv1x1_sigmoid_bwd_masked(x, w, out, dout, out_dw=None, out_db=None):
    """Head backward with the ReLU backward of the conv block below folded in (x = that block's post-ReLU output):
    -> (dx masked, dw, db, stat rows (rows, C) fp64 whose column sums are the block's bias gradient, abs-max buffer of dx) --
    the ``_egz_premasked`` hand-over of conv3x3_dgrad_masked (functions.ConvReLU.backward)."""
    _req(x, "x"); _req(out, "out"); _req(dout, "dout")
    C = x.shape[-1]
    M = x.numel() // C
    dx = torch.empty_like(x)
    dw = _out(out_dw, (1, C, 1, 1), x.device)
    db = _out(out_db, (1,), x.device)
    ws = workspace(LIB.egz_conv1x1_sigmoid_bwd_ws_bytes(C), x.device)
    stat = torch.empty((int(LIB.egz_conv1x1_sigmoid_bwd_rows(M, C)), C), dtype=torch.float64, device=x.device)
    amo = _new_absmax(x.device)
    check(LIB.egz_conv1x1_sigmoid_bwd_masked(x.data_ptr(), w.data_ptr(), out.data_ptr(), dout.data_ptr(), dx.data_ptr(),
                                             dw.data_ptr(), db.data_ptr(), stat.data_ptr(), amo.data_ptr(), M, C,
                                             ws.data_ptr(), ws.numel(), _stream()), "egz_conv1x1_sigmoid_bwd_masked")
    MASK_FUSE_STATS["produced"] += 1
    return dx, dw, db, stat, amo


def floss_fwd(inp: torch.Tensor, target: torch.Tensor, weighted: bool = True):
    """Returns (loss 0-dim tensor, weights or None).  inp/target: (B,1,H,W) or (B,H,W)."""
    _req(inp, "input"); _req(target, "target")
    B, H, W = inp.shape[0], inp.shape[-2], inp.shape[-1]
    loss = torch.empty((), dtype=torch.float32, device=inp.device)
    weights = torch.empty(inp.numel(), dtype=torch.float32, device=inp.device) if weighted else None
    ws = workspace(LIB.egz_loss_ws_bytes(B), inp.device)
    check(LIB.egz_floss_fwd(inp.data_ptr(), target.data_ptr(), _p(weights), loss.data_ptr(), B, H, W, int(weighted),
                            ws.data_ptr(), ws.numel(), _stream()), "egz_floss_fwd")
    return loss, weights


def floss_bwd(inp, target, weights, grad_out: Optional[torch.Tensor]) -> torch.Tensor:
    dinp = torch.empty_like(inp)
    check(LIB.egz_floss_bwd(inp.data_ptr(), target.data_ptr(), _p(weights), _p(grad_out), dinp.data_ptr(),
                            inp.numel(), _stream()), "egz_floss_bwd")
    return dinp


def mse_fwd(a: torch.Tensor, b: torch.Tensor, tanh_target: bool = False) -> torch.Tensor:
    """mean((a - b)^2), or mean((a - tanh(b))^2) with ``tanh_target`` (AT.py:138 in one pass)."""
    _req(a, "a"); _req(b, "b")
    loss = torch.empty((), dtype=torch.float32, device=a.device)
    ws = workspace(1024 * 8, a.device)
    check(LIB.egz_mse_fwd(a.data_ptr(), b.data_ptr(), loss.data_ptr(), a.numel(), ws.data_ptr(), ws.numel(),
                          int(tanh_target), _stream()), "egz_mse_fwd")
    return loss


def mse_fwd_grad(a: torch.Tensor, b: torch.Tensor, tanh_target: bool = False, ring: Optional[torch.Tensor] = None,
                 counter: Optional[torch.Tensor] = None):
    """(loss, d loss / d a) of nn.MSELoss in one launch (n <= 4096; the AT per-sample step).  ``ring`` / ``counter``: the loss is
    also parked in ring[counter[0] % len(ring)] (counter: a device int32 tensor, e.g. FusedAdam.step_dev)."""
    _req(a, "a"); _req(b, "b")
    loss = torch.empty((), dtype=torch.float32, device=a.device)
    da = torch.empty_like(a)
    check(LIB.egz_mse_fwd_grad(a.data_ptr(), b.data_ptr(), loss.data_ptr(), da.data_ptr(), a.numel(), int(tanh_target),
                               _p(ring), 0 if ring is None else ring.numel(), _p(counter), _stream()), "egz_mse_fwd_grad")
    return loss, da


def mse_bwd(a, b, grad_out, tanh_target: bool = False) -> torch.Tensor:
    da = torch.empty_like(a)
    check(LIB.egz_mse_bwd(a.data_ptr(), b.data_ptr(), _p(grad_out), da.data_ptr(), a.numel(), int(tanh_target), _stream()),
          "egz_mse_bwd")
    return da


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, lo: int = 0, hi: Optional[int] = None, nonfinite=None):
    """One Adam step over the flat buffers, or over their slice [lo, hi) (element offsets, multiples of 4).  ``nonfinite``: int32
    device word that gets bit 0 set when the step skipped an element because its gradient was NaN / inf (csrc/adam.hip)."""
    hi = p.numel() if hi is None else hi
    if hi <= lo:
        return
    o = 4 * lo
    check(LIB.egz_adam_step(p.data_ptr() + o, g.data_ptr() + o, m.data_ptr() + o, v.data_ptr() + o, hi - lo, lr, beta1, beta2,
                            eps, int(step), grad_scale, _p(nonfinite), _stream()), "egz_adam_step")


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, step_dev, grad_scale=1.0, nonfinite=None):
    """Adam step whose counter of completed steps lives on the device (int32 tensor): usable inside a captured hipGraph."""
    check(LIB.egz_adam_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
                                step_dev.data_ptr(), grad_scale, _p(nonfinite), _stream()), "egz_adam_step_dev")


# ----------------------------------------------------------------------------- AT: GEMM + LSTM cell
def gemm(a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, a_strides, b_strides, out: torch.Tensor = None,
         bias: torch.Tensor = None, accumulate: bool = False, relu: bool = False) -> torch.Tensor:
    """out[M][N] = op(A)[M][K] @ op(B)[K][N] (+out) (+bias) (relu).  a_strides = (stride_m, stride_k) of op(A) in
    elements, b_strides = (stride_k, stride_n) of op(B) -- transposes are expressed through strides."""
    _req(a, "A"); _req(b, "B")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    flags = (1 if accumulate else 0) | (2 if relu else 0)
    PROF.note_flops("egz_gemm", 2.0 * M * N * K)
    check(LIB.egz_gemm(a.data_ptr(), b.data_ptr(), out.data_ptr(), _p(bias), M, N, K, a_strides[0], a_strides[1],
                       b_strides[0], b_strides[1], N if out.dim() < 2 else out.stride(-2), flags, _stream()), "egz_gemm")
    return out


def linear_fwd(x2d: torch.Tensor, w: torch.Tensor, bias=None, relu=False, out=None, accumulate=False):
    """x2d [M][K] @ w[N][K]^T (+bias): the nn.Linear / LSTM gate projection."""
    M, K = x2d.shape
    N = w.shape[0]
    return gemm(x2d, w, M, N, K, (K, 1), (1, K), out=out, bias=bias, accumulate=accumulate, relu=relu)


def matmul_nn(a2d: torch.Tensor, w: torch.Tensor):
    """a2d [M][N] @ w[N][K] -> [M][K]  (data gradient of a Linear)."""
    M, N = a2d.shape
    K = w.shape[1]
    return gemm(a2d, w, M, K, N, (N, 1), (K, 1))


def matmul_tn(a2d: torch.Tensor, b2d: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False):
    """a2d[R][M]^T @ b2d[R][N] -> [M][N]  (weight gradient: dY^T X); ``out`` = an M*N destination (gradient sink)."""
    R, M = a2d.shape
    N = b2d.shape[1]
    return gemm(a2d, b2d, M, N, R, (1, M), (N, 1), out=None if out is None else _out(out, (M, N), a2d.device),
                accumulate=accumulate)


def matmul_tn_batched(a_list, b_list, outs=None):
    """[a^T @ b for a, b in zip(a_list, b_list)] in ONE launch (egz_gemm_batched): every a (R, M), every b (R, N) of one shape.
    ``outs``: per product an M*N destination (a gradient sink) or None (a fresh tensor)."""
    n = len(a_list)
    R, M = a_list[0].shape
    N = b_list[0].shape[1]
    res = []
    for i in range(n):
        _req(a_list[i], "A"); _req(b_list[i], "B")
        if tuple(a_list[i].shape) != (R, M) or tuple(b_list[i].shape) != (R, N):
            raise RuntimeError("matmul_tn_batched: the products must share one shape")
        o = None if outs is None else outs[i]
        res.append(torch.empty((M, N), dtype=torch.float32, device=a_list[i].device) if o is None else _out(o, (M, N), a_list[i].device))
    PROF.note_flops("egz_gemm", 2.0 * M * N * R * n)
    check(LIB.egz_gemm_batched(_ptr_table(a_list), _ptr_table(b_list), _ptr_table(res), n, M, N, R, 1, M, N, 1, N, 0, _stream()),
          "egz_gemm_batched")
    return res


def transpose2d(x: torch.Tensor) -> torch.Tensor:
    """[R][C] -> [C][R] (LDS-tiled; W_hh -> W_hh^T for the LSTM backward)."""
    _req(x, "x")
    R, C = x.shape
    out = torch.empty((C, R), dtype=torch.float32, device=x.device)
    check(LIB.egz_nhwc_to_nchw(x.data_ptr(), out.data_ptr(), 1, C, R, 1, _stream()), "egz_nhwc_to_nchw")
    return out


def copy_into(dst: torch.Tensor, src: torch.Tensor):
    """Stream-ordered device copy between contiguous fp32 buffers of the same size."""
    _req(dst, "dst"); _req(src, "src")
    if dst.numel() != src.numel():
        raise RuntimeError("copy_into: size mismatch")
    check(LIB.egz_copy(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), "egz_copy")


# The recurrence of the AT network as ONE persistent, weight-stationary launch per direction (egz_lstm_persist_fwd / _bwd) where its
# geometry allows (L = 2, H = 512, B <= 32): T=16 / B=32 forward 167 -> 90 us, backward 190 -> 119 us (profiles/r05_ab_notes.txt).
# EGAZE_LSTM_PERSIST=0: always the wavefront launches (egz_lstm_wave_fwd / _bwd: T + 1 / T + 3 launches).
LSTM_PERSIST = _os.environ.get("EGAZE_LSTM_PERSIST", "1") != "0"
# Hand-off scratch of the persistent launches: ONE buffer per (device, stream), zeroed when it is allocated and kept -- launches
# on a stream are ordered, so they may share their counters (every call zeroes those itself); its last line holds the two STICKY
# error words (forward, backward) that a timed-out hand-off raises and that no launch clears, so a failure survives the launches
# and graph replays that follow it until lstm_persist_check() looks.
_PERSIST_SYNC: dict = {}


def _persist_sync(dev: torch.device, stream_handle=None) -> torch.Tensor:
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream() if stream_handle is None else stream_handle)
    buf = _PERSIST_SYNC.get(key)
    if buf is None:
        if torch.cuda.is_current_stream_capturing():
            # (an allocation inside a capture would come from the graph's private pool and be zeroed by nothing)
            raise RuntimeError("persistent LSTM launch inside a hipGraph capture that did not go through hipops.capture() "
                               "(which creates the capture stream's hand-off scratch before the capture begins)")
        buf = torch.zeros((LIB.egz_lstm_persist_sync_words(),), dtype=torch.int32, device=dev)
        _PERSIST_SYNC[key] = buf
    return buf


def lstm_persist_errors() -> dict:
    """The sticky error words of every persistent-LSTM scratch (synchronises): {'fwd': e, 'bwd': e}, 0 = every in-launch hand-off
    arrived since the last reset, 1 + s = some launch had a block give up waiting in global step s (that launch's outputs -- and
    every gradient derived from them -- are NaN from that step on)."""
    out = {"fwd": 0, "bwd": 0}
    for buf in _PERSIST_SYNC.values():
        w = buf[-32:-30].tolist()
        out["fwd"] = max(out["fwd"], int(w[0]))
        out["bwd"] = max(out["bwd"], int(w[1]))
    return out


def lstm_persist_status() -> int:
    """0 when no persistent LSTM launch since the last reset lost a hand-off, else the larger of the two sticky words."""
    e = lstm_persist_errors()
    return max(e["fwd"], e["bwd"])


def lstm_persist_check(reset: bool = True) -> None:
    """Raise if a persistent LSTM launch timed out since the last check.  Called where the training loops synchronise anyway (the
    AT loss-ring drain, AT.trainLSTM's per-epoch read-back).  The poisoned step did not reach the weights (FusedAdam skips
    non-finite gradient elements), so the caller can continue on the wavefront launches: the check switches the persistent form
    off for the rest of the process before it raises."""
    e = lstm_persist_errors()
    if e["fwd"] or e["bwd"]:
        global LSTM_PERSIST
        LSTM_PERSIST = False
        if reset:
            for buf in _PERSIST_SYNC.values():
                buf[-32:].zero_()
        raise RuntimeError(f"persistent LSTM launch lost an in-launch hand-off (forward: {e['fwd']}, backward: {e['bwd']}; 1 + the "
                           f"global step a block gave up in): not every block got a compute unit -- another kernel was holding CUs "
                           f"for longer than 0.5 s.  The outputs of that step were NaN and FusedAdam skipped them; the "
                           f"persistent form is now off (wavefront launches from here on; EGAZE_LSTM_PERSIST=0 selects them "
                           f"from the start)")


class lstm_persistent:
    """``with hipops.lstm_persistent(False): ...`` -- the wavefront launches inside the block.  The persistent kernels hold one block
    per CU until the sequence is over, so they are the wrong form for an AT step that runs in the shadow of other work on the same
    device (bench.py's combined step: the SP step's MFMA kernels occupy every CU; measured +0.1 ... +0.2 ms on the SP step)."""

    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        global LSTM_PERSIST
        self.prev, LSTM_PERSIST = LSTM_PERSIST, self.on
        return self

    def __exit__(self, *exc):
        global LSTM_PERSIST
        LSTM_PERSIST = self.prev
        return False


def lstm_wave_fwd(gx0, w_ih, w_hh, bsum, h0, c0, want_acts: bool = True):
    """The stacked recurrence as a wavefront over (layer, step) (egz_lstm_wave_fwd): gx0 (T,B,4H) = layer 0's input projection
    of every step (bias included); w_ih / w_hh / bsum: lists of L tensors (w_ih[0] / bsum[0] unused); h0, c0 (L,B,H)
    -> hs_ext (L,T+1,B,H) [slot 0 of a layer = its h0, slots 1..T = the outputs], cs (L,T,B,H), acts (L,T,B,4H) | None,
    hn, cn (L,B,H)."""
    _req(gx0, "gx0"); _req(h0, "h0"); _req(c0, "c0")
    L = len(w_hh)
    T, B, H4 = gx0.shape
    Hd = H4 // 4
    for l in range(L):
        _req(w_hh[l], "w_hh")
        if l:
            _req(w_ih[l], "w_ih"); _req(bsum[l], "bias")
            if tuple(w_ih[l].shape) != (H4, Hd):
                raise RuntimeError("lstm_wave_fwd: the upper layers take the hidden size as their input size")
    dev = gx0.device
    hs = torch.empty((L, T + 1, B, Hd), dtype=torch.float32, device=dev)
    cs = torch.empty((L, T, B, Hd), dtype=torch.float32, device=dev)
    acts = torch.empty((L, T, B, H4), dtype=torch.float32, device=dev) if want_acts else None
    hn = torch.empty((L, B, Hd), dtype=torch.float32, device=dev)
    cn = torch.empty_like(hn)
    PROF.note_flops("egz_lstm_wave_fwd", 2.0 * T * B * H4 * Hd * (2 * L - 1))
    check(LIB.egz_lstm_wave_fwd(gx0.data_ptr(), _ptr_table([None] + list(w_ih[1:])), _ptr_table(w_hh),
                                _ptr_table([None] + list(bsum[1:])), h0.data_ptr(), c0.data_ptr(), hs.data_ptr(), cs.data_ptr(),
                                _p(acts), hn.data_ptr(), cn.data_ptr(), L, T, B, Hd, _stream()), "egz_lstm_wave_fwd")
    return hs, cs, acts, hn, cn


def lstm_wave_bwd(dh_top, dhn, dcn, acts, cs, c0, w_hh_t, w_ih_t):
    """Backward through time of the stack -> (dgates (L,T,B,4H), dh0 (L,B,H), dc0 (L,B,H)); w_hh_t / w_ih_t: lists of the
    transposed weights (H,4H) per layer (w_ih_t[0] unused)."""
    L, T, B, Hd = cs.shape
    dev = cs.device
    dgates = torch.empty((L, T, B, 4 * Hd), dtype=torch.float32, device=dev)
    dh0 = torch.empty((L, B, Hd), dtype=torch.float32, device=dev)
    dc0 = torch.empty_like(dh0)
    for name, t in (("dh_top", dh_top), ("dhn", dhn), ("dcn", dcn)):
        if t is not None:
            _req(t, name)
    # the gradient each lower layer receives from the layer above, formed one launch ahead of the cell backward that reads it
    dhin = torch.empty((L - 1, T, B, Hd), dtype=torch.float32, device=dev) if L > 1 else None
    PROF.note_flops("egz_lstm_wave_bwd", 2.0 * (T + 1) * B * 4 * Hd * Hd * L + 2.0 * T * B * 4 * Hd * Hd * (L - 1))
    check(LIB.egz_lstm_wave_bwd(_p(dh_top), _p(dhn), _p(dcn), acts.data_ptr(), cs.data_ptr(), c0.data_ptr(),
                                _ptr_table(w_hh_t), _ptr_table([None] + list(w_ih_t[1:])), dgates.data_ptr(), dh0.data_ptr(),
                                dc0.data_ptr(), _p(dhin), L, T, B, Hd, _stream()), "egz_lstm_wave_bwd")
    return dgates, dh0, dc0


def lstm_persist_ok(L: int, B: int, Hd: int) -> bool:
    """Whether the recurrence runs as the persistent launches (knob on and the geometry they are built for)."""
    if not (LSTM_PERSIST and L == 2 and Hd == 512 and 1 <= B <= 32):
        return False
    # one block per CU, all resident at once (backward: 64 unit slices x ceil(B / 8) batch tiles)
    return torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count >= max(128 * ((B + 15) // 16), 64 * ((B + 7) // 8))


def lstm_persist_fwd(gx0, w_ih, w_hh, b_ih, b_hh, h0, c0, want_acts: bool = True):
    """The stacked recurrence in ONE persistent, weight-stationary launch (egz_lstm_persist_fwd; L = 2, H = 512, B <= 32): gx0
    (T,B,4H) = layer 0's input projection WITHOUT bias; w_ih / w_hh / b_ih / b_hh: lists of L tensors as the module holds them
    (w_ih[0] unused) -> the outputs of lstm_wave_fwd.  The launch's hand-off counters live in a scratch of this call's own (the
    call zeroes them) and sticky error words live in this (device, stream)'s scratch (_persist_sync; lstm_persist_check())."""
    _req(gx0, "gx0"); _req(h0, "h0"); _req(c0, "c0")
    L = len(w_hh)
    T, B, H4 = gx0.shape
    Hd = H4 // 4
    for l in range(L):
        _req(w_hh[l], "w_hh"); _req(b_ih[l], "b_ih"); _req(b_hh[l], "b_hh")
        if l:
            _req(w_ih[l], "w_ih")
    dev = gx0.device
    hs = torch.empty((L, T + 1, B, Hd), dtype=torch.float32, device=dev)
    cs = torch.empty((L, T, B, Hd), dtype=torch.float32, device=dev)
    acts = torch.empty((L, T, B, H4), dtype=torch.float32, device=dev) if want_acts else None
    hn = torch.empty((L, B, Hd), dtype=torch.float32, device=dev)
    cn = torch.empty_like(hn)
    sync = _persist_sync(dev)
    PROF.note_flops("egz_lstm_persist_fwd", 2.0 * T * B * H4 * Hd * (2 * L - 1))
    check(LIB.egz_lstm_persist_fwd(gx0.data_ptr(), _ptr_table([None] + list(w_ih[1:])), _ptr_table(w_hh), _ptr_table(b_ih),
                                   _ptr_table(b_hh), h0.data_ptr(), c0.data_ptr(), hs.data_ptr(), cs.data_ptr(), _p(acts),
                                   hn.data_ptr(), cn.data_ptr(), sync.data_ptr(), L, T, B, Hd, _stream()), "egz_lstm_persist_fwd")
    return hs, cs, acts, hn, cn


def lstm_persist_bwd(dh_top, dhn, dcn, acts, cs, c0, w_hh, w_ih, db=None):
    """Backward through time in ONE persistent launch (egz_lstm_persist_bwd) -> (dgates (L,T,B,4H), dh0, dc0 (L,B,H)); w_hh / w_ih:
    the UNtransposed weights (w_ih[0] unused); db: list of 2 L destinations (b_ih_l0, b_hh_l0, b_ih_l1, b_hh_l1; None entries
    skipped) that receive the bias gradients -- the sums of dgates over steps and batch rows -- from the same launch."""
    L, T, B, Hd = cs.shape
    dev = cs.device
    dgates = torch.empty((L, T, B, 4 * Hd), dtype=torch.float32, device=dev)
    dh0 = torch.empty((L, B, Hd), dtype=torch.float32, device=dev)
    dc0 = torch.empty_like(dh0)
    for name, t in (("dh_top", dh_top), ("dhn", dhn), ("dcn", dcn)):
        if t is not None:
            _req(t, name)
    for l in range(L):
        _req(w_hh[l], "w_hh")
        if l:
            _req(w_ih[l], "w_ih")
    if db is not None:
        for t in db:
            if t is not None and (not t.is_contiguous() or t.numel() != 4 * Hd):
                raise RuntimeError("lstm_persist_bwd: a bias-gradient destination must be a contiguous (4H,) tensor")
    sync = _persist_sync(dev)
    PROF.note_flops("egz_lstm_persist_bwd", 2.0 * (T + 1) * B * 4 * Hd * Hd * L + 2.0 * T * B * 4 * Hd * Hd * (L - 1))
    check(LIB.egz_lstm_persist_bwd(_p(dh_top), _p(dhn), _p(dcn), acts.data_ptr(), cs.data_ptr(), c0.data_ptr(), _ptr_table(w_hh),
                                   _ptr_table([None] + list(w_ih[1:])), dgates.data_ptr(), dh0.data_ptr(), dc0.data_ptr(),
                                   _ptr_table(db) if db is not None else None, sync.data_ptr(), L, T, B, Hd, _stream()),
          "egz_lstm_persist_bwd")
    return dgates, dh0, dc0


def _ptr_table(tensors):
    """Host array of device pointers (None -> NULL) for the C-ABI entry points that take a parameter table."""
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def lstm_b1_fwd(params, inp, h0, c0, want_acts: bool = True):
    """The whole lstmnet step at T = 1, B = 1 in one call (csrc/lstm_b1.hip).  params: state-dict order, 4L + 2 tensors;
    inp (C,) raw; h0, c0 (L,H) -> (xt (C,), acts (L,4H) | None, hn (L,H), cn (L,H), out (N,))."""
    L = (len(params) - 2) // 4
    for i, t in enumerate(params):
        _req(t, f"param{i}")
    _req(inp, "input"); _req(h0, "h0"); _req(c0, "c0")
    C, Hd, N = params[0].shape[1], params[1].shape[1], params[-2].shape[0]
    dev = inp.device
    xt = torch.empty(C, dtype=torch.float32, device=dev)
    acts = torch.empty((L, 4 * Hd), dtype=torch.float32, device=dev) if want_acts else None
    hc = torch.empty((2, L, Hd), dtype=torch.float32, device=dev)     # (hn, cn) in one buffer: a caller can copy both at once
    hn, cn = hc[0], hc[1]
    out = torch.empty(N, dtype=torch.float32, device=dev)
    check(LIB.egz_lstm_b1_fwd(_ptr_table(params), L, inp.data_ptr(), h0.data_ptr(), c0.data_ptr(), xt.data_ptr(), _p(acts),
                              hn.data_ptr(), cn.data_ptr(), out.data_ptr(), C, Hd, N, _stream()), "egz_lstm_b1_fwd")
    return xt, acts, hn, cn, out


def lstm_b1_bwd(params, grads, dout, dhn, dcn, xt, acts, h0, c0, hn, cn, out):
    """Backward of lstm_b1_fwd: fills the tensors in ``grads`` (same order as params; None entries are skipped)."""
    L = (len(params) - 2) // 4
    C, Hd, N = params[0].shape[1], params[1].shape[1], params[-2].shape[0]
    for name, t in (("dout", dout), ("dhn", dhn), ("dcn", dcn)):
        if t is not None:
            _req(t, name)
    nbytes = LIB.egz_lstm_b1_ws_bytes(L, C, Hd, N)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=out.device)
    check(LIB.egz_lstm_b1_bwd(_ptr_table(params), _ptr_table(grads), L, dout.data_ptr(), _p(dhn), _p(dcn), xt.data_ptr(),
                              acts.data_ptr(), h0.data_ptr(), c0.data_ptr(), hn.data_ptr(), cn.data_ptr(), out.data_ptr(),
                              C, Hd, N, ws.data_ptr(), nbytes, _stream()), "egz_lstm_b1_bwd")


def lstm_cell_fwd(gates, c_prev, h_out, c_out, act):
    B, Hd = c_prev.shape
    check(LIB.egz_lstm_cell_fwd(gates.data_ptr(), c_prev.data_ptr(), h_out.data_ptr(), c_out.data_ptr(), _p(act), B, Hd,
                                _stream()), "egz_lstm_cell_fwd")


def lstm_cell_bwd(act, c, c_prev, dh, dc_in, dgates, dc_prev):
    B, Hd = c.shape
    check(LIB.egz_lstm_cell_bwd(act.data_ptr(), c.data_ptr(), c_prev.data_ptr(), dh.data_ptr(), _p(dc_in),
                                dgates.data_ptr(), dc_prev.data_ptr(), B, Hd, _stream()), "egz_lstm_cell_bwd")


def tanh_fwd(x):
    _req(x, "x")
    y = torch.empty_like(x)
    check(LIB.egz_tanh_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "egz_tanh_fwd")
    return y


def tanh_bwd(y, dy):
    dx = torch.empty_like(y)
    check(LIB.egz_tanh_bwd(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel(), _stream()), "egz_tanh_bwd")
    return dx


def add(a, b):
    _req(a, "a"); _req(b, "b")
    out = torch.empty_like(a)
    check(LIB.egz_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "egz_add")
    return out


# ----------------------------------------------------------------------------- validation metric (utils.computeAAEAUC)
_GAUSS_W = {}


def _gauss_weights(device, sigma: float = 14.0, truncate: float = 4.0):
    """The 1-D kernel exactly as scipy.ndimage builds it (_gaussian_kernel1d, order 0): radius = int(truncate*sigma+.5)."""
    key = (torch.device(device).index or 0, sigma, truncate)
    hit = _GAUSS_W.get(key)
    if hit is None:
        import numpy as np
        radius = int(truncate * sigma + 0.5)
        x = np.arange(-radius, radius + 1)
        phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
        phi = phi / phi.sum()
        hit = (torch.from_numpy(phi).to(device), radius)
        _GAUSS_W[key] = hit
    return hit


def aae_auc(out: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """out, gt: (B, 224, 224) fp32 on the GPU -> (B, 6) float64: AAE deg, fp count, gaze row/col, centroid row/col."""
    import math
    _req(out, "output"); _req(gt, "target")
    B, Hh, Ww = out.shape
    gw, radius = _gauss_weights(out.device)
    res = torch.empty((B, 6), dtype=torch.float64, device=out.device)
    check(LIB.egz_aae_auc(out.data_ptr(), gt.data_ptr(), B, Hh, Ww, gw.data_ptr(), radius, 112 / math.tan(math.pi / 6),
                          res.data_ptr(), _stream()), "egz_aae_auc")
    return res


# ----------------------------------------------------------------------------- input pipeline / AT glue
_NORM_CONST = {}


def u8_normalize(src: torch.Tensor, mean, std) -> torch.Tensor:
    """src: uint8 (..., C, H, W) on the GPU -> fp32 (u8 / 255 - mean[c]) / std[c] (bit-exact with the torch expression)."""
    if src.dtype != torch.uint8 or not src.is_cuda or not src.is_contiguous():
        raise ValueError("u8_normalize: contiguous CUDA uint8 tensor expected")
    C, plane = src.shape[-3], src.shape[-2] * src.shape[-1]
    key = (src.device.index or 0, tuple(float(v) for v in mean), tuple(float(v) for v in std))
    hit = _NORM_CONST.get(key)
    if hit is None:
        hit = (torch.tensor(key[1], dtype=torch.float32, device=src.device),
               torch.tensor(key[2], dtype=torch.float32, device=src.device))
        _NORM_CONST[key] = hit
    if hit[0].numel() != C:
        raise ValueError(f"u8_normalize: {C} channels but {hit[0].numel()} mean / std values")
    dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    check(LIB.egz_u8_normalize(src.data_ptr(), dst.data_ptr(), src.numel(), plane, C, hit[0].data_ptr(),
                               hit[1].data_ptr(), _stream()), "egz_u8_normalize")
    return dst


def crop_mean(feat_nhwc: torch.Tensor, gp, size: int, cell: int = 16) -> torch.Tensor:
    """feat_nhwc: (B,H,W,C) fp32; gp: B gaze points (row, col) in input pixels -> chn_weight (B, C)."""
    _req(feat_nhwc, "feature")
    B, Hh, Ww, C = feat_nhwc.shape
    if isinstance(gp, torch.Tensor) and gp.is_cuda and gp.dtype == torch.int32 and gp.is_contiguous() and gp.numel() == 2 * B:
        g = gp                           # already on the device (u8_center_of_mass): no host round trip
    else:
        g = torch.as_tensor(gp, dtype=torch.int32).reshape(B, 2).to(feat_nhwc.device)
    out = torch.empty((B, C), dtype=torch.float32, device=feat_nhwc.device)
    check(LIB.egz_crop_mean(feat_nhwc.data_ptr(), g.data_ptr(), out.data_ptr(), B, Hh, Ww, C, int(size), int(cell),
                            _stream()), "egz_crop_mean")
    return out


def window_mean(feat_nhwc: torch.Tensor, windows) -> torch.Tensor:
    """feat_nhwc: (B,H,W,C) fp32; windows: B x (y0, y1, x0, x1) half-open cell ranges -> (B, C) window means."""
    _req(feat_nhwc, "feature")
    B, Hh, Ww, C = feat_nhwc.shape
    win = [tuple(int(v) for v in w) for w in windows]
    if len(win) != B or any(not (0 <= y0 < y1 <= Hh and 0 <= x0 < x1 <= Ww) for y0, y1, x0, x1 in win):
        raise ValueError(f"window_mean: windows {win} do not fit a {Hh} x {Ww} map of batch {B}")
    g = torch.tensor(win, dtype=torch.int32).to(feat_nhwc.device)
    out = torch.empty((B, C), dtype=torch.float32, device=feat_nhwc.device)
    check(LIB.egz_window_mean(feat_nhwc.data_ptr(), g.data_ptr(), out.data_ptr(), B, Hh, Ww, C, _stream()),
          "egz_window_mean")
    return out


def pixel_weighted_sum(feat_nhwc: torch.Tensor, wmap) -> torch.Tensor:
    """feat_nhwc: (B,H,W,C); wmap: (B,H,W) per-pixel weights (host array / tensor) -> (B, C) = sum_p wmap[p] * feat[p]."""
    _req(feat_nhwc, "feature")
    B, Hh, Ww, C = feat_nhwc.shape
    wm = torch.as_tensor(wmap, dtype=torch.float32).reshape(B, Hh * Ww).contiguous().to(feat_nhwc.device)
    out = torch.empty((B, C), dtype=torch.float32, device=feat_nhwc.device)
    check(LIB.egz_pixel_weighted_sum(feat_nhwc.data_ptr(), wm.data_ptr(), out.data_ptr(), B, Hh * Ww, C, _stream()),
          "egz_pixel_weighted_sum")
    return out


def weighted_minmax(feat_nhwc: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """feat_nhwc: (B,H,W,C), w: (B,C) -> (B,H,W): channel-weighted sum, min-max normalised per image."""
    _req(feat_nhwc, "feature"); _req(w, "chn_weight")
    B, Hh, Ww, C = feat_nhwc.shape
    out = torch.empty((B, Hh, Ww), dtype=torch.float32, device=feat_nhwc.device)
    check(LIB.egz_weighted_minmax(feat_nhwc.data_ptr(), w.data_ptr(), out.data_ptr(), B, Hh * Ww, C, _stream()),
          "egz_weighted_minmax")
    return out


def u8_center_of_mass(maps: torch.Tensor, want_u8: bool = False):
    """maps: (B, H, W) fp32 in [0, 1] -> (com (B, 2) float64, gp (B, 2) int32 = floor(com)[, q (B, H, W) uint8]): the
    reference's ``ndimage.center_of_mass((map * 255).astype(np.uint8))`` (run_spatialstream.py:99-104,130-131), bit for bit."""
    _req(maps, "map")
    B, Hh, Ww = maps.shape
    com = torch.empty((B, 2), dtype=torch.float64, device=maps.device)
    gp = torch.empty((B, 2), dtype=torch.int32, device=maps.device)
    q = torch.empty((B, Hh, Ww), dtype=torch.uint8, device=maps.device) if want_u8 else None
    check(LIB.egz_u8_center_of_mass(maps.data_ptr(), B, Hh, Ww, com.data_ptr(), gp.data_ptr(), _p(q), _stream()),
          "egz_u8_center_of_mass")
    return (com, gp, q) if want_u8 else (com, gp)


def bilinear_up(src: torch.Tensor, scale: int, align_corners: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """src: (B, h, w) -> (B, h*scale, w*scale), ``nn.functional.interpolate(mode='bilinear')`` with either align_corners
    convention.  ``out``: a (B, H, W) destination whose samples may be strided (e.g. ``x[:, 1]`` of a (B, 2, H, W) tensor --
    channel 1 of late_fusion's input) but whose rows are dense."""
    _req(src, "src")
    B, h, w = src.shape
    Hh, Ww = h * scale, w * scale
    if out is None:
        out = torch.empty((B, Hh, Ww), dtype=torch.float32, device=src.device)
    if tuple(out.shape) != (B, Hh, Ww) or out.dtype != torch.float32 or not out.is_cuda or out.stride(2) != 1 or out.stride(1) != Ww:
        raise RuntimeError(f"bilinear_up: destination {tuple(out.shape)} / strides {out.stride()} does not fit {(B, Hh, Ww)}")
    check(LIB.egz_bilinear_up(src.data_ptr(), out.data_ptr(), B, h, w, int(scale), int(bool(align_corners)),
                              out.stride(0) if B > 1 else Hh * Ww, _stream()), "egz_bilinear_up")
    return out
