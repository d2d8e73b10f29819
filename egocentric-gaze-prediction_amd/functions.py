"""``torch.autograd.Function`` nodes of the SP / LF path -- one per *fused block*, each a short
sequence of C-ABI kernel launches (hipops).  Tensors crossing these nodes have the reference's
logical shape (B, C, H, W) but live in NHWC memory (``channels_last`` strides), so the module
surface (models/model_SP.py etc.) is unchanged while the kernels see their native layout.

Block boundaries follow the reference graph:
  ConvBNReLUPool  = Conv2d 3x3 -> BatchNorm2d -> ReLU [-> MaxPool2d(2,2)]        utils.py:64-76
  FusionBlock     = Conv3d (1,3,3) on the 2-deep stack -> MaxPool3d((2,1,1)) -> BN -> ReLU
                                                                                models/model_SP.py:38-47
  ConvReLU        = [Upsample x2 nearest ->] Conv2d 3x3 -> ReLU                  models/model_SP.py:13-29
  HeadSigmoid     = Conv2d 1x1 (C -> 1) -> Sigmoid                               models/model_SP.py:30,49
  FlossLoss       = floss / BCELoss                                              floss.py:9-41
"""
from __future__ import annotations

import torch

from . import hipops as H
from . import streams
from .streams import fork


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """Logical (B,C,H,W) tensor -> contiguous (B,H,W,C) buffer (zero-copy when already channels_last).  The abs-max scalar
    the producing kernel attached to ``x`` (hipops.carry_absmax) follows the values."""
    am = getattr(x, "_egz_absmax", None)
    pre = getattr(x, "_egz_presplit", False)
    if am is not None and getattr(x, "_egz_am_version", x._version) != x._version:
        # written in place since its producer attached the maximum (autograd accumulating a second gradient into it, a hook): the
        # scalar no longer bounds the values -- the consumer takes a standalone reduction / the fp32 route instead
        if pre:
            raise RuntimeError("a pre-split tensor was modified in place (it holds f16 pairs, not fp32 values)")
        am = None
    x = x.detach()
    xp = x.permute(0, 2, 3, 1)
    if not xp.is_contiguous():
        if pre:
            raise RuntimeError("a pre-split activation was re-laid out (it holds f16 pairs, not fp32 values)")
        if not x.is_contiguous():
            x = x.contiguous()
        xp = H.nchw_to_nhwc(x)
    if am is not None:
        xp._egz_absmax = am
    if pre:
        xp._egz_presplit = True
    return xp


def from_nhwc(y: torch.Tensor) -> torch.Tensor:
    res = H.carry_absmax(y.permute(0, 3, 1, 2), y)
    if getattr(res, "_egz_absmax", None) is not None:
        res._egz_am_version = res._version          # to_nhwc drops the scalar when the tensor has been written since
    return res


BIAS_ON_HELPER = True     # decoder bias gradients (column sums of the masked-dgrad stat rows) on the weight-gradient helper stream
DETACH_WGRAD = True       # weight-gradient forks that end in a gradient sink are left running (streams.fork.detach)
# (Issuing the weight gradient W_L BEHIND the data gradient D_L instead of in front of it measured no difference -- 35.58 vs
# 35.57 ms per step, profiles/r02_bench_ab_knobs.txt: with detached weight-gradient streams the helper queue is never empty --
# and the switch was removed in round 4.)


def _close_fork(f, sink, dw, *inputs):
    """End of a forked weight-gradient computation.  With a gradient sink the result is not consumed by autograd, so the
    helper stream is left running (streams.fork.detach): the weight gradient of layer L then overlaps the HBM-bound
    BN / ReLU backward pass and the data gradient of layer L-1 instead of holding the chain up.  Without a sink the
    gradient tensor goes back to autograd and the streams are joined."""
    if sink is not None and DETACH_WGRAD and not (f.enabled and torch.cuda.is_current_stream_capturing()):
        f.detach(*inputs)
        H.PENDING_PRODUCER[0] = f.side if f.enabled else None      # the stream the weight gradient is being written on
        _join_at_end_of_backward()
    else:
        f.join(dw)            # (inside a hipGraph capture every fork joins back: a capture must end with one open stream)


_JOIN_QUEUED = [False]


def _join_at_end_of_backward():
    """A detached weight-gradient stream is unknown to the autograd engine, so ``loss.backward()`` would return with
    ``p.grad`` still being written on it: reading a gradient, clipping, or ``zero_grad()`` without a ``step()`` in between
    would race.  The first detach of a backward pass queues ONE engine callback that makes the stream
    ``backward()`` was called on wait for every helper stream once the whole graph has been issued -- the same join the
    optimizer step performs, so the step itself is unchanged (measured), but gradients are ordinary stream-ordered tensors
    again when ``backward()`` returns.  Skipped inside a hipGraph capture (a capture may only join streams forked from it)."""
    if _JOIN_QUEUED[0] or not streams.ENABLED:
        return
    _JOIN_QUEUED[0] = True

    def _join():
        _JOIN_QUEUED[0] = False
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            streams.join_all_into_current(include_comm=False)

    try:
        torch.autograd.Variable._execution_engine.queue_callback(_join)
    except RuntimeError:          # not inside a backward pass (a test calling a backward function by hand)
        _JOIN_QUEUED[0] = False


def _finish(param, sink, grad):
    """End of a parameter-gradient computation: with a sink the gradient already sits in the optimizer's flat buffer --
    run the sink hooks and hand ``None`` to autograd; without one, return the tensor for autograd to accumulate."""
    if sink is None:
        return grad
    H.grad_done(param)
    return None


def _zero_bias_grad(dy: torch.Tensor, K: int) -> torch.Tensor:
    """Gradient of a conv bias that feeds a train-mode BatchNorm.  BN subtracts the batch mean, so the loss
    does not depend on that bias: sum(dy) = scale * (sum dz - N*mean(dz) - mean(dz*xhat) * sum(xhat)) = 0
    identically.  The reference's autograd evaluates that sum in fp32 and gets round-off noise (~1e-9,
    tests/test_oracle_golden.py); we return the exact value instead of spending an HBM pass on noise."""
    z = torch.empty((K,), dtype=torch.float32, device=dy.device)
    H.fill_zero(z)
    return z


def _first_conv_on_split(C: int, K: int) -> bool:
    """First conv of a stack on the split-half kernels after padding Cin to 32: pays for the wide flow stack (C = 20),
    not for RGB (C = 3, where 29 of 32 padded channels would be wasted work)."""
    return H.PRECISION == "split" and 16 <= C <= 32 and K % 64 == 0


class ConvBNReLUPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, training, momentum, eps, pool,
                first, out_buf=None, nbt=None, next_k=0):
        """``out_buf``: optional (B, H', W', K) NHWC destination of the block output (model_SP hands the two encoders the
        halves of ONE buffer so that the fusion conv reads the depth-2 stack without a concatenation copy).
        ``next_k``: filters of the conv-BN-ReLU block that consumes this block's output inside the same FusedSequential (0 =
        something else does).  In training, when the geometry allows (hipops.bn_defer_ok: the late-fusion widths), this
        block then leaves its [BN -> ReLU] to that block: it returns its PRE-BN conv output tagged with the batch
        coefficients, and the next block's conv / weight-gradient kernels normalise while staging it."""
        K, C = weight.shape[0], weight.shape[1]
        padded = first and _first_conv_on_split(C, K)
        bn_src = None       # (pre-BN conv output, batch coefficients) of the [BN -> ReLU] block that produced x (narrow layers)
        bn_in = getattr(x, "_egz_bn_defer", None)      # x is a deferred (pre-BN) tensor: the coefficients to apply on load
        if bn_in is not None and (first or not training):
            raise RuntimeError("a deferred-BatchNorm tensor reached a block that cannot normalise it")
        Bx, Hx, Wx = x.shape[0], x.shape[2], x.shape[3]
        defer = bool(training and not pool and next_k and out_buf is None and not padded and (C <= 32 or first)
                     and (not first or (C <= 3 and K == 32))
                     and H.bn_defer_ok(Bx, Hx, Wx, K, next_k, first))
        # x holds pre-split activations (the block below stored f16 pairs, hipops.PRESPLIT): this block's convolution and weight
        # gradient stage them without a split
        pre_in = bool(getattr(x, "_egz_presplit", False))
        if pre_in and (first or padded or not training or bn_in is not None):
            raise RuntimeError("a pre-split activation reached a block that cannot take it")
        # ... and this block's own output goes out pre-split when its consumer is the next conv-BN-ReLU block of the stack and
        # that block's launches can take it (the wide VGG layers, utils.py:64-76)
        presplit_out = bool(training and next_k and out_buf is None and not defer
                            and H.presplit_ok(Bx, Hx // 2 if pool else Hx, Wx // 2 if pool else Wx, K, next_k))
        # the per-channel max / min of y: bounds this block's output (presplit_out) and, in the backward pass, its gradient
        want_mm = bool(presplit_out or (training and not defer and not first and not padded
                                        and H.presplit_grad_ok(Bx, Hx, Wx, C, K)))
        if padded:
            # the 20-channel flow stack, zero-padded to 32 channels, on the split-half kernels (2.2 -> ~0.9 ms per step
            # for this one layer); the packed weight is padded the same way by the pack kernel
            xin = H.take_prepared_input(x, 32)              # (re-laid out ahead of time by the driver: hipops.prepare_network_input)
            if xin is None:
                xin = H.nchw_to_nhwc_pad(H._req(x.detach(), "network input (NCHW)"), 32)
            wp, st = H.conv_weight(weight, "fwd", H.F16X3, xin, K)
            H.ALGO_CHANNELS[0] = C
            try:
                y, stat = H.conv3x3_fwd(xin, wp, bias.detach() if bias is not None else None, K, ups=False,
                                        epi=H.EPI_BIAS_STATS if training else H.EPI_BIAS, dtype=H.F16X3, streamed=st,
                                        want_bound=want_mm)
            finally:
                H.ALGO_CHANNELS[0] = None
        elif first:
            xin = H._req(x.detach(), "network input (NCHW)")
            y, stat = H.conv_first_fwd(xin, weight.detach(), bias.detach() if bias is not None else None, training,
                                       want_minmax=defer, want_bound=want_mm)
        else:
            bn_src = getattr(x, "_egz_bn_src", None) if training else None
            xin = to_nhwc(x)
            dt = H.conv_dtype("fwd", K, C, xin)
            fold = (H.EVAL_FOLD and not training and not pool and out_buf is None and bn_in is None and dt == H.F16X3
                    and K % 64 == 0 and H.INFER_CALL)       # (grad mode is always off inside forward(): the caller's mode)
            if fold and torch.cuda.is_current_stream_capturing() and not H.bn_fold_is_warm(
                    weight, bias, gamma, beta, running_mean, running_var, eps):
                fold = False                 # (a capture with a cold or STALE fold cache: the unfolded path is capturable)
            if fold:
                # inference: BatchNorm folded into the convolution, bias + ReLU epilogue -- no normalise pass (hipops.bn_folded_conv)
                wf, bf = H.bn_folded_conv(weight, bias, gamma, beta, running_mean, running_var, eps)
                wpf, stf = H.conv_weight(wf, "fwd", dt, xin, K)
                if stf:
                    out, _ = H.conv3x3_fwd(xin, wpf, bf, K, ups=False, epi=H.EPI_BIAS_RELU, dtype=dt, streamed=True)
                    H.EVAL_FOLD_STATS["folded"] += 1
                    ctx.cfg = (training, pool, first, C, K, padded)
                    ctx.folded = True
                    return from_nhwc(out)
            wp, st = H.conv_weight(weight, "fwd", dt, xin, K)
            if (bn_in is not None or defer) and not (dt and st):
                raise RuntimeError("deferred BatchNorm: the convolution did not land on the streamed split-half kernel")
            y, stat = H.conv3x3_fwd(xin, wp, bias.detach() if bias is not None else None, K, ups=False,
                                    epi=H.EPI_BIAS_STATS if training else H.EPI_BIAS, dtype=dt, streamed=st,
                                    bn_in=bn_in, want_minmax=defer, pre_in=pre_in, want_bound=want_mm)
        B, Hh, Ww, _ = y.shape
        if training and not first and C <= 32 and K <= 32 and ctx.needs_input_grad[0]:
            # narrow (late-fusion) layer: build the data-gradient packing now -- in the backward pass the 5 us pack launch sits
            # on the critical path behind an already running weight-gradient kernel (58 us there, profiles/r03_lf_timeline)
            dtb = H.conv_dtype("dgrad", C, K, y)
            if dtb:
                H.conv_weight(weight, "dgrad", dtb, y, C)
        if bn_src is not None and not H.bnsums_ok(B, Hh, Ww, C, K, H.conv_dtype("dgrad", C, K, y)):
            bn_src = None
        if defer:
            # [BN -> ReLU] left to the next block: finalize the statistics and bound the output this block never writes
            coef, am = H.bn_finalize_deferred(stat, y._egz_minmax, float(B * Hh * Ww), gamma.detach(), beta.detach(),
                                              running_mean, running_var, momentum, eps, nbt)
            out = y
            out._egz_absmax = am
        else:
            mm = getattr(y, "_egz_mm", None) if presplit_out else None      # (absent: the conv took a route without the bound)
            pam = None
            if training:
                coef = H.bn_finalize(stat, float(B * Hh * Ww), gamma.detach(), beta.detach(), running_mean,
                                     running_var, momentum, eps, nbt, mm=mm)
                if mm is not None:
                    coef, pam = coef
            else:
                coef = H.bn_eval_coeffs(gamma, beta, running_mean, running_var, eps)
            out = H.bn_relu_pool_fwd(y, coef, pool, out=out_buf, presplit_am=pam)
        # max |xin| (left on xin by its producer, or by the conv launch above): the weight gradient scales x with it
        ctx.save_for_backward(xin, y, coef, weight, bias, gamma, beta, getattr(xin, "_egz_absmax", None),
                              *(bn_src if bn_src is not None else (None, None)))
        ctx.cfg = (training, pool, first, C, K, padded)
        ctx.deferred_in = bn_in is not None
        ctx.x_pre = pre_in
        ctx.y_mm = getattr(y, "_egz_mm", None) if training else None
        if bn_in is not None and bn_src is None:
            ctx.deferred_coef = bn_in                  # (no fused BN sums: the weight gradient still needs the coefficients)
        res = from_nhwc(out)
        if training and not pool and H.BNSUMS_FUSE:
            # the block above (if it is a narrow conv) folds this BatchNorm's backward sums into its data-gradient kernel
            res._egz_bn_src = (y, coef)
        if defer:
            res._egz_bn_defer = coef
        return res

    @staticmethod
    def backward(ctx, dout):
        xin, y, coef, weight, bias, gamma, beta, xam, bn_y, bn_coef = ctx.saved_tensors
        H.carry_absmax(xin, xam)
        training, pool, first, C, K, padded = ctx.cfg
        if not training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not part of the reference path "
                                      "(SP.py:119 trains in model.train(); eval runs under torch.no_grad())")
        ng = ctx.needs_input_grad
        sg, sb = H.grad_sink(gamma, ng[3]), H.grad_sink(beta, ng[4])
        sums = getattr(dout, "_egz_bnsums", None)       # produced with dout by the narrow dgrad kernel of the block above
        if sums is not None and (pool or sums[1] != dout._version or sums[0].shape[1:] != (2, K)):
            sums = None                                 # (an accumulated / modified gradient: the sums no longer describe it)
        sums = None if sums is None else sums[0]
        dx = dw = db = None
        sbias = H.grad_sink(bias, ng[2])           # analytically zero: the sink keeps zero_grad()'s zeros
        if ng[2] and sbias is None:
            db = _zero_bias_grad(y, K)
        sw = H.grad_sink(weight, ng[1] and not padded)
        if first and not padded and ng[1] and not ng[0] and H.bn_bwd_first_wgrad_ok(C, K, pool):
            # first block of the late-fusion stack: BatchNorm backward and the conv's weight gradient in ONE pass over
            # (y, dout) -- the gradient w.r.t. the conv output is consumed in registers and never stored
            dw, dgamma, dbeta = H.bn_bwd_first_wgrad(y, to_nhwc(dout), coef, xin, out_dgamma=sg, out_dbeta=sb, out_dw=sw,
                                                     sums=sums)
            return (None, _finish(weight, sw, dw), _finish(bias, sbias, db), _finish(gamma, sg, dgamma if ng[3] else None),
                    _finish(beta, sb, dbeta if ng[4] else None), None, None, None, None, None, None, None, None, None, None)
        dn = to_nhwc(dout)
        # dy as pre-split pairs (hipops.PRESPLIT_GRAD): needs max |dout| (left on dout by the data-gradient kernel that produced
        # it) and the max / min of y (kept from the forward pass) for the bound, and consumers that can take the pairs
        dout_am = getattr(dn, "_egz_absmax", None)
        gpre = bool(not first and not padded and ctx.y_mm is not None and dout_am is not None
                    and H.presplit_grad_ok(y.shape[0], y.shape[1], y.shape[2], C, K))
        dy, dgamma, dbeta = H.bn_relu_pool_bwd(y, dn, coef, pool, out_dgamma=sg, out_dbeta=sb, sums=sums,
                                               presplit=(dout_am, ctx.y_mm) if gpre else None)

        def data_grad():
            if not ng[0]:
                return None
            if first:
                raise NotImplementedError("gradient w.r.t. the network input is not needed by the reference path")
            dt = H.conv_dtype("dgrad", C, K, dy)
            wp, st = H.conv_weight(weight, "dgrad", dt, dy, C)
            if bn_y is not None and st and dt:
                # narrow layer on top of a [BN -> ReLU] block: the same launch accumulates that BatchNorm's backward sums
                dxn, bsum = H.conv3x3_dgrad_bnsums(dy, wp, C, dt, bn_y, bn_coef, pre_in=gpre)
                res = from_nhwc(dxn)
                res._egz_bnsums = (bsum, res._version)
                return res
            return from_nhwc(H.conv3x3_dgrad(dy, wp, C, dtype=dt, streamed=st, pre_in=gpre))

        with fork("wgrad") as f:                # the weight gradient runs on a helper stream (both only read dy)
            if ng[1]:
                if padded:
                    H.ALGO_CHANNELS[0] = C
                    try:
                        dw = H.conv3x3_wgrad(xin, dy)[:, :C].contiguous()      # gradient of the zero-padded channels dropped
                    finally:
                        H.ALGO_CHANNELS[0] = None
                else:
                    if first:
                        dw = H.conv_first_wgrad(xin, dy, out=sw)
                    else:       # deferred input: xin is the pre-BN tensor of the block below, normalised while it is staged
                        x_bn = None
                        if ctx.deferred_in:
                            x_bn = bn_coef if bn_coef is not None else ctx.deferred_coef
                        dw = H.conv3x3_wgrad(xin, dy, out=sw, x_bn=x_bn, x_pre=ctx.x_pre, dy_pre=gpre)
        dx = data_grad()
        # (the coefficient rows of a deferred input are read by the detached weight-gradient kernel too: keep them alive for it)
        _close_fork(f, sw, dw, xin, dy, (bn_coef if bn_coef is not None else getattr(ctx, "deferred_coef", None))
                    if ctx.deferred_in else None)
        return (dx, _finish(weight, sw, dw), _finish(bias, sbias, db), _finish(gamma, sg, dgamma if ng[3] else None),
                _finish(beta, sb, dbeta if ng[4] else None), None, None, None, None, None, None, None, None, None, None)


class ConvReLU(torch.autograd.Function):
    """[Upsample x2 ->] Conv2d 3x3 -> ReLU.  ``relu_below``: the input is the output of another ConvReLU block (decoder
    chains, models/model_SP.py:13-29).  Then this node's data gradient IS the gradient w.r.t. that block's post-ReLU output,
    and the ReLU mask of the block below (input > 0), its bias gradient (column sums of the masked gradient) and the abs-max
    of the result are produced in the epilogue of this node's dgrad kernel (hipops.conv3x3_dgrad_masked); the gradient
    tensor handed down carries them (``_egz_premasked``), and the block below skips its own ReLU-backward pass."""

    @staticmethod
    def forward(ctx, x, weight, bias, ups, relu_below=False):
        K, C = weight.shape[0], weight.shape[1]
        xin = to_nhwc(x)
        dt = H.conv_dtype("fwd", K, C, xin)
        wp, st = H.conv_weight(weight, "ups_fwd" if ups else "fwd", dt, xin, K)
        y, _ = H.conv3x3_fwd(xin, wp, bias.detach() if bias is not None else None, K, ups="phase" if ups else False,
                             epi=H.EPI_BIAS_RELU, dtype=dt, streamed=st)
        ctx.save_for_backward(xin, y, weight, bias, getattr(xin, "_egz_absmax", None))
        ctx.cfg = (ups, C, K, bool(relu_below))
        return from_nhwc(y)

    @staticmethod
    def backward(ctx, dout):
        xin, y, weight, bias, xam = ctx.saved_tensors
        H.carry_absmax(xin, xam)
        ups, C, K, relu_below = ctx.cfg
        ng = ctx.needs_input_grad
        dx = dw = db = None
        sbias = H.grad_sink(bias, ng[2])
        pre = getattr(dout, "_egz_premasked", None)
        bias_stat = None
        if pre is not None and tuple(dout.shape) == (y.shape[0], K, y.shape[1], y.shape[2]):
            # the block above already applied this block's ReLU mask in its dgrad epilogue
            dy = to_nhwc(dout)
            stat, am = pre
            H.MASK_FUSE_STATS["consumed"] += 1
            if am is not None:
                dy._egz_absmax = am
            # (the bias gradient -- column sums of the stat rows the dgrad kernel above left -- is issued on the weight-gradient
            # helper stream below: on this stream its two small launches sat between two data-gradient launches of the serial
            # decoder chain with nothing beside them, ~20 us per layer; profiles/r06_step_windows.txt)
            bias_stat = stat if ng[2] else None
            if bias_stat is not None and not BIAS_ON_HELPER:
                db = H.colsum_f64(bias_stat, K, out=sbias)
                bias_stat = None
        elif ng[2]:
            dy, db = H.relu_bwd_bias(y, to_nhwc(dout), out_db=sbias)
        else:
            dy = H.relu_bwd(y, to_nhwc(dout))
        sw = H.grad_sink(weight, ng[1])

        def data_grad():
            if not ng[0]:
                return None
            dt = H.conv_dtype("dgrad", C, K, dy)
            role = "ups_dgrad" if ups else "dgrad"
            wp, st = H.conv_weight(weight, role, dt, dy, C)
            if relu_below and st and H.MASK_FUSE and C % 64 == 0:
                dxn, stat, am = H.conv3x3_dgrad_masked(dy, wp, C, dt, xin, ups)
                d = from_nhwc(dxn)
                d._egz_premasked = (stat, am if H._want_absmax() else None)
                return d
            if ups:       # gradient w.r.t. the low-res input directly (4x4 / stride-2 gather over dy)
                return from_nhwc(H.conv3x3_ups_dgrad(dy, wp, C, dtype=dt, streamed=st))
            return from_nhwc(H.conv3x3_dgrad(dy, wp, C, dtype=dt, streamed=st))

        with fork("wgrad") as f:
            if bias_stat is not None:
                db = H.colsum_f64(bias_stat, K, out=sbias)
            if ng[1]:
                dw = H.conv3x3_wgrad(xin, dy, ups=ups, out=sw)
        dx = data_grad()
        _close_fork(f, sw, dw, xin, dy, bias_stat)
        side = H.PENDING_PRODUCER[0]                      # the helper stream both gradients are being written on (detached fork)
        gw = _finish(weight, sw, dw)
        if bias_stat is not None:
            if sbias is not None:
                H.PENDING_PRODUCER[0] = side              # (the bias sink's hooks must see the same producer stream)
            elif f.enabled:
                f.join(db)                                # no sink: the tensor goes back to autograd on this stream
        return dx, gw, _finish(bias, sbias, db), None, None


class FusionBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fs, ft, weight, bias, gamma, beta, running_mean, running_var, training, momentum, eps, nbt=None):
        K, C = weight.shape[0], weight.shape[1]
        a, b = to_nhwc(fs), to_nhwc(ft)
        if (a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
                and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
                and b.data_ptr() == a.data_ptr() + 4 * a.numel()):
            # the encoders wrote the two halves of one buffer (model_SP.forward): the depth-2 stack, folded into the
            # batch dim, already exists -- no concatenation copy
            x2 = a.as_strided((2 * a.shape[0],) + tuple(a.shape[1:]), a.stride())
        else:
            x2 = H.stack2(a, b)
        dt = H.conv_dtype("fwd", K, C, x2)
        wp, st = H.conv_weight(weight, "fwd", dt, x2, K)
        y2, _ = H.conv3x3_fwd(x2, wp, bias.detach() if bias is not None else None, K, ups=False, epi=H.EPI_BIAS, dtype=dt,
                              streamed=st)
        z = H.pairmax_fwd(y2)
        B, Hh, Ww, _ = z.shape
        if training:
            coef = H.bn_finalize(H.channel_stats(z), float(B * Hh * Ww), gamma.detach(), beta.detach(),
                                 running_mean, running_var, momentum, eps, nbt)
        else:
            coef = H.bn_eval_coeffs(gamma, beta, running_mean, running_var, eps)
        out = H.bn_relu_pool_fwd(z, coef, False)
        ctx.save_for_backward(x2, y2, z, coef, weight, bias, gamma, beta, getattr(x2, "_egz_absmax", None))
        ctx.cfg = (training, C, K)
        return from_nhwc(out)

    @staticmethod
    def backward(ctx, dout):
        x2, y2, z, coef, weight, bias, gamma, beta, xam = ctx.saved_tensors
        H.carry_absmax(x2, xam)
        training, C, K = ctx.cfg
        if not training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not part of the reference path")
        ng = ctx.needs_input_grad
        sg, sb = H.grad_sink(gamma, ng[4]), H.grad_sink(beta, ng[5])
        dz, dgamma, dbeta = H.bn_relu_pool_bwd(z, to_nhwc(dout), coef, False, out_dgamma=sg, out_dbeta=sb)
        dy2 = H.pairmax_bwd(y2, dz)
        dfs = dft = dw = db = None
        sbias = H.grad_sink(bias, ng[3])
        if ng[3] and sbias is None:
            db = _zero_bias_grad(dy2, K)
        sw = H.grad_sink(weight, ng[2])

        def data_grad():
            if not (ng[0] or ng[1]):
                return None, None
            dt = H.conv_dtype("dgrad", C, K, dy2)
            wp, st = H.conv_weight(weight, "dgrad", dt, dy2, C)
            dx2 = H.conv3x3_dgrad(dy2, wp, C, dtype=dt, streamed=st)
            B = dx2.shape[0] // 2
            # (max |dx2| over BOTH halves rides on each: an upper bound is all the encoders' last BatchNorm backward needs)
            return from_nhwc(H.carry_absmax(dx2[:B], dx2)), from_nhwc(H.carry_absmax(dx2[B:], dx2))

        with fork("wgrad") as f:
            if ng[2]:
                dw = H.conv3x3_wgrad(x2, dy2, out=sw).view(weight.shape)
        dfs, dft = data_grad()
        _close_fork(f, sw, dw, x2, dy2)
        return (dfs, dft, _finish(weight, sw, dw), _finish(bias, sbias, db),
                _finish(gamma, sg, dgamma if ng[4] else None), _finish(beta, sb, dbeta if ng[5] else None),
                None, None, None, None, None, None)


class HeadSigmoid(torch.autograd.Function):
    """Conv2d 1x1 (C -> 1) -> Sigmoid.  ``relu_below``: the input is the output of a ConvReLU block (the decoder's last
    3x3 conv, models/model_SP.py:28-32); its ReLU backward, bias-gradient sums and abs-max are then taken in this node's
    backward kernel and handed down as ``_egz_premasked`` (see ConvReLU) instead of a 1.2 GB pass of their own."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu_below=False):
        xin = to_nhwc(x)
        out, _ = H.conv1x1_sigmoid_fwd(xin, H._req(weight.detach(), "weight"), bias.detach() if bias is not None else None)
        ctx.save_for_backward(xin, out, weight, bias)
        ctx.relu_below = bool(relu_below)
        B, Hh, Ww = out.shape
        return out.view(B, 1, Hh, Ww)

    @staticmethod
    def backward(ctx, dout):
        xin, out, weight, bias = ctx.saved_tensors
        ng = ctx.needs_input_grad
        sw, sb = H.grad_sink(weight, ng[1]), H.grad_sink(bias, ng[2] and bias is not None)
        if ctx.relu_below and ng[0] and H.MASK_FUSE and H.HEAD_MASK_FUSE:
            dx, dw, db, stat, am = H.conv1x1_sigmoid_bwd_masked(xin, weight.detach(), out, dout.contiguous(), out_dw=sw,
                                                                out_db=sb)
            d = from_nhwc(dx)
            d._egz_premasked = (stat, am if H._want_absmax() else None)
            return d, _finish(weight, sw, dw if ng[1] else None), _finish(bias, sb, db if ng[2] else None), None
        dx, dw, db = H.conv1x1_sigmoid_bwd(xin, weight.detach(), out, dout.contiguous(), need_dx=ng[0], out_dw=sw,
                                           out_db=sb)
        return (from_nhwc(dx) if dx is not None else None, _finish(weight, sw, dw if ng[1] else None),
                _finish(bias, sb, db if ng[2] else None), None)


class FlossLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, target, weighted):
        x = H._req(inp.detach().contiguous(), "input")
        t = H._req(target.detach().contiguous(), "target")
        if x.shape != t.shape:
            raise ValueError(f"Using a target size ({tuple(t.shape)}) that is different to the input size "
                             f"({tuple(x.shape)}) is deprecated. Please ensure they have the same size.")
        loss, weights = H.floss_fwd(x, t, weighted)
        ctx.save_for_backward(x, t, weights if weights is not None else torch.empty(0, device=x.device))
        ctx.weighted = weighted
        return loss

    @staticmethod
    def backward(ctx, gout):
        x, t, weights = ctx.saved_tensors
        g = gout.detach().contiguous().to(torch.float32)
        return H.floss_bwd(x, t, weights if ctx.weighted else None, g), None, None


class Cat2Planes(torch.autograd.Function):
    """``torch.cat((f, g), 1)`` of two (B, 1, H, W) maps as one kernel (models/late_fusion.py:19).  The gradient of a
    concatenation is its two halves: views of the incoming gradient, no kernel."""

    @staticmethod
    def forward(ctx, f, g):
        if f.dim() != 4 or f.shape != g.shape or f.shape[1] != 1:
            raise RuntimeError(f"late_fusion: two (B, 1, H, W) maps expected, got {tuple(f.shape)} and {tuple(g.shape)}")
        return H.cat2_planes(f.detach().contiguous(), g.detach().contiguous())

    @staticmethod
    def backward(ctx, dout):
        return dout[:, 0:1], dout[:, 1:2]


class MSELoss(torch.autograd.Function):
    """nn.MSELoss (AT.py:83).  ``tanh_target``: the loss against tanh(target) -- the reference's
    ``criterion(pred, tanh(target))`` (AT.py:138) without a separate tanh pass over the target."""

    @staticmethod
    def forward(ctx, a, b, tanh_target=False):
        a_ = H._req(a.detach().contiguous(), "input")
        b_ = H._req(b.detach().contiguous(), "target")
        ctx.save_for_backward(a_, b_)
        ctx.tanh_target = bool(tanh_target)
        return H.mse_fwd(a_, b_, ctx.tanh_target)

    @staticmethod
    def backward(ctx, gout):
        a_, b_ = ctx.saved_tensors
        return H.mse_bwd(a_, b_, gout.detach().contiguous(), ctx.tanh_target), None, None
