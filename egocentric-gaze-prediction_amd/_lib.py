"""ctypes binding of the C-ABI shared library ``csrc/libegaze_hip.so`` (see include/egaze_hip.h).

There is NO fallback: if the library is missing or does not export a declared symbol the import of
any compute module raises, and every op raises on non-HIP tensors.  Build it with
``python __graft_entry__.py`` (or ``csrc/build.sh``); hipcc cross-compiles gfx950 without a GPU.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# EGAZE_HIP_LIB: load another build of the same C-ABI (kernel A/B runs: csrc/build.sh -D... -o variant); default in-tree .so
LIB_PATH = os.environ.get("EGAZE_HIP_LIB") or os.path.join(_HERE, "csrc", "libegaze_hip.so")

P = c_void_p          # device pointer
S = c_void_p          # hipStream_t

# name -> (restype, argtypes).  Must list every symbol include/egaze_hip.h declares
# (tests/test_cabi_symbols.py cross-checks the header against this table and the .so).
SIGNATURES = {
    "egz_version": (c_char_p, []),
    "egz_last_error": (c_char_p, []),
    "egz_mfma_probe": (c_int, [P, P, c_int, c_int, S]),
    "egz_u8_center_of_mass": (c_int, [P, c_int, c_int, c_int, P, P, P, S]),
    "egz_bilinear_up": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_long, S]),
    # --- 3x3 conv, implicit GEMM on f32 MFMA
    "egz_pack_w3x3_elems": (c_size_t, [c_int, c_int, c_int]),
    "egz_pack_w3x3_ups_fwd": (c_int, [P, P, c_int, c_int, S]),
    "egz_pack_w3x3_ups_dgrad": (c_int, [P, P, c_int, c_int, S]),
    "egz_conv3x3_ups_dgrad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, S]),
    "egz_pack_w3x3_fwd": (c_int, [P, P, c_int, c_int, S]),
    "egz_pack_w3x3_dgrad": (c_int, [P, P, c_int, c_int, S]),
    "egz_conv3x3_stat_rows": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "egz_conv3x3_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, S]),
    "egz_pack_w3x3_split": (c_int, [P, P, c_int, c_int, c_int, c_int, S]),
    "egz_conv3x3_fwd_split_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "egz_conv3x3_fwd_split": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P, P, S]),
    "egz_conv3x3_streamed_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "egz_pack_w3x3_split_frag": (c_int, [P, P, c_int, c_int, c_int, c_int, S]),
    "egz_pack_w3x3_frag_blocks": (c_int, [c_int, c_int]),
    "egz_pack_w3x3_frag_batch": (c_int, [P, c_int, c_int, S]),
    "egz_conv3x3_streamed_splits": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "egz_conv3x3_fwd_streamed_splitk_stat_rows": (c_int, [c_int, c_int, c_int]),
    "egz_conv3x3_fwd_streamed_splitk_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "egz_conv3x3_fwd_streamed_splitk": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_size_t, c_int, P, S]),
    "egz_conv3x3_streamed_stat_rows": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "egz_conv3x3_fwd_streamed": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, S]),
    "egz_colsum_f64": (c_int, [P, c_int, c_int, c_int, P, P, c_size_t, S]),
    "egz_conv3x3_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "egz_conv3x3_wgrad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P, P, P, S]),
    "egz_conv3x3_wgrad_narrow_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "egz_conv3x3_wgrad_presplit_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    # --- first encoder conv (NCHW input, Cin 3 / 20)
    "egz_conv_first_stat_rows": (c_int, [c_int, c_int, c_int]),
    "egz_conv_first_stat_rows_for": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "egz_conv_first_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, S]),
    "egz_conv_first_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "egz_conv_first_wgrad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, S]),
    # --- BatchNorm / ReLU / pool / fusion max / misc streaming passes
    "egz_bn_ws_bytes": (c_size_t, [c_int]),
    "egz_bn_finalize": (c_int, [P, c_int, c_int, c_double, P, P, P, P, c_float, c_float, P, P, P, P, P, P,
                                c_size_t, S]),
    "egz_bn_finalize_deferred": (c_int, [P, c_int, c_int, c_double, P, P, P, P, c_float, c_float, P, P, P, P, P, P,
                                         c_int, P, S]),
    "egz_bn_finalize_bound": (c_int, [P, c_int, c_int, c_double, P, P, P, P, c_float, c_float, P, P, P, P, P, P,
                                      c_size_t, P, P, S]),
    "egz_bn_eval_coeffs": (c_int, [c_int, P, P, P, P, c_float, P, P, S]),
    "egz_bn_relu_pool_fwd_presplit": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, S]),
    "egz_bn_relu_pool_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, S]),
    "egz_bn_relu_pool_bwd_ws_bytes": (c_size_t, [c_int]),
    "egz_bn_relu_pool_bwd": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P,
                                     c_size_t, P, P, c_int, S]),
    "egz_bn_relu_pool_bwd_presplit": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P,
                                              c_size_t, P, P, c_int, P, P, S]),
    "egz_bn_bwd_first_wgrad_ws_bytes": (c_size_t, [c_int, c_int]),
    "egz_bn_bwd_first_wgrad": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P, c_int, S]),
    "egz_pairmax_fwd": (c_int, [P, P, c_long, S]),
    "egz_pairmax_bwd": (c_int, [P, P, P, c_long, P, S]),
    "egz_absmax_elems": (c_int, []),
    "egz_absmax": (c_int, [P, c_long, P, S]),
    "egz_channel_stats_rows": (c_int, []),
    "egz_channel_stats": (c_int, [P, c_long, c_int, P, S]),
    "egz_relu_bwd": (c_int, [P, P, P, c_long, S]),
    "egz_relu_bwd_bias_ws_bytes": (c_size_t, [c_int]),
    "egz_relu_bwd_bias": (c_int, [P, P, P, P, c_long, c_int, P, c_size_t, P, S]),
    "egz_upsample2x_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, S]),
    "egz_colsum": (c_int, [P, c_long, c_int, P, P, c_size_t, S]),
    "egz_nchw_to_nhwc": (c_int, [P, P, c_int, c_int, c_int, c_int, S]),
    "egz_nchw_to_nhwc_pad": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, S]),
    "egz_nhwc_to_nchw": (c_int, [P, P, c_int, c_int, c_int, c_int, S]),
    "egz_copy": (c_int, [P, P, c_long, S]),
    "egz_fill_zero": (c_int, [P, c_size_t, S]),
    # --- head + losses
    "egz_conv1x1_sigmoid_fwd": (c_int, [P, P, P, P, P, c_long, c_int, S]),
    "egz_conv1x1_sigmoid_bwd_ws_bytes": (c_size_t, [c_int]),
    "egz_conv1x1_sigmoid_bwd": (c_int, [P, P, P, P, P, P, P, c_long, c_int, P, c_size_t, S]),
    "egz_conv1x1_sigmoid_bwd_rows": (c_int, [c_long, c_int]),
    "egz_conv1x1_sigmoid_bwd_masked": (c_int, [P, P, P, P, P, P, P, P, P, c_long, c_int, P, c_size_t, S]),
    "egz_loss_ws_bytes": (c_size_t, [c_int]),
    "egz_floss_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, c_size_t, S]),
    "egz_floss_bwd": (c_int, [P, P, P, P, P, c_long, S]),
    "egz_mse_fwd": (c_int, [P, P, P, c_long, P, c_size_t, c_int, S]),
    "egz_mse_bwd": (c_int, [P, P, P, P, c_long, c_int, S]),
    "egz_mse_fwd_grad": (c_int, [P, P, P, P, c_long, c_int, P, c_int, P, S]),
    # --- AT: generic f32-MFMA GEMM + LSTM cell
    "egz_gemm": (c_int, [P, P, P, P, c_int, c_int, c_int, c_long, c_long, c_long, c_long, c_long, c_int, S]),
    "egz_gemm_batched": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_long, c_long, c_long, c_long, c_long, c_int, S]),
    "egz_lstm_cell_fwd": (c_int, [P, P, P, P, P, c_int, c_int, S]),
    "egz_lstm_cell_bwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, S]),
    "egz_lstm_wave_fwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, S]),
    "egz_lstm_persist_sync_words": (c_int, []),
    "egz_lstm_persist_fwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, S]),
    "egz_lstm_persist_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, S]),
    "egz_lstm_wave_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, S]),
    "egz_lstm_b1_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "egz_lstm_b1_fwd": (c_int, [P, c_int, P, P, P, P, P, P, P, P, c_int, c_int, c_int, S]),
    "egz_lstm_b1_bwd": (c_int, [P, P, c_int, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P, c_size_t, S]),
    "egz_tanh_fwd": (c_int, [P, P, c_long, S]),
    "egz_tanh_bwd": (c_int, [P, P, P, c_long, S]),
    "egz_add": (c_int, [P, P, P, c_long, S]),
    # --- optimizer
    "egz_cat2_planes": (c_int, [P, P, P, c_int, c_long, S]),
    "egz_u8_normalize": (c_int, [P, P, c_long, c_long, c_int, P, P, S]),
    "egz_crop_mean": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, S]),
    "egz_window_mean": (c_int, [P, P, P, c_int, c_int, c_int, c_int, S]),
    "egz_pixel_weighted_sum": (c_int, [P, P, P, c_int, c_int, c_int, S]),
    "egz_weighted_minmax": (c_int, [P, P, P, c_int, c_int, c_int, S]),
    "egz_aae_auc": (c_int, [P, P, c_int, c_int, c_int, P, c_int, c_double, P, S]),
    "egz_adam_step": (c_int, [P, P, P, P, c_long, c_double, c_double, c_double, c_double, c_int, c_double, P, S]),
    "egz_adam_step_dev": (c_int, [P, P, P, P, c_long, c_double, c_double, c_double, c_double, P, c_double, P, S]),
}


class EgazeHipError(RuntimeError):
    """Raised when a C-ABI entry point returns a non-zero code (mirrors the reference's convention of
    plain Python exceptions: utils.generalException / RuntimeError)."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is mandatory (no CPU fallback). "
            "Build it with `python __graft_entry__.py` or `csrc/build.sh`.")
    # torch first: it ships its own libamdhip64.so.7 + HSA runtime, and the process must hold exactly one HIP
    # runtime -- the one that owns the tensors' device memory.  Loading this library first would bind the system
    # runtime instead and every launch would fail with "no ROCm-capable device".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild the extension") from e
        fn.restype = res
        fn.argtypes = args
    return lib


LIB = _load()


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = LIB.egz_last_error().decode(errors="replace")
        raise EgazeHipError(f"{what or 'egaze-hip'} failed (code {rc}): {msg}")


def version() -> str:
    return LIB.egz_version().decode()
