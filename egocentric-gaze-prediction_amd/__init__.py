"""egocentric-gaze-prediction_amd -- MI355X (gfx950) implementation of the SP / AT / LF hot path of
hyf015/egocentric-gaze-prediction behind the reference's own Python module API.

Import name: the directory name is not a Python identifier, so ``egaze_amd.py`` at the repository root
registers this package as ``egaze_amd`` (``from egaze_amd.models.model_SP import model_SP`` ...).

Sub-modules mirror the reference's top-level files: ``utils`` (make_layers/cfg/...), ``floss``,
``models.model_SP``, ``models.LSTMnet``, ``models.late_fusion``, ``SP``, ``AT``, ``LF``, ``gaze_full``.
Compute lives in ``csrc/libegaze_hip.so`` (C ABI, include/egaze_hip.h); there is no CPU fallback.
"""
from ._lib import version, EgazeHipError, LIB_PATH  # noqa: F401  (fails loudly when the .so is missing)
