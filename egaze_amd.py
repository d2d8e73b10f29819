"""Import shim: ``import egaze_amd`` loads the package that lives in ``egocentric-gaze-prediction_amd/``.

The mandated directory name contains hyphens and cannot appear in an ``import`` statement, so this
module loads that directory as a regular package under the single canonical name ``egaze_amd``
(``from egaze_amd.models.model_SP import model_SP``, ``from egaze_amd.utils import make_layers, cfg`` ...).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "egocentric-gaze-prediction_amd")
_spec = importlib.util.spec_from_file_location(
    "egaze_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["egaze_amd"] = _pkg          # replaces this shim; sub-modules import as egaze_amd.<name>
_spec.loader.exec_module(_pkg)
