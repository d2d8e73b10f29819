"""The C-ABI contract in three places must agree: include/egaze_hip.h (declarations), _lib.SIGNATURES (ctypes binding)
and the built csrc/libegaze_hip.so (exports).  No GPU needed: nothing is launched."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "egaze_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    decls = {}
    for m in re.finditer(r"\b([A-Za-z_][\w \*]*?)\b(egz_\w+)\s*\(([^;{]*)\)\s*;", text):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else args.count(",") + 1
        decls[m.group(2)] = n
    return decls


def test_header_binding_and_library_agree():
    import egaze_amd  # noqa: F401  (loads the library, resolves every bound symbol)
    from egaze_amd import _lib
    decl = _declared()
    assert len(decl) >= 59
    assert set(decl) == set(_lib.SIGNATURES), (sorted(set(decl) ^ set(_lib.SIGNATURES)))
    for name, (_, argtypes) in _lib.SIGNATURES.items():
        assert len(argtypes) == decl[name], (name, len(argtypes), decl[name])           # same arity
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("egz_")}
    assert exported == set(decl), sorted(exported ^ set(decl))                           # no strays either way
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert getattr(lib, name) is not None
    assert _lib.version().startswith("egaze-hip")
