"""CPU-side checks of the drop-in boundary: libfbdqn.so loads and exports exactly the symbols
include/fbdqn.h declares (no compute call is made here)."""
import ctypes
import os
import re

from dqnflappybird_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fbdqn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fb_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(L.SIGNATURES)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(L.LIB_PATH), "build libfbdqn.so first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(L.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.fb_version() >= 100


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(ROOT, "include", "fbdqn.h")).read()
    assert "torch" not in text.lower().replace("torch.tensor.data_ptr()", "")
    assert "at::" not in text and "#include <hip" not in text


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dqnflappybird_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in src.replace("no CPU fallback", ""), (dp, f)
