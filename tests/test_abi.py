"""The C-ABI library loads and exports every function include/*.h declares; without a GPU it refuses to
create an engine (NTTS_ENODEV) -- there is no CPU fallback in the product library."""
import ctypes
import glob
import os
import re

import pytest

from neutts import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(ntts_\w+)\s*\(", src):
            names.add(m.group(1))
    return sorted(names)


def test_header_declares_functions():
    names = declared_functions()
    assert "ntts_backbone_create" in names and "ntts_backbone_decode" in names and len(names) >= 20


def test_library_exports_every_declared_symbol(hip_lib):
    lib = ctypes.CDLL(hip_lib)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_binding_covers_header(hip_lib):
    lib = _hip.load_library(hip_lib)   # sets argtypes for every bound symbol; raises on drift
    assert lib.ntts_abi_version() == _hip.ABI_VERSION


def test_no_cpu_fallback(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_hip.NeuTTSHipError) as ei:
        _hip.BackboneEngine(dict(vocab_size=64, hidden_size=64, intermediate_size=64, num_layers=1, num_heads=1,
                                 num_kv_heads=1, max_context=64, max_batch=1), 0, hip_lib)
    assert ei.value.code == -2  # NTTS_ENODEV


def test_emulator_exports_same_abi(emu_lib):
    lib = ctypes.CDLL(emu_lib)
    assert not [n for n in declared_functions() if not hasattr(lib, n)]


def test_product_never_imports_oracle():
    """The oracle is the checker, never the thing shipped: no file of the product package may import or execute
    anything under oracle/ (nor the SIMT emulator, nor /root/reference)."""
    pkg = os.path.join(ROOT, "neutts-air_amd")
    bad = []
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                src = open(os.path.join(base, f), errors="ignore").read()
                pat = r"^\s*(from|import)\s+oracle\b|/root/reference"
                if f != "build.py":                          # build.py also holds the recipe of the test-only emulator library
                    pat += r"|^\s*(#\s*include|from|import)\b.*simt_emu"
                if re.search(pat, src, flags=re.M):
                    bad.append(os.path.relpath(os.path.join(base, f), ROOT))
    assert not bad, f"product files referencing test infrastructure: {bad}"
