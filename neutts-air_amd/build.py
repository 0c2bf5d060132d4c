"""Build libneutts_hip.so (gfx950) in-tree with hipcc.  No CPU fallback is built or shipped.

    python neutts-air_amd/build.py            # product library
    python neutts-air_amd/build.py --emu      # tests/simt_emu/libneutts_emu.so (TEST infrastructure:
                                              # the same sources on the CPU SIMT emulator)
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["backbone.cpp", "kapi.cpp", "codec.cpp", "encoder.cpp", "stream.cpp"]
LIB = os.path.join(HERE, "libneutts_hip.so")
EMU_DIR = os.path.join(ROOT, "tests", "simt_emu")
EMU_LIB = os.path.join(EMU_DIR, "libneutts_emu.so")


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    for d in deps:
        for base, _, files in os.walk(d) if os.path.isdir(d) else [(os.path.dirname(d), [], [os.path.basename(d)])]:
            for f in files:
                if f.endswith((".h", ".cpp", ".hip", ".py")) and os.path.getmtime(os.path.join(base, f)) > t:
                    return True
    return False


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale(LIB, [CSRC, os.path.join(ROOT, "include"), __file__]):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-I" + CSRC, "-Wno-unused-result", "-Wno-unused-value", "-o", LIB + ".tmp"] + _sources()
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)      # a failed build never destroys the previous library
    return LIB


def build_emu(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale(EMU_LIB, [CSRC, EMU_DIR, os.path.join(ROOT, "include")]):
        return EMU_LIB
    cxx = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
    cmd = [cxx, "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-I" + EMU_DIR, "-Wno-unused-result",
           "-Wno-unknown-pragmas", "-Wno-pass-failed", "-o", EMU_LIB + ".tmp", os.path.join(EMU_DIR, "emu.cpp")] + _sources()
    if verbose:
        print("[build-emu]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(EMU_LIB + ".tmp", EMU_LIB)
    return EMU_LIB


if __name__ == "__main__":
    if "--emu" in sys.argv:
        print(build_emu(force="--force" in sys.argv))
    else:
        print(build(force="--force" in sys.argv))
