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


def _newest_header():
    t = os.path.getmtime(__file__)
    for d in (CSRC, os.path.join(ROOT, "include")):
        for base, _, files in os.walk(d):
            for f in files:
                if f.endswith(".h"):
                    t = max(t, os.path.getmtime(os.path.join(base, f)))
    return t


def build(force: bool = False, verbose: bool = True) -> str:
    """One object per translation unit, compiled side by side (a header edit rebuilds all five, a .cpp edit only its own), then one link."""
    if not force and not _stale(LIB, [CSRC, os.path.join(ROOT, "include"), __file__]):
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = _newest_header()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + CSRC, "-Wno-unused-result", "-Wno-unused-value"]
    jobs = []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_t, os.path.getmtime(src)):
            jobs.append([hipcc] + flags + ["-x", "hip", "-c", src, "-o", obj])
    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=int(os.environ.get("NTTS_BUILD_JOBS", "5"))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in _sources()]
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs)
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
