#!/usr/bin/env python
"""Did a source change alter the device code of kernels that were already measured / validated on the GPU?

    python tools/isa_diff.py <commit>        # compares gfx950 ISA of every kernel between <commit> and the working tree

Compiles csrc/{backbone,codec,kapi}.cpp of both trees to assembly (hipcc -S --cuda-device-only, no GPU needed), strips
comments / label numbering and reports kernels whose instruction stream differs, plus kernels that are new.  Used when
experimental (default-off) kernel variants are added between GPU sessions: the validated kernels must come out identical."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def asm(tree, name, out):
    src = os.path.join(tree, "neutts-air_amd", "csrc")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-I" + src, "-S", "--cuda-device-only",
                    os.path.join(src, name + ".cpp"), "-o", out], check=True, stderr=subprocess.DEVNULL)


def funcs(path):
    t = open(path).read()
    out = {}
    for m in re.finditer(r"^(_ZN4ntts\S*?):[^\n]*\n", t, re.M):
        a = m.end()
        body = re.sub(r";.*", "", t[a:t.index(".Lfunc_end", a)])
        out[m.group(1)] = re.sub(r"[ \t]+$", "", re.sub(r"\.LBB\d+_", ".LBB_", body), flags=re.M)   # (stripped comments leave their padding behind)
    return out


def main():
    commit = sys.argv[1]
    with tempfile.TemporaryDirectory() as tmp:
        old = os.path.join(tmp, "old")
        os.makedirs(old)
        ar = subprocess.run(["git", "-C", ROOT, "archive", commit, "neutts-air_amd/csrc", "include"], check=True, capture_output=True)
        subprocess.run(["tar", "-x", "-C", old], input=ar.stdout, check=True)
        rc = 0
        for name in ("backbone", "codec", "kapi"):
            a_s, b_s = os.path.join(tmp, name + "_old.s"), os.path.join(tmp, name + "_new.s")
            asm(old, name, a_s)
            asm(ROOT, name, b_s)
            a, b = funcs(a_s), funcs(b_s)
            # a template parameter appended with its default value renames every instantiation without changing it: with
            # ISA_DIFF_DROP_DEFAULT=<mangled suffix>, e.g. ELi4E for a trailing "int TN = 4" of gemm_kernel, new-tree names
            # ending in that argument are compared under their old names (function bodies refer to themselves by name too)
            drop = os.environ.get("ISA_DIFF_DROP_DEFAULT")
            if drop:
                tail = drop + "EEvNS_8GemmArgsE"
                ren = {n: n[:-len(tail)] + "EEEvNS_8GemmArgsE" for n in b if n.endswith(tail)}
                b = {ren.get(n, n): body for n, body in b.items()}
                strip = lambda body: re.sub(r"_ZN4ntts\w+", "SYM", body)     # directives repeat the (renamed) symbol
                a = {n: strip(body) for n, body in a.items()}
                b = {n: strip(body) for n, body in b.items()}
            changed = sorted(n for n in set(a) & set(b) if a[n] != b[n])
            print(f"{name}.cpp: {len(a)} kernels at {commit}, {len(b)} now; changed {len(changed)}, new {len(set(b) - set(a))}, "
                  f"gone {len(set(a) - set(b))}")
            for n in changed:
                print("   CHANGED", n)
                rc = 1
            for n in sorted(set(b) - set(a)):
                print("   new    ", n)
    sys.exit(rc)


if __name__ == "__main__":
    main()
