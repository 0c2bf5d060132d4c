"""What the codec's waveform error WOULD be for a given GEMM-operand format, simulated on the CPU with the oracle alone (no GPU):
every Linear / Conv1d / attention matmul of oracle/codec_ref.py runs with BOTH operands rounded to the format (fp32 accumulate),
everything else fp32 -- exactly what the codec engine's GEMMs do (csrc/codec.cpp).  Used to choose the default codec precision
(bf16 operands 7.0e-3 relative; fp16 operands have 3 more mantissa bits at the same matrix-core rate).

    python tools/codec_operand_sim.py [--fixture codec_neucodec] [--set 1]

Also reports the activation / weight magnitudes every GEMM sees (fp16's range is 6.1e-5 .. 65504 for normal numbers).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import codec_ref as cr  # noqa: E402


def rms(x):
    return float(np.sqrt(np.mean(np.square(np.asarray(x, dtype=np.float64)))))


class Rounded:
    """Context: F.linear / F.conv1d / torch.matmul with operands rounded to `dt` inside oracle.codec_ref."""

    def __init__(self, dt, stats=None, skip_head=False):
        self.dt, self.stats, self.skip_head = dt, stats, skip_head

    def r(self, x, what):
        if self.stats is not None:
            a = x.detach().abs()
            nz = a[a > 0]
            self.stats.append((what, float(a.max()), float(nz.min()) if nz.numel() else 0.0, float(a.pow(2).mean().sqrt())))
        return x if self.dt is None else x.to(self.dt).to(torch.float32)

    def __enter__(self):
        self.lin, self.conv, self.mm = F.linear, F.conv1d, torch.matmul
        me = self

        def linear(x, w, b=None):
            return me.lin(me.r(x, "x"), me.r(w, "w"), b)

        def conv1d(x, w, b=None, **kw):
            return me.conv(me.r(x, "x"), me.r(w, "w"), b, **kw)

        def matmul(a, b):
            return me.mm(me.r(a, "a"), me.r(b, "b"))

        cr.F.linear, cr.F.conv1d, cr.torch.matmul = linear, conv1d, matmul
        return self

    def __exit__(self, *a):
        cr.F.linear, cr.F.conv1d, cr.torch.matmul = self.lin, self.conv, self.mm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default="codec_neucodec")
    ap.add_argument("--set", type=int, default=1)
    ap.add_argument("--gain", type=float, default=1.0, help="multiply every STFT magnitude (head bias + ln gain)")
    a = ap.parse_args()
    from common import load_codec_fixture
    z, cfg, w = load_codec_fixture(a.fixture)
    if a.gain != 1.0:
        w = dict(w)
        b = w["decoder.head.linear.bias"].clone()
        b[: b.numel() // 2] += float(np.log(a.gain))
        w["decoder.head.linear.bias"] = b
    codes = torch.tensor(z[f"codes_{a.set}"][0, 0].tolist(), dtype=torch.long)[None, None, :]
    ref = cr.decode_code(cfg, w, codes)[0, 0].numpy()
    print(f"{a.fixture} set {a.set}: {codes.shape[-1]} frames, signal rms {rms(ref):.3e}")
    stats = []
    with Rounded(None, stats):
        cr.decode_code(cfg, w, codes)
    xs = [s for s in stats if s[0] in ("x", "a", "b")]
    ws = [s for s in stats if s[0] == "w"]
    print(f"activation operands: max |x| {max(s[1] for s in xs):.3e}, smallest rms {min(s[3] for s in xs):.3e}; "
          f"weights: max |w| {max(s[1] for s in ws):.3e}, smallest rms {min(s[3] for s in ws):.3e}")
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        with Rounded(dt):
            got = cr.decode_code(cfg, w, codes)[0, 0].numpy()
        e = rms(got - ref)
        print(f"operands rounded to {name}: waveform rms error {e:.3e}, relative {e / rms(ref):.3e}")


if __name__ == "__main__":
    main()
