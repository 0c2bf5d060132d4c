""""the GPU sits 0.06-0.11 relative RMS from the fp8 oracle while the emulator is exact; the explanation (fp32
summation order flips e4m3 roundings) is plausible but asserted, not demonstrated per layer".  Demonstrated here on the CPU with the
ORACLE ALONE: take the fp8 oracle and, right before every GEMM input is quantised to e4m3, move a small share of its bf16 values by ONE
bf16 ulp up or down -- the kind of difference two correct bf16 implementations show (MI355X's bf16 logits sit 0.3 bf16 ulps mean / 2 max
from the oracle's on random weights, tests/test_gpu_parity_matrix.py).  An e4m3 value has 3 mantissa bits: one bf16 ulp (2^-8 relative)
carries a value across an e4m3 rounding boundary with probability ~2^-8 / 2^-4 = 1 / 16, and then the value moves by a whole e4m3 step
(6-12 %); the GEMM behind it changes ALL its outputs a little, a few per cent of them by a bf16 ulp, and the next quantisation point
starts from there.  What comes out at the logits:

  * the noise level is a FIXED POINT of the quantised pipeline, not a function of how much was perturbed: 1 % of the values moved gives
    0.06-0.08 relative rms at 2 layers, 25 % gives 0.08-0.11 -- "exactly 0 or this much", which is what the two builds show (the
    emulator's matrix core is the oracle's fp32 chain: 0; MI355X's is not: 0.050-0.068);
  * it grows with the number of quantisation points like the measured distance does: 2 / 4 / 19 layers 0.07 / 0.08 / 0.11 here,
    0.050-0.068 / 0.08 / 0.097-0.108 on MI355X.

So ulp-level differences in front of the quantisation points reproduce the whole GPU-vs-oracle distance of the fp8 model, layer count
by layer count; no kernel error is needed to explain it.  (Parity of the fp8 variant stays unpinned all the same: there is no third-party
reference for it.)"""
import numpy as np
import torch

from oracle import backbone_ref as br


def _perturb(frac, seed):
    g = torch.Generator().manual_seed(seed)

    def hook(x):
        x32 = x.to(torch.float32)
        pick = torch.rand(x32.shape, generator=g) < frac
        sign = torch.where(torch.rand(x32.shape, generator=g) < 0.5, -1.0, 1.0)
        ulp = torch.pow(2.0, torch.floor(torch.log2(x32.abs().clamp_min(1e-30))) - 7)     # one bf16 ulp of each value
        return torch.where(pick, x32 + sign * ulp, x32).to(x.dtype)
    return hook


def _rel(a, b):
    fin = torch.isfinite(a) & torch.isfinite(b)
    return float(torch.sqrt(torch.mean((a[fin] - b[fin]) ** 2)) / torch.sqrt(torch.mean(b[fin] ** 2)))


def _distance(cfg, wq, frac, n_prompts=4):
    eos, d = cfg.vocab_size - 1, []
    for u in range(n_prompts):
        p = br.synthetic_prompt(cfg, 10 + u, 40)
        clean = br.generate(cfg, wq, p, len(p) + 1, eos, min_new_tokens=1, keep_logits=True).logits[0]
        br.FP8_ACT_HOOK = _perturb(frac, 100 + u)
        try:
            noisy = br.generate(cfg, wq, p, len(p) + 1, eos, min_new_tokens=1, keep_logits=True).logits[0]
        finally:
            br.FP8_ACT_HOOK = None
        d.append(_rel(noisy, clean))
    return float(np.mean(d))


def test_fp8_logits_distance_is_reproduced_by_one_ulp_differences_before_the_quantisation_points():
    by_depth, by_frac = {}, {}
    for layers in (2, 4, 19):
        cfg = br.BackboneConfig(vocab_size=512, hidden_size=384, intermediate_size=1024, num_layers=layers, num_heads=6, num_kv_heads=2)
        w = br.make_weights(cfg, 23, peak_sigma=0.5)
        wq = br.fp8_quantize_weights(br.cast_weights(w, torch.bfloat16), br.default_fp8_input_scales(cfg))
        by_depth[layers] = _distance(cfg, wq, 0.01)
        if layers == 2:
            by_frac = {f: _distance(cfg, wq, f) for f in (0.0, 0.01, 0.06, 0.25)}
    print("fp8 oracle, 1 % of every GEMM input moved by one bf16 ulp before its quantisation, relative rms of the logits: "
          + ", ".join(f"{k} layers {v:.3f}" for k, v in by_depth.items()) + "  (MI355X vs the oracle: 0.050-0.068 / 0.08 / 0.097-0.108)")
    print("   by the share of values moved, 2 layers: " + ", ".join(f"{f:g} -> {v:.3f}" for f, v in by_frac.items()))
    assert by_frac[0.0] == 0.0                                              # the hook itself changes nothing
    assert 0.03 <= by_depth[2] <= 0.12 and 0.04 <= by_depth[4] <= 0.16 and 0.06 <= by_depth[19] <= 0.25, by_depth
    assert by_depth[2] < by_depth[19]                                       # more quantisation points, more of it
    assert by_frac[0.25] <= 2.0 * by_frac[0.01]                             # a fixed point, not proportional to what was perturbed
