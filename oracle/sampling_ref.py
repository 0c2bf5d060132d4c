"""CPU restatement of the device sampler's DRAW (TEST INFRASTRUCTURE, see oracle/__init__.py).

What the reference runs (ref:neutts/neutts.py:338-347: do_sample=True, temperature=1.0, top_k=50) ends in
GenerationMixin._sample (hf:generation/utils.py:2894-2941): TemperatureLogitsWarper (scores / T), TopKLogitsWarper
(hf:generation/logits_process.py:542-595: scores < k-th largest -> -inf, ties at the k-th value kept), softmax,
torch.multinomial(1).  torch's generator stream is not reproducible by a device kernel, so the engine specifies its own
draw -- counter-based, a function of (request seed, step) only -- and THIS file is that specification in numpy; the
membership / distribution tests (tests/test_emu_sampling.py, tests/test_gpu_backbone.py) tie it to the HF semantics, the
exact-draw tests tie the kernel (neutts-air_amd/csrc/kernels/sample.h sample_topk_row) to this file token for token:

  survivors = every token whose processed logit (bf16 value, EOS mask applied) is >= the k-th largest, in token-id order
  e_a       = fp32 exp((logit_a - max) * (1 / T))
  total     = e_0 + e_1 + ... in that order, fp32
  u         = (Philox4x32-10(counter = (step, 0, 0, 0), key = (seed & 0xffffffff, seed >> 32))[0] >> 8) / 2^24
  token     = first survivor a with (e_0 + ... + e_a) > u * total      (the last survivor if none)

Parity unpinned against a third party by construction (there is none for the draw); pinned to Philox's published
known-answer vectors below (Salmon et al., SC'11; Random123 kat_vectors: philox4x32-10).
"""
from __future__ import annotations

import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(counter, key):
    """One Philox4x32-10 block: counter = 4 x uint32, key = 2 x uint32 -> 4 x uint32 (python ints)."""
    c = [int(x) & 0xFFFFFFFF for x in counter]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c


def uniform(seed: int, step: int) -> float:
    """The draw's uniform in [0, 1): 24 bits of the block's first word."""
    return (philox4x32_10([step, 0, 0, 0], [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF])[0] >> 8) / 16777216.0


def sample_topk(logits, k: int, temperature: float, seed: int, step: int):
    """logits: the processed row (float32 holding bf16 values, -inf at a masked EOS).  Returns (token, margin): margin is the
    relative distance of u * total from the nearer of the two cumulative sums around it -- a draw with a margin of a few fp32
    ulps may legitimately fall on the neighbouring token on a device whose exp differs from numpy's in the last bit."""
    x = np.asarray(logits, dtype=np.float32)
    k = min(int(k), x.size, 512)
    kth = np.partition(x, x.size - k)[x.size - k]
    idx = np.flatnonzero(x >= kth)[:512]                  # token-id order; ties at the k-th value kept (capped like the kernel)
    v = x[idx]
    it = np.float32(1.0) / np.float32(temperature)
    e = np.exp(((v - v.max()) * it).astype(np.float32)).astype(np.float32)
    total = np.float32(0.0)
    for a in e:
        total = np.float32(total + a)
    target = np.float32(np.float32(uniform(seed, step)) * total)
    acc, pick, margin = np.float32(0.0), int(idx[-1]), 1.0
    cums = []
    for a in e:
        acc = np.float32(acc + a)
        cums.append(float(acc))
    for j, cacc in enumerate(cums):
        if cacc > float(target):
            pick = int(idx[j])
            break
    margin = min(abs(c - float(target)) for c in cums) / max(float(total), 1e-30)
    return pick, margin
