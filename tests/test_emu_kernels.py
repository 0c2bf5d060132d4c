"""Kernel sources executed on the CPU SIMT emulator (tests/simt_emu) vs a plain torch fp32 reference of
the same op.  Checks indexing / fragment layouts / swizzles / edge handling before any GPU time is spent;
the same cases run on the real device in test_gpu_kernels.py."""
import ctypes as C

import numpy as np
import pytest
import torch

from neutts import _hip


def _bf16(t):
    return t.to(torch.bfloat16)


def run_gemm(lib, x, w, bias, variant):
    M, K = x.shape
    N = w.shape[0]
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=x.device)
    rc = lib.ntts_k_gemm_bf16(C.c_void_p(x.data_ptr()), K, C.c_void_p(w.data_ptr()),
                              C.c_void_p(bias.data_ptr()) if bias is not None else None,
                              C.c_void_p(out.data_ptr()), N, M, N, K, variant)
    assert rc == 0
    return out


GEMM_CASES = [  # (M, N, K, variant, bias)
    (70, 200, 128, 1, True),     # L tile, ragged M and N (N not a multiple of 16)
    (130, 128, 192, 1, False),   # L tile, 2 m-blocks
    (5, 64, 64, 2, True),        # S tile, tiny M
    (100, 176, 256, 2, True),    # S tile, ragged
    (37, 64, 448, 3, False),     # S tile split-K + slab reduce
    (300, 272, 128, 4, True),    # XL tile (256 x 256, 16 waves), ragged M and N
    (300, 272, 192, 5, True),    # XL tile on the 4-slot ring of 32-wide K slices (64-byte LDS rows, other swizzle)
    (70, 200, 64, 5, False),     # ... a single 64-wide K tile = 2 slices, fewer than the ring holds
    (130, 144, 320, 6, True),    # L tile, 3-slot ring of 32-wide slices
]


@pytest.mark.parametrize("M,N,K,variant,has_bias", GEMM_CASES)
def test_gemm_emu(emu_lib, M, N, K, variant, has_bias):
    lib = _hip.load_library(emu_lib)
    g = torch.Generator().manual_seed(M * 1000 + N)
    x = _bf16(torch.randn(M, K, generator=g))
    w = _bf16(torch.randn(N, K, generator=g) / K ** 0.5)
    b = _bf16(torch.randn(N, generator=g)) if has_bias else None
    out = run_gemm(lib, x, w, b, variant).float()
    ref = x.float() @ w.float().t()          # asymmetric operands: a transposed C would not pass
    if b is not None:
        ref = ref + b.float()
    ref = _bf16(ref).float()
    err = (out - ref).abs()
    tol = 2.0 ** -7 * ref.abs().clamp(min=1e-2)   # 2 bf16 ulps: fp32 summation order differs
    assert bool((err <= tol).all()), f"max err {err.max()} at {err.argmax()}"
    assert (out != ref).float().mean() < 0.02     # and almost every element is bit-identical


def test_rmsnorm_emu(emu_lib):
    lib = _hip.load_library(emu_lib)
    g = torch.Generator().manual_seed(5)
    for rows, cols in [(3, 448), (9, 896), (2, 1024)]:
        x = _bf16(torch.randn(rows, cols, generator=g) * 3)
        w = _bf16(1 + 0.1 * torch.randn(cols, generator=g))
        y = torch.zeros_like(x)
        assert lib.ntts_k_rmsnorm_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(y.data_ptr()),
                                       rows, cols, 1e-6) == 0
        from oracle.backbone_ref import rms_norm
        ref = rms_norm(x, w, 1e-6)
        assert (y != ref).float().mean() < 0.002 and (y.float() - ref.float()).abs().max() <= 2 ** -6 * ref.float().abs().max()


def mfma_probe_expected():
    l = np.arange(64)[:, None]
    r = np.arange(4)[None, :]
    return np.stack([(l >> 4) * 4 + r + 0 * l, (l & 15) + 0 * r, (((l & 15) * 2 + 1) % 32 + 1) + 0 * r]).astype(np.float32)


def test_mfma_probe_emu(emu_lib):
    lib = _hip.load_library(emu_lib)
    out = torch.zeros(3 * 64 * 4)
    assert lib.ntts_k_mfma_probe(C.c_void_p(out.data_ptr())) == 0
    assert np.array_equal(out.numpy().reshape(3, 64, 4), mfma_probe_expected())
