"""Kernel-level parity on a real MI355X, through the C-ABI of libneutts_hip.so, against a plain torch fp32
reference of the same op (computed on the host CPU: nothing here trusts a torch-ROCm GPU op)."""
import ctypes as C

import numpy as np
import pytest
import torch

from neutts import _hip
from test_emu_kernels import GEMM_CASES, mfma_probe_expected, run_gemm, _bf16

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _hip.load_library(hip_lib)


def test_mfma_layout_probe(lib):
    out = torch.zeros(3 * 64 * 4, device="cuda")
    assert lib.ntts_k_mfma_probe(C.c_void_p(out.data_ptr())) == 0
    got = out.cpu().numpy().reshape(3, 64, 4)
    assert np.array_equal(got, mfma_probe_expected()), got


BIG = [  # the decode-batch and prefill shapes of NeuTTS-Air
    (256, 1152, 896, 2, True),    # QKV, decode batch, S tile
    (256, 896, 896, 3, False),    # o_proj, split-K slabs + reduce
    (256, 896, 4864, 3, False),   # down_proj, split-K
    (500, 1152, 896, 1, True),    # prefill QKV, L tile, ragged M
    (1000, 2048, 1024, 1, True),  # codec-like
    (1500, 1152, 896, 4, True),   # XL tile (256 x 256, 1024 threads): prefill QKV, ragged M
    (3000, 896, 4864, 4, False),  # XL tile: prefill down_proj, N not a multiple of 256
    (1500, 1152, 896, 5, True),   # XL tile on the 4-slot ring of 32-wide K slices: prefill QKV, ragged M
    (3000, 896, 4864, 5, False),  # ... prefill down_proj (152 slices)
    (2000, 9728, 896, 5, False),  # ... prefill gate/up width
    (1500, 1152, 896, 7, True),   # natural-order 256 x 288 tile: prefill QKV (4 column blocks, none padded), ragged M
    (700, 1000, 448, 7, False),   # ... N = 3 tiles + 136 columns
]


@pytest.mark.parametrize("M,N,K,variant,has_bias", GEMM_CASES + BIG)
def test_gemm(lib, M, N, K, variant, has_bias):
    g = torch.Generator().manual_seed(M * 1000 + N)
    x = _bf16(torch.randn(M, K, generator=g))
    w = _bf16(torch.randn(N, K, generator=g) / K ** 0.5)
    b = _bf16(torch.randn(N, generator=g)) if has_bias else None
    out = run_gemm(lib, x.cuda(), w.cuda(), b.cuda() if b is not None else None, variant).float().cpu()
    ref = x.float() @ w.float().t()
    if b is not None:
        ref = ref + b.float()
    ref = _bf16(ref).float()
    err = (out - ref).abs()
    tol = 2.0 ** -7 * ref.abs().clamp(min=1e-2)
    assert bool((err <= tol).all()), f"max err {err.max()} at {np.unravel_index(int(err.argmax()), err.shape)}"
    assert (out != ref).float().mean() < 0.02


def test_rmsnorm(lib):
    from oracle.backbone_ref import rms_norm
    g = torch.Generator().manual_seed(5)
    for rows, cols in [(3, 448), (256, 896), (1000, 1024)]:
        x = _bf16(torch.randn(rows, cols, generator=g) * 3)
        w = _bf16(1 + 0.1 * torch.randn(cols, generator=g))
        xd, wd = x.cuda(), w.cuda()
        y = torch.zeros_like(xd)
        assert lib.ntts_k_rmsnorm_bf16(C.c_void_p(xd.data_ptr()), C.c_void_p(wd.data_ptr()), C.c_void_p(y.data_ptr()),
                                       rows, cols, 1e-6) == 0
        ref = rms_norm(x, w, 1e-6)
        y = y.cpu()
        assert (y != ref).float().mean() < 0.002
        assert (y.float() - ref.float()).abs().max() <= 2 ** -6 * ref.float().abs().max()


def test_hbm_copy_bandwidth(lib):
    gbps = C.c_double()
    assert lib.ntts_k_membw(1 << 30, 10, C.byref(gbps)) == 0
    assert gbps.value > 2000, f"HBM copy only {gbps.value:.0f} GB/s"


def test_fp8_quantize(lib, hip_lib):
    from test_emu_kernels import fp8_quantize_case
    fp8_quantize_case(hip_lib)


@pytest.mark.parametrize("M,N,K,variant,has_bias", [(70, 200, 256, 1, True), (5, 64, 128, 2, False), (300, 272, 384, 4, True),
                                                    (256, 1280, 768, 2, True), (2000, 4096, 768, 4, False), (256, 768, 2048, 1, False)])
def test_fp8_gemm(lib, hip_lib, M, N, K, variant, has_bias):
    from test_emu_kernels import fp8_gemm_case
    fp8_gemm_case(hip_lib, M, N, K, variant, has_bias)


def test_silu_all_bf16_inputs(lib):
    """The SiLU of the gate/up epilogues on EVERY bf16 input: both forms (division by 1 + expf(-x); the fast form the kernels
    use) round to exactly torch's bf16 SiLU for every finite input, so swapping one for the other cannot move a token."""
    from test_emu_kernels import silu_all_bf16
    for variant in (0, 1, 2):      # 2 = the packed two-at-a-time form of the GEMM epilogues (silu_fast2)
        x, got, ref = silu_all_bf16(lib, "cuda", variant)
        fin = torch.isfinite(x.float())
        bad = (got.view(torch.int16)[fin] != ref.view(torch.int16)[fin])
        # -0.0 vs +0.0 for tiny negative inputs would show up here too: the bit patterns must agree
        assert int(bad.sum()) == 0, (variant, int(bad.sum()), x[fin][bad][:8], got[fin][bad][:8], ref[fin][bad][:8])
