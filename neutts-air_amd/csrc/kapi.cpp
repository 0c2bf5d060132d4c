// kapi.cpp -- kernel-level C entry points (parity tests, micro-benchmarks).  Device pointers in,
// NULL stream, blocking.  See include/neutts_hip.h.
#include <ntts/dev.h>

#include "../../include/neutts_hip.h"
#include "kernels/gemm.h"
#include "kernels/norm.h"

using namespace ntts;

extern "C" int ntts_k_gemm_bf16(const void* A, int64_t lda, const void* W, const void* bias, void* C, int64_t ldc,
                                int32_t M, int32_t N, int32_t K, int32_t variant) {
    if (!A || !W || !C || M < 1 || N < 1 || K < 64 || (K % 64) || (lda % 8) || (ldc % 8)) return NTTS_EINVAL;
    GemmArgs a{};
    a.X = (const bf16_t*)A; a.ldx = lda; a.W = (const bf16_t*)W; a.ldw = K; a.bias = (const bf16_t*)bias;
    a.out = C; a.ldo = ldc; a.M = M; a.N = N; a.K = K;
    if (variant == 0) variant = M > 64 ? 1 : 2;
    if (variant == 1) NTTS_GEMM_L(EPI_BF16, a, 1, (hipStream_t)0);
    else if (variant == 2) NTTS_GEMM_S(EPI_BF16, a, 1, (hipStream_t)0);
    else if (variant == 4) NTTS_GEMM_XL(EPI_BF16, a, 1, (hipStream_t)0);
    else if (variant == 5) gemm_launch<4, 4, 4, EPI_BF16, 4, 0, 32>(a, 1, (hipStream_t)0);   // XL tile, 4 ring slots of K = 32
    else if (variant == 6) gemm_launch<2, 2, 4, EPI_BF16, 3, 0, 32>(a, 1, (hipStream_t)0);   // L tile, 3 ring slots of K = 32
    else if (variant == 7) gemm_launch<4, 3, 4, EPI_BF16, 2, 0, 64, false, false, 6>(a, 1, (hipStream_t)0);   // natural-order 256 x 288 tile, 12 waves (prefill QKV)
    else if (variant == 3) {  // split-K slabs reduced by the norm kernel (the decode o_proj / down_proj path)
        if (bias || (N % 16) || ldc != N) return NTTS_EINVAL;
        const int ks = 4, ns = gemm_nsplit(K, ks);
        float* slabs = nullptr;
        if (hipMalloc((void**)&slabs, (size_t)ns * M * N * sizeof(float)) != hipSuccess) return NTTS_ENOMEM;
        a.out = slabs; a.ldo = N;
        NTTS_GEMM_S(EPI_SPLITK, a, ks, (hipStream_t)0);
        NormArgs n{};
        n.slabs = slabs; n.nslab = ns; n.slab_rows = M; n.resid_out = (bf16_t*)C; n.M = M; n.H = N;
        add_rmsnorm_launch(n, (hipStream_t)0);
        if (hipDeviceSynchronize() != hipSuccess) { hipFree(slabs); return NTTS_EHIP; }
        hipFree(slabs);
    } else return NTTS_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return NTTS_EHIP;
    return hipGetLastError() == hipSuccess ? NTTS_OK : NTTS_EHIP;
}

// ---- fp8 probes: the conversion the producers use, and one fp8 GEMM (row-major e4m3 operands), for tests against torch
NTTS_KERNEL(256) void fp8_quantize_kernel(const float* in, unsigned char* out, long n, float inv_scale) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < n) { const unsigned short q = f2fp8x2(in[i] * inv_scale, in[i + 1] * inv_scale); out[i] = (unsigned char)(q & 0xff); out[i + 1] = (unsigned char)(q >> 8); }
    else if (i < n) out[i] = f2fp8c(in[i] * inv_scale);
}
extern "C" int ntts_k_fp8_quantize(const float* in_dev, void* out_dev, int64_t n, float inv_scale) {
    if (!in_dev || !out_dev || n < 1) return NTTS_EINVAL;
    NTTS_LAUNCH((fp8_quantize_kernel), dim3((unsigned)((n + 511) / 512)), dim3(256), (hipStream_t)0, in_dev, (unsigned char*)out_dev, (long)n, inv_scale);
    if (hipDeviceSynchronize() != hipSuccess) return NTTS_EHIP;
    return hipGetLastError() == hipSuccess ? NTTS_OK : NTTS_EHIP;
}
extern "C" int ntts_k_gemm_fp8(const void* A, const void* W, const float* wscale, float xscale, const void* bias, void* C,
                               int32_t M, int32_t N, int32_t K, int32_t variant) {
    if (!A || !W || !wscale || !C || M < 1 || N < 1 || K < 128 || (K % 128) || (N % 8)) return NTTS_EINVAL;
    GemmArgs a{};
    a.X = (const bf16_t*)A; a.ldx = K; a.W = (const bf16_t*)W; a.ldw = K; a.bias = (const bf16_t*)bias;
    a.out = C; a.ldo = N; a.M = M; a.N = N; a.K = K; a.wscale = wscale; a.xscale = xscale;
    if (variant == 2) gemm_launch<4, 1, 1, EPI_BF16, 4, 0, 64, false, true>(a, 1, (hipStream_t)0);          // 64 x 64
    else if (variant == 4) gemm_launch<4, 4, 4, EPI_BF16, 2, 0, 64, false, true>(a, 1, (hipStream_t)0);     // 256 x 256
    else gemm_launch<2, 2, 4, EPI_BF16, 2, 0, 64, false, true>(a, 1, (hipStream_t)0);                       // 128 x 128
    if (hipDeviceSynchronize() != hipSuccess) return NTTS_EHIP;
    return hipGetLastError() == hipSuccess ? NTTS_OK : NTTS_EHIP;
}

extern "C" int ntts_k_rmsnorm_bf16(const void* x, const void* w, void* y, int32_t rows, int32_t cols, float eps) {
    if (!x || !w || !y || rows < 1 || cols < 8 || (cols % 8) || cols > 2048) return NTTS_EINVAL;
    NormArgs n{};
    n.o_bf16 = (const bf16_t*)x; n.norm_w = (const bf16_t*)w; n.normed_out = (bf16_t*)y; n.M = rows; n.H = cols; n.eps = eps;
    add_rmsnorm_launch(n, (hipStream_t)0);
    if (hipDeviceSynchronize() != hipSuccess) return NTTS_EHIP;
    return hipGetLastError() == hipSuccess ? NTTS_OK : NTTS_EHIP;
}

NTTS_KERNEL(256) void membw_copy_kernel(const u32x4* src, u32x4* dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}

extern "C" int ntts_k_membw(size_t bytes, int32_t iters, double* gbps) {
    if (!gbps || bytes < 4096 || iters < 1) return NTTS_EINVAL;
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { if (a) hipFree(a); return NTTS_ENOMEM; }
    hipMemset(a, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long n = bytes / 16;
    NTTS_LAUNCH((membw_copy_kernel), dim3(2048), dim3(256), (hipStream_t)0, (const u32x4*)a, (u32x4*)b, n);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) NTTS_LAUNCH((membw_copy_kernel), dim3(2048), dim3(256), (hipStream_t)0, (const u32x4*)a, (u32x4*)b, n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *gbps = ms > 0 ? 2.0 * (double)(n * 16) * iters / (ms * 1e-3) / 1e9 : 0;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(a); hipFree(b);
    return hipDeviceSynchronize() == hipSuccess ? NTTS_OK : NTTS_EHIP;
}

// Micro-benchmark of one GEMM tile configuration (tools/ubench_gemm.py): `copies` rotating weight buffers keep W
// HBM-cold when copies * N * K * 2 B exceeds the 256 MB Infinity Cache.  config: tile family, abl: ablation bits
// of gemm_kernel.  Returns the average microseconds per launch (back-to-back launches on the NULL stream).
NTTS_KERNEL(64) void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 0; }
// touch `n16` 16-byte words so that they are resident in the memory-side cache (and this XCD's L2) afterwards
NTTS_KERNEL(256) void prefetch_kernel(const u32x4* src, long n16, int* sink) {
    unsigned int acc = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) acc ^= src[i][0];
    if (acc == 0x12345678u && sink) *sink = 1;
}

NTTS_KERNEL(256) void random_fill_kernel(bf16_t* dst, long n, unsigned int seed) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        unsigned int h = (unsigned int)i * 0x9e3779b1u + seed;
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
        // sign + exponent 120 .. 127 (|v| in [2^-7, 2)) + 7 random mantissa bits: a wide spread of magnitudes, no inf / nan
        dst[i] = (bf16_t)(((h & 1u) << 15) | ((120u + ((h >> 1) & 7u)) << 7) | ((h >> 4) & 0x7fu));
    }
}

template <int WM, int WN, int TM, int NS, int BK = 64>
static void probe_launch(const GemmArgs& a, int ks, int abl) {
    constexpr int EPI = EPI_BF16;
    switch (abl) {
        case 1: gemm_launch<WM, WN, TM, EPI, NS, 1, BK>(a, ks, 0); break;
        case 2: gemm_launch<WM, WN, TM, EPI, NS, 2, BK>(a, ks, 0); break;
        case 3: gemm_launch<WM, WN, TM, EPI, NS, 3, BK>(a, ks, 0); break;
        case 4: gemm_launch<WM, WN, TM, EPI, NS, 4, BK>(a, ks, 0); break;
        case 7: gemm_launch<WM, WN, TM, EPI, NS, 7, BK>(a, ks, 0); break;
        default: gemm_launch<WM, WN, TM, EPI, NS, 0, BK>(a, ks, 0); break;
    }
}

static void probe_dispatch(int config, const GemmArgs& a, int ks, int abl) {
    switch (config) {
        case 0: NTTS_LAUNCH((empty_kernel), dim3(256), dim3(64), (hipStream_t)0, (int*)nullptr); break;
        case 10: probe_launch<4, 1, 1, 2>(a, ks, abl); break;   // 64 x 64, 4 waves
        case 11: probe_launch<4, 1, 1, 3>(a, ks, abl); break;
        case 12: probe_launch<4, 1, 1, 4>(a, ks, abl); break;
        case 13: probe_launch<4, 1, 1, 6>(a, ks, abl); break;
        case 20: probe_launch<2, 1, 1, 4>(a, ks, abl); break;   // 32 x 64, 2 waves
        case 21: probe_launch<2, 1, 2, 4>(a, ks, abl); break;   // 64 x 64, 2 waves
        case 22: probe_launch<1, 1, 4, 4>(a, ks, abl); break;   // 64 x 64, 1 wave
        case 23: probe_launch<2, 2, 2, 3>(a, ks, abl); break;   // 64 x 128, 4 waves
        case 24: probe_launch<4, 2, 1, 3>(a, ks, abl); break;   // 64 x 128, 8 waves
        case 25: probe_launch<8, 1, 1, 3>(a, ks, abl); break;   // 128 x 64, 8 waves
        case 26: probe_launch<4, 1, 2, 3>(a, ks, abl); break;   // 128 x 64, 4 waves
        case 30: probe_launch<2, 2, 4, 2>(a, ks, abl); break;   // 128 x 128, 4 waves (prefill tile)
        case 31: probe_launch<2, 2, 4, 3>(a, ks, abl); break;
        case 40: probe_launch<4, 2, 4, 2>(a, ks, abl); break;   // 256 x 128, 8 waves
        case 41: probe_launch<2, 4, 4, 2>(a, ks, abl); break;   // 128 x 256, 8 waves
        case 42: probe_launch<4, 4, 4, 2>(a, ks, abl); break;   // 256 x 256, 16 waves
        case 43: probe_launch<4, 2, 4, 3>(a, ks, abl); break;   // 256 x 128, 8 waves, 3 stages (144 KB)
        case 44: probe_launch<2, 2, 8, 2>(a, ks, abl); break;   // 256 x 128, 4 waves (128 x 64 per wave)
        case 45: probe_launch<2, 4, 8, 2>(a, ks, abl); break;   // 256 x 256, 8 waves (128 x 64 per wave)
        case 46: probe_launch<4, 4, 4, 4, 32>(a, ks, abl); break;   // 256 x 256, 16 waves, 4 slots of K = 32 (128 KB)
        case 47: probe_launch<4, 4, 4, 3, 32>(a, ks, abl); break;   // ... 3 slots (96 KB)
        case 48: probe_launch<2, 4, 8, 4, 32>(a, ks, abl); break;   // 256 x 256, 8 waves, 4 slots of K = 32
        case 49: probe_launch<2, 2, 4, 4, 32>(a, ks, abl); break;   // 128 x 128, 4 waves, 4 slots of K = 32 (64 KB: 2 blocks / CU)
        case 50: probe_launch<4, 1, 4, 3>(a, ks, abl); break;   // 256 x 64, 4 waves: all decode rows in one block
        case 51: probe_launch<8, 1, 2, 3>(a, ks, abl); break;   // 256 x 64, 8 waves
        case 52: probe_launch<8, 1, 2, 2>(a, ks, abl); break;
        case 53: probe_launch<8, 2, 2, 2>(a, ks, abl); break;   // 256 x 128, 16 waves
        case 54: probe_launch<4, 2, 2, 3>(a, ks, abl); break;   // 128 x 128, 8 waves
        case 55: probe_launch<4, 4, 5, 2>(a, ks, abl); break;   // 320 x 256, 16 waves (80 x 64 per wave): 11 % fewer LDS-DMA bytes per FLOP than 256 x 256; 144 KB
        case 57: probe_launch<2, 4, 10, 2>(a, ks, abl); break;  // 320 x 256, 8 waves (160 x 64 per wave, 2 waves per SIMD: up to 256 registers each)
        case 58: probe_launch<2, 4, 12, 2>(a, ks, abl); break;  // 384 x 256, 8 waves (192 x 64 per wave)
        case 56: probe_launch<4, 4, 6, 2>(a, ks, abl); break;   // 384 x 256, 16 waves (96 x 64 per wave): 17 % fewer; 160 KB = all of a CU's LDS
        default: break;
    }
}

extern "C" int ntts_k_gemm_probe(int32_t M, int32_t N, int32_t K, int32_t config, int32_t abl, int32_t copies, int32_t iters,
                                 double* us) {
    if (!us || M < 1 || N < 16 || K < 64 || (K % 64) || copies < 1 || iters < 1) return NTTS_EINVAL;
    bf16_t *X = nullptr, *W = nullptr, *C = nullptr;
    const size_t wn = (size_t)N * K;
    if (hipMalloc((void**)&X, (size_t)M * K * 2) != hipSuccess || hipMalloc((void**)&W, wn * 2 * copies) != hipSuccess ||
        hipMalloc((void**)&C, (size_t)M * N * 2) != hipSuccess)
        return NTTS_ENOMEM;
    hipMemset(X, 0x11, (size_t)M * K * 2);
    hipMemset(W, 0x22, wn * 2 * copies);
    if (abl & 32) {   // operands with the statistics of real activations / weights (hashed bf16 values in (-2, 2)) instead of one constant:
                      // the matrix cores' power draw -- and with it the clock the chip sustains -- depends on how many bits toggle
        NTTS_LAUNCH((random_fill_kernel), dim3(2048), dim3(256), (hipStream_t)0, X, (long)M * K, 0x9e3779b9u);
        NTTS_LAUNCH((random_fill_kernel), dim3(2048), dim3(256), (hipStream_t)0, W, (long)(wn * copies), 0x85ebca6bu);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int ks = config >= 1000 ? config / 1000 : 1;   // config = 1000 * ksplit + tile: the K split in gridDim.y (timing only: every split stores the same bf16 tile)
    config %= 1000;
    const bool pf = (abl & 8) != 0;
    const int tile_major = (abl & 16) ? 1 : 0;     // weights addressed tile-major (same bytes, sequential per workgroup)
    const bool x_ktm = (abl & 64) != 0;            // X addressed k-tile-major ([K / 64][M][64]: a block's X tile is one contiguous run; timing only)
    abl &= 7;   // (bits 8, 16, 32, 64 are handled here)
    auto run = [&](int i) {
        if (pf) NTTS_LAUNCH((prefetch_kernel), dim3(256), dim3(256), (hipStream_t)0, (const u32x4*)(W + (size_t)((i + 1) % copies) * wn), (long)(wn / 8), (int*)nullptr);
        GemmArgs a{};
        a.X = X; a.ldx = K; a.W = W + (size_t)(i % copies) * wn; a.ldw = K; a.out = C; a.ldo = N; a.M = M; a.N = N; a.K = K;
        a.w_tile_major = tile_major;
        if (x_ktm) { a.ldx = 64; a.x_kt_stride = (long)M * 128; }
        probe_dispatch(config, a, ks, abl);
    };
    for (int i = 0; i < copies && i < 8; ++i) run(i);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) run(i);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *us = (double)ms * 1e3 / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(X); hipFree(W); hipFree(C);
    return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess ? NTTS_OK : NTTS_EHIP;
}

// MFMA lane-layout probe (diagnostics): three products whose results spell out which (row, col) each
// (lane, reg) of the accumulator holds and whether A and B agree on the k slot.  Expected, under the
// layout documented in ntts/dev.h:  out[0][l][r] = (l>>4)*4 + r,  out[1][l][r] = l & 15,
// out[2][l][r] = ((l & 15) * 2 + 1) % 32 + 1.
NTTS_KERNEL(64) void mfma_probe_kernel(float* out) {
    const int l = lane_id(), g = l >> 4, c = l & 15;
    for (int probe = 0; probe < 3; ++probe) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) {
            float av, bv;
            if (probe == 0) { av = (g == 0 && j == 0) ? (float)c : 0.f; bv = (g == 0 && j == 0) ? 1.f : 0.f; }
            else if (probe == 1) { av = (g == 0 && j == 0) ? 1.f : 0.f; bv = (g == 0 && j == 0) ? (float)c : 0.f; }
            else { av = (float)(g * 8 + j + 1); bv = (g * 8 + j == (c * 2 + 1) % 32) ? 1.f : 0.f; }
            a[j] = (short)f2bf(av);
            b[j] = (short)f2bf(bv);
        }
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        d = mfma16(a, b, d);
        for (int r = 0; r < 4; ++r) out[(probe * 64 + l) * 4 + r] = d[r];
    }
}

// ---- launch-chain floor: n dependent launches of a kernel that does (almost) nothing, replayed from one hipGraph like the
//      decode step.  What a chain of n kernels costs before any of them moves a byte.
NTTS_KERNEL(256) void chain_probe_kernel(int* sink) {
    if (threadIdx.x == 0 && sink[blockIdx.x & 1023] == 0x7fffffff) sink[0] = 1;   // one 4-byte load per workgroup, never stores
}
// the smallest kernel that does what every decode kernel must: one HBM-cold 16-byte load per thread, then one 16-byte store
NTTS_KERNEL(256) void chain_touch_kernel(const u32x4* src, u32x4* dst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    u32x4 v = src[i];
    v[0] += 1;
    dst[i] = v;
}
extern "C" int ntts_k_launch_chain_probe(int32_t n_kernels, int32_t grid, int32_t block, int32_t iters, double* us_per_chain) {
    if (!us_per_chain || n_kernels < 1 || n_kernels > 4096 || grid < 1 || (block != 256 && block != -256) || iters < 1) return NTTS_EINVAL;
    const bool touch = block < 0;      // block = -256: chain_touch_kernel, every launch of a replay on its own (cold) 16 B x grid x 256 region
    block = 256;
    if (touch) {
        u32x4 *src = nullptr, *dst = nullptr;
        const size_t per = (size_t)grid * 256, total = per * n_kernels;
        hipStream_t st = nullptr; hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        if (hipMalloc((void**)&src, total * 16) != hipSuccess || hipMalloc((void**)&dst, per * 16) != hipSuccess) { hipFree(src); return NTTS_ENOMEM; }
        hipMemset(src, 1, total * 16);
        int rc = NTTS_EHIP;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            for (int i = 0; i < n_kernels; ++i) NTTS_LAUNCH((chain_touch_kernel), dim3(grid), dim3(256), st, (const u32x4*)(src + per * i), dst);
            if (hipStreamEndCapture(st, &g) == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
                hipEventRecord(e0, st);
                for (int i = 0; i < iters; ++i) hipGraphLaunch(ge, st);
                hipEventRecord(e1, st);
                if (hipEventSynchronize(e1) == hipSuccess) { float ms = 0; hipEventElapsedTime(&ms, e0, e1); *us_per_chain = (double)ms * 1e3 / iters; rc = NTTS_OK; }
                hipEventDestroy(e0); hipEventDestroy(e1);
            }
        }
        if (ge) hipGraphExecDestroy(ge);
        if (g) hipGraphDestroy(g);
        if (st) hipStreamDestroy(st);
        hipFree(src); hipFree(dst);
        (void)hipGetLastError();
        return rc;
    }
    int* sink = nullptr;
    hipStream_t st = nullptr;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    if (hipMalloc((void**)&sink, 4096) != hipSuccess) return NTTS_ENOMEM;
    hipMemset(sink, 0, 4096);
    int rc = NTTS_EHIP;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        for (int i = 0; i < n_kernels; ++i) NTTS_LAUNCH((chain_probe_kernel), dim3(grid), dim3(block), st, sink);
        if (hipStreamEndCapture(st, &g) == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
            hipEventRecord(e0, st);
            for (int i = 0; i < iters; ++i) hipGraphLaunch(ge, st);
            hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) == hipSuccess) {
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                *us_per_chain = (double)ms * 1e3 / iters;
                rc = NTTS_OK;
            }
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
    }
    if (ge) hipGraphExecDestroy(ge);
    if (g) hipGraphDestroy(g);
    if (st) hipStreamDestroy(st);
    hipFree(sink);
    (void)hipGetLastError();
    return rc;
}

NTTS_KERNEL(256) void silu_probe_kernel(const bf16_t* in, bf16_t* out, long n, int variant) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (variant == 2) {   // the packed form of the GEMM epilogues with their guard: pairs (i, i ^ 1), silu_fast below -86.5
        const float a = bf2f(in[i]), b = bf2f(in[(i ^ 1) < n ? (i ^ 1) : i]);
        if (a < -86.5f || b < -86.5f) out[i] = f2bf(silu_fast(a));
        else out[i] = f2bf(silu_fast2(f32x2{a, b})[0]);
        return;
    }
    out[i] = f2bf(variant ? silu_fast(bf2f(in[i])) : silu_f(bf2f(in[i])));
}
extern "C" int ntts_k_silu_probe(const void* in_bf16_dev, void* out_bf16_dev, int64_t n, int32_t variant) {
    if (!in_bf16_dev || !out_bf16_dev || n < 1) return NTTS_EINVAL;
    NTTS_LAUNCH((silu_probe_kernel), dim3((unsigned)((n + 255) / 256)), dim3(256), (hipStream_t)0, (const bf16_t*)in_bf16_dev,
                (bf16_t*)out_bf16_dev, (long)n, (int)variant);
    return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess ? NTTS_OK : NTTS_EHIP;
}

extern "C" int ntts_k_mfma_probe(float* out_dev_768) {
    if (!out_dev_768) return NTTS_EINVAL;
    NTTS_LAUNCH((mfma_probe_kernel), dim3(1), dim3(64), (hipStream_t)0, out_dev_768);
    return hipDeviceSynchronize() == hipSuccess ? NTTS_OK : NTTS_EHIP;
}
