// gemv.h -- the decode step's linears at SMALL batch (M <= 16 rows: BASELINE.json configs[1], batch 1):
//     out[M, N] = f(X)[M, K] * W[N, K]^T      f = identity, or the fused "sum split-K slabs + residual + RMSNorm" prologue
//
// Replaces the same nn.Linear calls as gemm.h (hf:models/qwen2/modeling_qwen2.py:46-48,206-208,233,464-465) with the same
// rounding contract (fp32 accumulate, one rounding to bf16 where the Linear output is materialised), for the regime
// where the GEMM tiles of gemm.h make no sense: at M = 1 a 64 x 64 tile is 98 % padding, the step is a chain of GEMVs and
// the only thing that matters is (a) how fast every weight byte streams in ONCE and (b) how many launches the chain has
// (at batch 1 a layer's weights are 30 MB = 6 us at 5 TB/s, while every launch boundary costs ~2 us).
//
// Structure (MI355X_MICROARCH.md / cdna_hip_programming.md "GEMV / M <= 16 decode weights"):
//   * one WAVE owns 16 output features over a K slice; no LDS staging of W, no barrier in the main loop: the W fragments go
//     HBM -> VGPR with the non-temporal policy, a whole chunk of k-tiles requested at once (double-buffered chunks), so a
//     wave has 8-28 KB of its private weight stream in flight.  With the engine's tile-major weight layout a wave's
//     16 rows x 128 B of one k-tile are 2 KB CONTIGUOUS: two fully coalesced 1 KB wave-loads.
//   * the matrix core is used as a 16 x 16 x 32 dot-product engine: A = W fragment (16 features x 32 k), B = X fragment
//     (32 k x 16 token rows, rows >= M are clamped duplicates and never stored).  Lane (l15, g) holds k = g*16 .. g*16+15 of
//     its row for BOTH operands (first MFMA: the low 8, second: the high 8): any k order works as long as A and B agree.
//   * split-K over gridDim.y writes fp32 slabs [split][slab_rows][N] that the consumer reduces -- same hand-off as gemm.h.
//   * PRO (prologue fusion): the consumer of a split-K GEMM normally is add_rmsnorm (norm.h); here every workgroup redoes
//     that little piece of work itself -- sum the slabs in order, round (the Linear output), add the residual, round,
//     RMSNorm, times the norm weight -- for all M rows into LDS, and block 0 writes the new residual stream.  At M <= 8
//     that is a few KB per workgroup out of L2, and it removes two of a layer's seven launches (and the embedding
//     gather of layer 0).  The arithmetic and its ORDER are add_rmsnorm_kernel<2>'s (norm.h rmsnorm_row_wave).
#pragma once
#include <ntts/dev.h>

#include "gemm.h"
#include "norm.h"

namespace ntts {

constexpr int kGemvRows = 16;          // token rows one MFMA covers (rows >= M are padding)
constexpr int kGemvXld = 1024 + 8;     // LDS row stride of the prologue's X panel (bf16): H <= 1024, +16 B de-aliases the banks

struct GemvArgs {
    const bf16_t* X;       // [M][ldx] bf16 (PRO = false)
    long ldx;
    NormArgs pro;          // PRO = true: what add_rmsnorm would have been given (slabs | o_bf16 | gather) + resid_in/out + norm_w
    const bf16_t* W;
    long ldw;
    int w_tile_major;
    void* out;
    long ldo;
    long slab_rows;        // EPI_SPLITK: rows per slab of `out`
    int M, N, K;           // M <= 16, N % 16 == 0, K % 64 == 0
    int k_tiles_per_split;
    // EPI_ARGMAX (lm_head)
    float* part_val;
    int* part_idx;
    int part_stride;       // partials per row = N / 16
    const int* mask_eos;
    float* logits;
    long ld_logits;
    bf16_t* logits_bf16;
    long ld_logits_bf16;
};

template <int EPI, bool PRO>
NTTS_KERNEL(256) void gemv_kernel(GemvArgs p) {
    constexpr int CH = PRO ? 7 : 4;                        // k-tiles per register chunk (two chunks live: W only, or W + X)
    NTTS_SHARED bf16_t xs[PRO ? kGemvRows * kGemvXld : 8];
    const int lane = lane_id(), w = wave_id();
    const int g = lane >> 4, l15 = lane & 15;
    const int f0 = (blockIdx.x * 4 + w) * 16;              // this wave's 16 features
    const int ktiles = p.K >> 6;
    const int kt0 = blockIdx.y * p.k_tiles_per_split;
    int nk = ktiles - kt0;
    if (nk > p.k_tiles_per_split) nk = p.k_tiles_per_split;
    const bool active = f0 < p.N && nk > 0;                // wave-uniform

    // ---- this wave's weight stream: requested BEFORE the prologue so that it is in flight while the norm is computed
    const bf16_t* wbase = nullptr;
    long wstep = 64;
    if (active) {
        if (p.w_tile_major) { wbase = p.W + (long)(f0 >> 6) * 64 * p.K + ((f0 & 63) + l15) * 64 + g * 16; wstep = 4096; }
        else wbase = p.W + (long)(f0 + l15) * p.ldw + g * 16;
    }
    bf16x8 wa[2][CH][2], xb[2][PRO ? 1 : CH][2];
    auto load_w = [&](int c, auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int kt = c * CH + j;
            const bf16_t* src = wbase + (long)(kt0 + (kt < nk ? kt : nk - 1)) * wstep;   // past the slice: a duplicate, skipped below
            wa[BUF][j][0] = ld16_nt<bf16x8>(src);
            wa[BUF][j][1] = ld16_nt<bf16x8>(src + 8);
        }
    };
    const bf16_t* xrow = nullptr;
    if constexpr (!PRO) {
        int m = l15 < p.M ? l15 : p.M - 1;
        xrow = p.X + (long)m * p.ldx + g * 16;
    }
    auto load_x = [&](int c, auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;
        if constexpr (!PRO) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int kt = c * CH + j;
                const bf16_t* src = xrow + (long)(kt0 + (kt < nk ? kt : nk - 1)) * 64;
                xb[BUF][j][0] = ld16<bf16x8>(src);
                xb[BUF][j][1] = ld16<bf16x8>(src + 8);
            }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    const int nchunks = (nk + CH - 1) / CH;
    if (active) { load_w(0, B0{}); load_x(0, B0{}); }

    if constexpr (PRO) {
        // rows m = w, w + 4, ...: one wave per row, exactly add_rmsnorm_kernel<2>'s arithmetic; padding rows are zero
        const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
        for (int m = w; m < kGemvRows; m += 4) {
            bf16_t* dst = xs + m * kGemvXld;
            if (m < p.M) rmsnorm_row_wave<2>(p.pro, m, true, writer, dst);
            else
                for (int c = lane; c < (p.K >> 3); c += 64) *(bf16x8*)(dst + c * 8) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        sync();
    }
    if (!active) return;

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int c, auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int kt = c * CH + j;
            if (kt < nk) {                                   // wave-uniform
                bf16x8 x0, x1;
                if constexpr (PRO) {
                    const bf16_t* xp = xs + l15 * kGemvXld + (kt0 + kt) * 64 + g * 16;
                    x0 = ld16<bf16x8>(xp);
                    x1 = ld16<bf16x8>(xp + 8);
                } else {
                    x0 = xb[BUF][j][0];
                    x1 = xb[BUF][j][1];
                }
                acc = mfma16(wa[BUF][j][0], x0, acc);
                acc = mfma16(wa[BUF][j][1], x1, acc);
            }
        }
    };
    for (int c = 0; c < nchunks; c += 2) {
        if (c + 1 < nchunks) { load_w(c + 1, B1{}); load_x(c + 1, B1{}); }
        compute(c, B0{});
        if (c + 1 < nchunks) {
            if (c + 2 < nchunks) { load_w(c + 2, B0{}); load_x(c + 2, B0{}); }
            compute(c + 1, B1{});
        }
    }

    // ---- epilogue: lane (g, l15) holds features f0 + g*4 + r (r = 0..3) of token row l15
    const int m = l15;
    const bool mok = m < p.M;
    const int nf = f0 + g * 4;
    if constexpr (EPI == EPI_SPLITK) {
        if (mok) *(f32x4*)((float*)p.out + ((long)blockIdx.y * p.slab_rows + m) * p.ldo + nf) = acc;
    } else if constexpr (EPI == EPI_SILU_MUL) {
        // packed rows (backbone.cpp gu_map): rows 0-7 of the wave's 16 = gate, rows 8-15 = up of the same 8 features
        float up[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) up[r] = shfl_xor(acc[r], 32);
        if (g < 2 && mok) {
            alignas(8) bf16_t o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gt = rbf(acc[r]), u = rbf(up[r]);      // gate_proj / up_proj outputs (bf16)
                o[r] = f2bf(rbf(silu_f(gt)) * u);                  // act_fn output (bf16), product (bf16)
            }
            const int fb = (f0 >> 6) * 32 + ((f0 & 63) >> 4) * 8 + g * 4;
            *(u32x2*)((bf16_t*)p.out + (long)m * p.ldo + fb) = *(u32x2*)&o[0];
        }
    } else if constexpr (EPI == EPI_ARGMAX) {
        float best = -INFINITY;
        int bidx = 0x7fffffff;
        const int meos = (mok && p.mask_eos) ? p.mask_eos[m] : 0;     // eos id + 1, or 0
        alignas(8) bf16_t lo[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nf + r;
            float v = rbf(acc[r]);                                   // lm_head output is bf16, then .float()
            if (n == meos - 1) v = -INFINITY;
            lo[r] = f2bf(v);
            if (mok && p.logits) p.logits[(long)m * p.ld_logits + n] = v;
            if (v > best) { best = v; bidx = n; }                     // ascending n + strict '>' = first max wins
        }
        if (p.logits_bf16 && mok) *(u32x2*)(p.logits_bf16 + (long)m * p.ld_logits_bf16 + nf) = *(u32x2*)&lo[0];
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) {
            const float ov = shfl_xor(best, sh);
            const int oi = shfl_xor(bidx, sh);
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        if (mok && g == 0) {
            const long pi = (long)m * p.part_stride + (f0 >> 4);
            p.part_val[pi] = best;
            p.part_idx[pi] = bidx;
        }
    } else {
        static_assert(EPI == EPI_SPLITK || EPI == EPI_SILU_MUL || EPI == EPI_ARGMAX, "epilogues of the small-batch kernel");
    }
}

// number of split-K slabs gemv_launch produces for (K, ksplit)
inline int gemv_nsplit(int K, int ksplit) { return gemm_nsplit(K, ksplit); }

template <int EPI, bool PRO>
inline void gemv_launch(GemvArgs p, int ksplit, hipStream_t s) {
    const int ktiles = p.K / 64;
    if (ksplit < 1) ksplit = 1;
    if (ksplit > ktiles) ksplit = ktiles;
    p.k_tiles_per_split = (ktiles + ksplit - 1) / ksplit;
    const int nsplit = (ktiles + p.k_tiles_per_split - 1) / p.k_tiles_per_split;
    if constexpr (EPI == EPI_ARGMAX) p.part_stride = p.N / 16;
    NTTS_LAUNCH((gemv_kernel<EPI, PRO>), dim3((p.N + 63) / 64, nsplit), dim3(256), s, p);
}

}  // namespace ntts
