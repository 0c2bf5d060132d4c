// gemm_wreg.h -- decode-batch linears with the WEIGHT slice held in registers and the batch's activations in an LDS ring:
//     out[M, N] = X[M, K] * W[N, K]^T  (+ epilogue),   M = the decode batch (<= 256 rows per workgroup), K slice <= 896
//
// Replaces the same nn.Linear calls as gemm.h (hf:models/qwen2/modeling_qwen2.py:46-48 gate/up/down, :206-208,233 q/k/v/o) with
// the same rounding contract (fp32 accumulate, ONE rounding where the Linear output is materialised).
//
// Why a third GEMM structure (measured on MI355X, tools/ubench/loadpath*.hip, profiles/r02i_loadpath_*.txt):
//   * a short HBM-cold stream is cheap when nothing stalls it -- 256 workgroups x 64 KB (16.8 MB, a layer's gate/up matrix)
//     land 1.5-2 us after the launch gap -- but the ring of gemm.h couples it to the L2-resident X operand: every k-tile
//     iteration ends at a barrier that waits for the SLOWEST of the tile's requests, i.e. an HBM miss of W (~1 us), with one
//     or two tiles in flight.  The batch-256 gate/up GEMM therefore spends 8.7 us in its k-loop on 17 MB of weights.
//   * L2-resident lines stream into one CU at 100-130 GB/s once >= 32-64 KB are in flight; HBM-cold ones at whatever share
//     of the chip's ~6 TB/s the CU gets.
// So the two operands are decoupled here:
//   * W: a FEATURE wave owns 16 output features and requests its WHOLE K slice HBM -> VGPR at kernel entry (14 k-tiles x
//     2 x 16 B per lane = 112 VGPRs; the engine's tile-major weight layout makes a wave's 16 rows x 128 B of one k-tile 2 KB
//     contiguous).  Nothing ever waits on these loads except their first use (counted s_waitcnt, in issue order): the weight
//     stream runs at the rate HBM delivers it, from the first instruction of the kernel.
//   * X: LOADER waves (separate waves: their LDS-DMA queue is not behind the feature waves' weight loads, a wave's loads
//     return in order) bring the batch's rows in through a ring of NSX k-tile slots [BM rows][64 k] by LDS-DMA in full 128-byte
//     lines (fragment-shaped loads straight to VGPRs would touch twice the lines per instruction), swizzled on the source
//     side exactly like gemm.h; one barrier per k-tile, NSX - 1 tiles (96 KB at BM = 256) in flight.
//   * matrix core as in gemm.h / gemv.h: A = W fragment (16 features x 32 k), B = X fragment (32 k x 16 rows); lane (g, l15)
//     holds k-chunks g and 4 + g of its row for both operands; D: column = row l15 of the m-fragment, rows = features g*4 + r.
// Geometry: workgroup = 4 feature waves (64 features) + 4 loader waves; grid = (N / 64, K splits, M / BM).  EPI_SPLITK writes fp32
// slabs the consumer reduces (same hand-off as gemm.h); EPI_SILU_MUL needs the whole K in one slice (K <= 896).
#pragma once
#include <ntts/dev.h>

#include "gemm.h"

namespace ntts {

constexpr int kWregKT = 14;   // k-tiles of 64 a feature wave holds: 14 x 64 = 896 (NeuTTS-Air's hidden size), 112 VGPRs

// ABL (micro-benchmark ablation, always 0 in the product): 1 = no LDS reads / MFMA, 2 = no LDS-DMA, 4 = no stores, 8 = no weight loads
template <int EPI, int MF, int NSX = 4, int ABL = 0>
NTTS_KERNEL(512) void gemm_wreg_kernel(GemmArgs p) {
    constexpr int KT = kWregKT;
    constexpr int FW = 4, LW = 4;
    constexpr int BM = MF * 16;
    constexpr int SLOT = BM * 64;              // elements per ring slot: BM rows x 64 k (128-byte rows)
    constexpr int NINST = BM / 8;              // 1-KB wave-instructions per slot (8 rows x 128 B each)
    static_assert(NINST % LW == 0, "loader split");
    constexpr int PER_L = NINST / LW;
    static_assert(NSX >= 2 && (NSX - 2) * PER_L <= 63, "vmcnt range");
    NTTS_SHARED bf16_t xs[NSX * SLOT];

    const int lane = lane_id(), w = wave_id();
    const int g = lane >> 4, l15 = lane & 15;
    const int m0 = blockIdx.z * BM;
    const int split = blockIdx.y;
    const int ktiles = p.K >> 6;
    const int kt0 = split * p.k_tiles_per_split;
    int nk = ktiles - kt0;
    if (nk > p.k_tiles_per_split) nk = p.k_tiles_per_split;     // 1 .. KT (launcher)

    if (w >= FW) {
        // ---- loader waves: the X ring.  Slot row rho <-> batch row m0 + rho (clamped: the clamped duplicates are computed on
        //      and never stored); the 16-byte chunk held at physical position lane % 8 is logical chunk (lane % 8) ^ swz(rho).
        const int lw = w - FW;
        const char* src[PER_L];
#pragma unroll
        for (int i = 0; i < PER_L; ++i) {
            const int inst = lw + i * LW;
            const int rho = inst * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((rho >> 1) & 7);
            int m = m0 + rho;
            if (m > p.M - 1) m = p.M - 1;
            src[i] = (const char*)p.X + ((long)m * p.ldx + (long)kt0 * 64) * 2 + c * 16;
        }
        auto stage = [&](int kt, int slot) {
            if constexpr (ABL & 2) return;
#pragma unroll
            for (int i = 0; i < PER_L; ++i) glds16(src[i] + (long)kt * 128, xs + slot * SLOT + (lw + i * LW) * 512);
        };
#pragma unroll
        for (int s = 0; s < NSX - 1; ++s)
            if (s < nk) stage(s, s);
        int slot = NSX - 1;                     // slot of tile kt + NSX - 1
        for (int kt = 0; kt < nk; ++kt) {
            // tile kt has landed; tiles kt+1 .. kt+NSX-2 may stay in flight (at the tail the plain drain costs nothing extra)
            if (kt + NSX - 2 < nk) wait_vmem_le<(NSX - 2) * PER_L>(); else wait_vmem();
            sync_keep_dma();                    // barrier kt: tile kt visible to the feature waves, slot of tile kt-1 free
            if (kt + NSX - 1 < nk) stage(kt + NSX - 1, slot);
            slot = slot + 1 == NSX ? 0 : slot + 1;
        }
        return;
    }

    // ---- feature waves: the whole weight slice first (branch-free: a slice shorter than KT re-reads its last tile, unused)
    const int f0r = (blockIdx.x * FW + w) * 16;
    const bool active = f0r < p.N;              // wave-uniform
    const int f0 = active ? f0r : 0;            // an inactive wave streams (and discards) group 0: it still takes part in the barriers
    const bf16_t* wbase;
    long wstep = 64;
    if (p.w_tile_major) { wbase = p.W + (long)(f0 >> 6) * 64 * p.K + (long)kt0 * 4096 + ((f0 & 63) + l15) * 64 + g * 8; wstep = 4096; }
    else wbase = p.W + (long)(f0 + l15) * p.ldw + (long)kt0 * 64 + g * 8;
    bf16x8 wa[KT][2];
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const bf16_t* s = wbase + (long)(j < nk ? j : nk - 1) * wstep;
        if constexpr (ABL & 8) { wa[j][0] = wa[j][1] = bf16x8{(short)j, 1, 2, 3, 4, 5, 6, (short)lane}; continue; }
        wa[j][0] = ld16<bf16x8>(s);             // k-chunk g      (k = g*8 .. g*8+7 of the tile)
        wa[j][1] = ld16<bf16x8>(s + 32);        // k-chunk 4 + g  (k = 32 + g*8 ..)
    }

    f32x4 acc[MF];
#pragma unroll
    for (int a = 0; a < MF; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    // X fragment of m-fragment a: slot row a*16 + l15, chunks g / 4+g; swz(a*16 + l15) = (l15 >> 1) & 7 for every a
    const int sw = (l15 >> 1) & 7;
    const int xo0 = l15 * 64 + ((g ^ sw) << 3), xo1 = l15 * 64 + (((4 + g) ^ sw) << 3);
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        if (j < nk) {                           // block-uniform
            sync_keep_dma();                    // barrier j (the loaders arrive once tile j has landed)
            if constexpr (ABL & 1) { acc[j % MF][0] += bf2f((bf16_t)wa[j][0][0]) + bf2f((bf16_t)wa[j][1][1]); continue; }
            const bf16_t* base = xs + (j % NSX) * SLOT;
            // Software pipeline over groups of G m-fragments: the LDS reads of group q + PD are issued before the MFMAs of
            // group q (left to itself hipcc keeps two fragment registers and alternates read -> wait -> MFMA: 0.9 us per
            // k-tile instead of the 0.25 us either pipe needs).  Within a group the two MFMAs of one accumulator are G apart.
            constexpr int G = 2, NG = MF / G, PD = 2;
            static_assert(MF % G == 0 && NG > PD, "fragment groups");
            bf16x8 xb[PD + 1][G][2];
            auto ldg = [&](int q, bf16x8 (&d)[G][2]) {
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    d[u][0] = ld16<bf16x8>(base + (q * G + u) * 1024 + xo0);
                    d[u][1] = ld16<bf16x8>(base + (q * G + u) * 1024 + xo1);
                }
            };
#pragma unroll
            for (int q = 0; q < PD; ++q) ldg(q, xb[q]);
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                if (q + PD < NG) ldg(q + PD, xb[(q + PD) % (PD + 1)]);
                sched_fence();
#pragma unroll
                for (int u = 0; u < G; ++u) acc[q * G + u] = mfma16(wa[j][0], xb[q % (PD + 1)][u][0], acc[q * G + u]);
#pragma unroll
                for (int u = 0; u < G; ++u) acc[q * G + u] = mfma16(wa[j][1], xb[q % (PD + 1)][u][1], acc[q * G + u]);
                sched_fence();
            }
        }
    }
    if (!active) return;                        // (no barrier below)
    if constexpr (ABL & 4) {
        float t = 0.f;
#pragma unroll
        for (int a = 0; a < MF; ++a) t += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
        if (t != 12345.678f) return;            // keeps the accumulators live without storing
    }

    // ---- epilogue: lane (g, l15) holds features f0 + g*4 + r (r = 0..3) of batch row m0 + a*16 + l15
    const int nf = f0 + g * 4;
    if constexpr (EPI == EPI_SPLITK) {
#pragma unroll
        for (int a = 0; a < MF; ++a) {
            const int m = m0 + a * 16 + l15;
            if (m < p.M) *(f32x4*)((float*)p.out + ((long)split * p.M + m) * p.ldo + nf) = acc[a];
        }
    } else if constexpr (EPI == EPI_SILU_MUL) {
        // packed rows (backbone.cpp gu_map): rows 0-7 of the wave's 16 = gate, rows 8-15 = up of the same 8 features
        const int fb = (f0 >> 6) * 32 + ((f0 & 63) >> 4) * 8 + g * 4;
#pragma unroll
        for (int a = 0; a < MF; ++a) {
            const int m = m0 + a * 16 + l15;
            float up[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) up[r] = shfl_xor(acc[a][r], 32);
            if (g < 2 && m < p.M) {
                alignas(8) bf16_t o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gt = rbf(acc[a][r]), u = rbf(up[r]);   // gate_proj / up_proj outputs (bf16)
                    o[r] = f2bf(rbf(silu_f(gt)) * u);                  // act_fn output (bf16), product (bf16)
                }
                *(u32x2*)((bf16_t*)p.out + (long)m * p.ldo + fb) = *(u32x2*)&o[0];
            }
        }
    } else {
        static_assert(EPI == EPI_SPLITK || EPI == EPI_SILU_MUL, "epilogues of the weight-in-registers kernel");
    }
}

// k-tiles per split this kernel can take for (K, requested ksplit), or 0 when the slice would not fit the registers
inline int gemm_wreg_ktps(int K, int ksplit) {
    const int ktiles = K / 64;
    if (ksplit < 1) ksplit = 1;
    if (ksplit > ktiles) ksplit = ktiles;
    const int per = (ktiles + ksplit - 1) / ksplit;
    return per <= kWregKT ? per : 0;
}

// returns false (nothing launched) when the shape does not fit: K slice > 896, or a non-split epilogue with K > 896
template <int EPI>
inline bool gemm_wreg_launch(GemmArgs p, int ksplit, hipStream_t s) {
    if (p.wscale || (p.K % 64) || p.K < 64 || (p.N % 16)) return false;      // bf16 operands only
    const int per = gemm_wreg_ktps(p.K, ksplit);
    if (!per) return false;
    const int ktiles = p.K / 64, nsplit = (ktiles + per - 1) / per;
    if (EPI != EPI_SPLITK && nsplit != 1) return false;
    p.k_tiles_per_split = per;
    const unsigned nb = (unsigned)((p.N + 63) / 64);
    if (p.M <= 128) NTTS_LAUNCH((gemm_wreg_kernel<EPI, 8>), dim3(nb, nsplit, (p.M + 127) / 128), dim3(512), s, p);
    else NTTS_LAUNCH((gemm_wreg_kernel<EPI, 16>), dim3(nb, nsplit, (p.M + 255) / 256), dim3(512), s, p);
    return true;
}

}  // namespace ntts
