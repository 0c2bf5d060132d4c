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
//   * a workgroup = 4 FEATURE waves + 4 HELPER waves.  A feature wave owns 16 output features over the workgroup's K slice:
//     no LDS staging of W, no barrier in its main loop; its W fragments go HBM -> VGPR, the WHOLE slice (<= 16 k-tiles =
//     32 KB per wave) requested at kernel entry, before anything else (default cache policy: the non-temporal one measured
//     SLOWER here, lm_head 76.5 vs 65.7 us, gate/up 10.2 vs 9.1 -- profiles/r02b_sweep_b1.log).  With the engine's tile-major weight layout a
//     wave's 16 rows x 128 B of one k-tile are 2 KB CONTIGUOUS: two fully coalesced 1 KB wave-loads.
//   * the helper waves build the X panel of the slice in LDS meanwhile (so that X costs the feature waves no global loads
//     and no registers): a plain copy of the bf16 rows, or -- PRO, prologue fusion -- what add_rmsnorm (norm.h) would have
//     computed: sum the producer's split-K slabs in order, round (the Linear output), add the residual, round, RMSNorm,
//     times the norm weight; block 0 also writes the new residual stream.  Every workgroup redoes that little piece of
//     work (at M <= 8 a few KB out of L2), which removes two of a layer's seven launches and the embedding gather of
//     layer 0 (the QKV projection's own kernel, qkv_rope.h gemv_qkv_rope_kernel, shares this prologue).  The arithmetic and its ORDER are add_rmsnorm_kernel<2>'s (norm.h rmsnorm_row_wave).  Because the helpers
//     are separate waves, their (L2) loads do not queue behind the feature waves' HBM weight stream (a wave's loads
//     return in order) and their registers cost the feature waves nothing.
//   * the matrix core is used as a 16 x 16 x 32 dot-product engine: A = W fragment (16 features x 32 k), B = X fragment
//     (32 k x 16 token rows; rows >= M hold whatever the LDS held: a B column only feeds its own output column, which is
//     never stored).  Lane (l15, g) holds k = g*16 .. g*16+15 of its row for BOTH operands (first MFMA: the low 8, second:
//     the high 8): any k order works as long as A and B agree.
//   * split-K over gridDim.y writes fp32 slabs [split][slab_rows][N] that the consumer reduces -- same hand-off as gemm.h.
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
    const float* xslabs;   // PRO = false alternative to X: [n_xslab][slab_rows][ldx] fp32 partials (the context-split attention's
    int n_xslab;           //   chunk outputs, attn_decode.h): X = bf16(sum over the slabs, in order) -- one rounding, the attention output's
    NormArgs pro;          // PRO = true: what add_rmsnorm would have been given (slabs | o_bf16 | gather) + resid_in/out + norm_w
    const bf16_t* W;
    long ldw;
    int w_tile_major;
    void* out;
    long ldo;
    long slab_rows;        // EPI_SPLITK: rows per slab of `out`
    int M, N, K;           // M <= 16, N % 16 == 0, K % 64 == 0 (PRO: K = pro.H <= 1024)
    int k_tiles_per_split; // <= 16
    // EPI_ARGMAX (lm_head)
    float* part_val;
    int* part_idx;
    int part_stride;       // partials per row = N / 16
    const int* mask_eos;
    int eos_col1;          // compacted head (gemm.h GemmArgs::eos_col1): the EOS row's column + 1, 0 = column is the token id
    int n_valid;           // columns >= n_valid are padding of a compacted head (N is a multiple of 16): their logits are -inf
    float* logits;
    long ld_logits;
    bf16_t* logits_bf16;
    long ld_logits_bf16;
    // F8 kernels (fp8 model, gemm.h GemmArgs): X and W hold e4m3 BYTES (ldx / K count elements = bytes; a 128-byte k-tile is 128 k-values, K % 128 == 0),
    // the panel in LDS holds bytes (PRO: pro.out_fp8_inv quantises the normalised rows), out = acc * (xscale * wscale[n]); EPI_SILU_MUL emits e4m3
    // bytes when out_fp8_inv > 0 (down_proj's input).  Byte for byte the staging is the bf16 kernel's: a lane's 32 bytes per k-tile are 32 k-values
    // instead of 16 and feed four fp8 MFMAs instead of two bf16 ones.
    const float* wscale;
    float xscale, out_fp8_inv;
    unsigned long long* tl;   // diagnostics (ntts_backbone_gemv_timeline): [workgroups][16] phase timestamps -- slots 0..7 feature wave 0,
                              // 8..15 helper wave 0; null in the product path.  With it set the feature wave also WAITS for its whole weight
                              // slice before the barrier (so that "weights landed" has a timestamp), which the product path never does
};

// KT = k-tiles of the workgroup's K slice this instantiation is unrolled for (the slice may be shorter: the surplus loads
// re-read the last tile and their fragments are masked to zero -- NO per-element branches: a load guarded by a run-time
// condition makes hipcc wait for each element before it requests the next, cdna_hip_programming.md "three .s-level traps" (c),
// which is what a first version of this kernel did: 66 s_waitcnt in 770 instructions, slower than the LDS-DMA tiles).
// FW = feature waves per workgroup, + 4 helper waves.  gate/up takes FW = 3 (203 workgroups of 48 features instead of 152 of 64: 86 instead
// of 114 KB of weights per CU; 10.8 -> 10.0 us at batch 1, round 3); FW = 2 (304 workgroups -- more than CUs, two rounds) costs 16.1-16.4 us
// (profiles/r02f_sweep_b1_nw16_late_fw2.log, r03g_sweep_b1_gemv_prologue.log).
template <int EPI, bool PRO, int KT, int FW = 4, bool F8 = false>
NTTS_KERNEL((FW + 4) * 64) void gemv_kernel(GemvArgs p) {
    NTTS_SHARED bf16_t xs[kGemvRows * kGemvXld];
    constexpr int ESZ = F8 ? 1 : 2;                               // bytes per operand element
    const int lane = lane_id(), w = wave_id();
    const int g = lane >> 4, l15 = lane & 15;
    const int ktiles = F8 ? p.K >> 7 : p.K >> 6;                  // 128-byte k-tiles
    const int kt0 = blockIdx.y * p.k_tiles_per_split;
    int nk = ktiles - kt0;
    if (nk > p.k_tiles_per_split) nk = p.k_tiles_per_split;       // <= KT (launcher)
    const int col0 = PRO ? 0 : kt0 * 64;                          // K index held by panel column 0 (PRO: the panel is the whole row)

    const long tlb = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 16;
    auto mark = [&](int slot) { if (p.tl && lane == 0) p.tl[tlb + slot] = now_ticks(); };
    if (w >= FW) {
        // ---- helper waves: the X panel.  Row m is built by helper (m & 3); rows >= M are left alone.
        const int hw = w - FW;
        if (hw == 0) mark(8);
        if constexpr (PRO) {
            const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
            bool first = true;                                    // the issue barrier (norm.h): once per wave, behind its first row's requests
            for (int m = hw; m < p.M; m += 4) { rmsnorm_row_wave<2, 8, true>(p.pro, m, true, writer, xs + m * kGemvXld, first); first = false; }
            if (first) sync_keep_dma();                           // (a helper without a row)
        } else {
            const int nch = nk * 8;                               // 16-byte chunks per row of the slice: <= 128 = 2 per lane
            for (int m = hw; m < p.M; m += 4) {
                const int c0 = lane, c1 = lane + 64;                // both loads first (clamped addresses), then the guarded stores
                if (!F8 && p.xslabs) {                              // (wave-uniform) sum the fp32 slabs in order, round once (bf16 model only)
                    float a0[8], a1[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) a0[e] = a1[e] = 0.f;
                    for (int sl = 0; sl < p.n_xslab; ++sl) {
                        const float* src = p.xslabs + ((long)sl * p.slab_rows + m) * p.ldx + kt0 * 64;
                        const f32x4 u0 = ld16<f32x4>(src + (c0 < nch ? c0 : 0) * 8), u1 = ld16<f32x4>(src + (c0 < nch ? c0 : 0) * 8 + 4);
                        const f32x4 w0 = ld16<f32x4>(src + (c1 < nch ? c1 : 0) * 8), w1 = ld16<f32x4>(src + (c1 < nch ? c1 : 0) * 8 + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a0[e] += u0[e]; a0[4 + e] += u1[e]; a1[e] += w0[e]; a1[4 + e] += w1[e]; }
                    }
                    bf16x8 t0, t1;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { t0[e] = (short)f2bf(a0[e]); t1[e] = (short)f2bf(a1[e]); }
                    if (c0 < nch) *(bf16x8*)(xs + m * kGemvXld + c0 * 8) = t0;
                    if (c1 < nch) *(bf16x8*)(xs + m * kGemvXld + c1 * 8) = t1;
                    continue;
                }
                const bf16_t* src = (const bf16_t*)((const char*)p.X + ((long)m * p.ldx) * ESZ) + kt0 * 64;   // (a k-tile is 128 bytes either way)
                const bf16x8 t0 = ld16<bf16x8>(src + (c0 < nch ? c0 : 0) * 8);
                const bf16x8 t1 = ld16<bf16x8>(src + (c1 < nch ? c1 : 0) * 8);
                if (c0 < nch) *(bf16x8*)(xs + m * kGemvXld + c0 * 8) = t0;
                if (c1 < nch) *(bf16x8*)(xs + m * kGemvXld + c1 * 8) = t1;
            }
        }
        if (hw == 0) mark(9);
        sync();
        if (hw == 0) mark(10);
        return;
    }

    // ---- feature waves: request the whole weight slice (branch-free), then wait for the panel
    if (w == 0) mark(0);
    const int f0r = (blockIdx.x * FW + w) * 16;
    const bool active = f0r < p.N;                                // wave-uniform
    const int f0 = active ? f0r : 0;                              // an inactive wave streams (and discards) group 0: no branch around the loads
    const bf16_t* wbase;
    long wstep = 64;
    if (p.w_tile_major) { wbase = (const bf16_t*)((const char*)p.W + (long)(f0 >> 6) * 64 * p.K * ESZ) + ((f0 & 63) + l15) * 64 + g * 16; wstep = 4096; }
    else wbase = (const bf16_t*)((const char*)p.W + (long)(f0 + l15) * p.ldw * ESZ) + g * 16;
    bf16x8 wa[KT][2];
    if constexpr (PRO) sync_keep_dma();                           // the helpers' requests go first (norm.h issue_barrier).  Round 5 re-measured the other order --
                                                                  // weights requested before the prologue's rows are out -- at batch 1 / 8: step 0.807 -> 0.843 (gate/up), 0.820 (QKV),
                                                                  // 0.856 (both); 1.029 -> 1.049 / 1.066 (profiles/r05j_sweep_b*_gemv_weights_first.log): not kept
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const bf16_t* src = wbase + (long)(kt0 + (j < nk ? j : nk - 1)) * wstep;
        wa[j][0] = ld16<bf16x8>(src);
        wa[j][1] = ld16<bf16x8>(src + 8);
    }
    if (w == 0) mark(1);
    if (p.tl) { wait_vmem(); if (w == 0) mark(2); }
    sync();                                                       // the panel is complete
    if (w == 0) mark(3);
    if (!active) return;

    // One accumulator chain (the fp32 order of the K sum is part of the result), so the chain's length is what this loop costs:
    // the X fragments are read from LDS kXPf k-tiles AHEAD of the matrix-core instructions that use them -- left to itself hipcc
    // issues each tile's two ds_reads only after the previous tile's MFMAs (no registers to spare for hoisting them next to 128
    // VGPRs of weights): 16 x (LDS latency + 2 MFMAs) = 1.57 us at K = 896 (tools/gemv_timeline.py), with the ring the MFMAs alone
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bf16_t* xrow = xs + l15 * kGemvXld + g * 16 - col0;
    // (not for the lm_head: 13.6 k workgroups stream 390 MB and what matters there is how fast a CU turns workgroups over -- with the
    //  ring, 182 instead of 142 VGPRs and the pinned order, it went 70 -> 90 us; profiles/r03g_sweep_b1_gemv_prologue.log)
    constexpr int kXPf = EPI == EPI_ARGMAX ? 1 : KT < 4 ? KT : 4;
    bf16x8 xq[kXPf][2];
    auto xload = [&](int j, bf16x8 (&d)[2]) {
        const bf16_t* xp = xrow + (kt0 + (j < nk ? j : nk - 1)) * 64;
        d[0] = ld16<bf16x8>(xp);
        d[1] = ld16<bf16x8>(xp + 8);
    };
#pragma unroll
    for (int j = 0; j < kXPf; ++j) xload(j, xq[j]);
    if constexpr (kXPf > 1) sched_fence();
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const short keep = j < nk ? (short)-1 : (short)0;         // surplus tiles contribute 0 * x
        const bf16x8 x0 = xq[j % kXPf][0], x1 = xq[j % kXPf][1];
        if (j + kXPf < KT) xload(j + kXPf, xq[j % kXPf]);
        if constexpr (kXPf > 1) sched_fence();                    // (hipcc sinks the reads back next to their MFMAs otherwise)
        if constexpr (F8) {
            const i64x2 a0 = __builtin_bit_cast(i64x2, wa[j][0] & keep), a1 = __builtin_bit_cast(i64x2, wa[j][1] & keep);
            const i64x2 b0 = __builtin_bit_cast(i64x2, x0), b1 = __builtin_bit_cast(i64x2, x1);
            acc = mfma16_fp8(a0[0], b0[0], acc);
            acc = mfma16_fp8(a0[1], b0[1], acc);
            acc = mfma16_fp8(a1[0], b1[0], acc);
            acc = mfma16_fp8(a1[1], b1[1], acc);
        } else {
            acc = mfma16(wa[j][0] & keep, x0, acc);
            acc = mfma16(wa[j][1] & keep, x1, acc);
        }
    }

    if (p.tl && w == 0 && lane == 0) p.tl[tlb + 4] = now_ticks() + (acc[0] == 1.2345e30f ? 1 : 0);   // (after the matrix-core chain)
    // ---- epilogue: lane (g, l15) holds features f0 + g*4 + r (r = 0..3) of token row l15
    const int m = l15;
    const bool mok = m < p.M;
    const int nf = f0 + g * 4;
    if constexpr (F8) {                                           // static input scale x per-output-channel weight scale (W row nf + r)
        const f32x4 ws = ld16<f32x4>(p.wscale + nf);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] *= p.xscale * ws[r];
    }
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
                o[r] = f2bf(rbf(silu_fast(gt)) * u);               // act_fn output (bf16), product (bf16)
            }
            const int fb = (f0 >> 6) * 32 + ((f0 & 63) >> 4) * 8 + g * 4;
            if (F8 && p.out_fp8_inv > 0.f) {                       // down_proj's input of the fp8 model: e4m3(bf16 value / input scale)
                alignas(4) unsigned short q2[2];
                q2[0] = f2fp8x2(bf2f(o[0]) * p.out_fp8_inv, bf2f(o[1]) * p.out_fp8_inv);
                q2[1] = f2fp8x2(bf2f(o[2]) * p.out_fp8_inv, bf2f(o[3]) * p.out_fp8_inv);
                *(unsigned int*)((unsigned char*)p.out + (long)m * p.ldo + fb) = *(unsigned int*)&q2[0];
            } else *(u32x2*)((bf16_t*)p.out + (long)m * p.ldo + fb) = *(u32x2*)&o[0];
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
            if ((meos && n == (p.eos_col1 > 0 ? p.eos_col1 - 1 : meos - 1)) || n >= p.n_valid) v = -INFINITY;
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
    if (p.tl) { wait_vmem(); if (w == 0) mark(5); }
}


// the split actually used for (K, requested ksplit): a K slice must fit the LDS panel and the register chunk (<= 16 k-tiles)
inline int gemv_ksplit(int K, int ksplit) {
    const int ktiles = K / 64, need = (ktiles + 15) / 16;
    if (ksplit < need) ksplit = need;
    if (ksplit > ktiles) ksplit = ktiles;
    return ksplit < 1 ? 1 : ksplit;
}

template <int EPI, bool PRO, int FW = 4, bool F8 = false>
inline void gemv_launch(GemvArgs p, int ksplit, hipStream_t s) {
    static_assert(EPI != EPI_SPLITK || FW == 4, "the split-K instantiations below are written for 4 feature waves (grid, block and the helper barrier)");
    const int ktiles = p.K / (F8 ? 128 : 64);
    ksplit = gemv_ksplit(p.K / (F8 ? 2 : 1), ksplit);             // (PRO: the panel is the whole normalised row, K = H <= 1024; fp8: 128 k-values per k-tile)
    p.k_tiles_per_split = (ktiles + ksplit - 1) / ksplit;
    const int nsplit = (ktiles + p.k_tiles_per_split - 1) / p.k_tiles_per_split;
    if constexpr (EPI == EPI_ARGMAX) p.part_stride = p.N / 16;
    const int kps = p.k_tiles_per_split;
    const dim3 grid((p.N + 16 * FW - 1) / (16 * FW), nsplit), block((FW + 4) * 64);
    if constexpr (EPI == EPI_SPLITK) {                            // the split-K GEMVs come in every slice length
        if (kps <= 2) { NTTS_LAUNCH((gemv_kernel<EPI, PRO, 2, 4, F8>), grid, block, s, p); return; }
        if (kps <= 4) { NTTS_LAUNCH((gemv_kernel<EPI, PRO, 4, 4, F8>), grid, block, s, p); return; }
        if (kps <= 8) { NTTS_LAUNCH((gemv_kernel<EPI, PRO, 8, 4, F8>), grid, block, s, p); return; }
    }
    // (K = 896 is 14 k-tiles: a 14-tile instantiation without the two surplus requests per wave -- re-reads of the wave's last tile -- measured
    //  the same, 0.9658-0.9726 ms per step either way at batch 1: the surplus requests hit in the cache; not instantiated)
    if constexpr (F8 && EPI != EPI_SPLITK) {                        // (fp8: K = H <= 1024 is at most 8 k-tiles)
        if (kps <= 8) { NTTS_LAUNCH((gemv_kernel<EPI, PRO, 8, FW, F8>), grid, block, s, p); return; }
    }
    NTTS_LAUNCH((gemv_kernel<EPI, PRO, 16, FW, F8>), grid, block, s, p);
}

// number of split-K slabs gemv_launch produces for (K, ksplit)
inline int gemv_nsplit(int K, int ksplit, bool f8 = false) { return gemm_nsplit(K, gemv_ksplit(f8 ? K / 2 : K, ksplit), f8 ? 128 : 64); }

}  // namespace ntts
