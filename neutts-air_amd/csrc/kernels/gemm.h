// gemm.h -- bf16 "NT" GEMM on the gfx950 matrix cores:  out[M,N] = X[M,K] * W[N,K]^T  (+ epilogue)
//
// Replaces every nn.Linear on the path (hf:models/qwen2/modeling_qwen2.py:46-48,206-208,233,464-465;
// hf:models/xcodec2/modeling_xcodec2.py linears and, through overlapping rows, its Conv1d's).
// Rounding contract of an nn.Linear in bf16: fp32 accumulate, bias added in fp32, ONE rounding to bf16.
//
// Structure (MI355X-first, cdna_hip_programming.md section 5):
//   * block = WM x WN waves; the wave tile is (TM*16) rows of M  x  64 rows of W (= 64 output columns).
//   * operands are SWAPPED on the matrix core: A-operand = W fragment (rows = output features),
//     B-operand = X fragment (cols = tokens).  With W rows stored in LDS "tile-major"
//     (LDS row j*16 + g*4 + r  <->  feature g*16 + j*4 + r of the wave's 64) each lane ends up owning
//     16 CONSECUTIVE output features of one token, so the epilogue stores 32 contiguous bytes per lane
//     and a wave store covers full 128-byte lines (instead of 2-byte scatters of the natural C layout).
//   * K is consumed in tiles of 64 (= one 128-byte line per row).  Tiles are brought in by LDS-DMA
//     (global_load_lds_dwordx4): 8 rows x 128 B per wave-instruction, destination lane-linear, bank
//     conflicts removed by swizzling the SOURCE chunk (c ^ ((row>>1)&7)) and applying the same
//     involution on the ds_read_b128 side (rule 21 of the guide).  NS LDS buffers in a ring, one barrier/tile:
//     the DMAs of tiles t+1 .. t+NS-1 are in flight under the MFMAs of tile t (s_waitcnt vmcnt(N) retires them
//     in order).  The skinny decode GEMMs (M = batch, <= 1 block per CU) are latency-bound and want NS = 4;
//     the big prefill / codec GEMMs hide latency with 2 co-resident blocks per CU and keep NS = 2.
//   * blockIdx -> tile mapping is XCD-aware (8 XCDs, private L2s): each XCD walks a contiguous chunk
//     of a grouped (8 m-blocks x all n-blocks) order so co-resident blocks share W and X panels in L2.
//   * split-K (gridDim.y) writes fp32 slabs that the consumer kernel reduces (no in-launch hand-off).
#pragma once
#include <ntts/dev.h>

namespace ntts {

enum GemmEpi {
    EPI_BF16 = 0,      // out bf16 [M][N] = bf16(acc + bias)
    EPI_SILU_MUL = 1,  // W rows packed gate/up (see pack_gate_up); out bf16 [M][N/2] = silu(g)*u, HF rounding
    EPI_SPLITK = 2,    // out fp32 [split][M][N] raw partial sums
    EPI_ARGMAX = 3,    // per-row (max, first index) partials over this wave's 64 columns, logits = bf16(acc)
    EPI_BF16_SILU = 4, // out bf16 = bf16(silu(acc + bias))   (codec MLP fc1; the codec reference is fp32)
    EPI_F32 = 5,       // out fp32 [M][N] = acc + bias (+ fp32 residual): codec residual stream / ISTFT head / DFT
    EPI_RESID = 6,     // out bf16 [M][N] = bf16(resid_bf16 + bf16(acc + bias)): o_proj + residual add (may be in place)
    EPI_SILU_SPLIT3 = 7  // codec, precision = high: v = silu(acc + bias) leaves as a SPLIT bf16 operand row [hi | lo | hi] of 3 N columns
                         // (hi = bf16(v), lo = bf16(v - hi); ldo = 3 N): the next GEMM's K-loop over [wh | wh | wl] sees v to ~16 mantissa bits
};

struct GemmArgs {
    const bf16_t* X;
    long ldx;
    const bf16_t* W;
    long ldw;
    int w_tile_major;    // 0: W is [N][K] row-major (ldw).  1: "tile-major": for each group of 64 rows, the 64 x 64 blocks of
                         // consecutive K tiles follow each other (block = 64 rows x 128 B = 8 KB contiguous, group = 64 x K
                         // elements contiguous): a workgroup's weight stream is ONE sequential run of HBM addresses instead
                         // of 64 row segments of 128 B, 2*K bytes apart, revisited once per K tile (DRAM pages re-opened)
    const bf16_t* bias;  // [N] or nullptr
    const float* bias_f32;  // alternative fp32 bias (codec path keeps its affine terms in fp32)
    void* out;
    long ldo;
    int M, N, K;  // K % 64 == 0
    int k_tiles_per_split;
    int mblocks, nblocks;
    // EPI_ARGMAX
    float* part_val;
    int* part_idx;
    int part_stride;       // partials per row = nblocks * WN
    const int* mask_eos;   // [M] value e+1 > 0 -> logit[e] = -inf for that row (MinNewTokens processor)
    int eos_col1;          // > 0: W is a COMPACTED head (ntts_backbone_set_logits_range: the rows of a token range + the EOS row) and the EOS row is
                           // column eos_col1 - 1 -- the mask applies there whatever id mask_eos carries; 0 (a zeroed struct): column = token id
    float* logits;         // optional fp32 [M][ld_logits] dump of the processed logits
    long ld_logits;
    bf16_t* logits_bf16;   // optional bf16 [M][ld_logits_bf16] processed logits for the top-k sampler
    long ld_logits_bf16;
    // EPI_F32
    const float* resid;    // optional fp32 [M][ldr] added in the epilogue (may alias out)
    long ldr;
    // EPI_RESID
    const bf16_t* resid_bf16;   // [M][ldrb]
    long ldrb;
    // F8 kernels (fp8 e4m3 operands): X and W hold BYTES ([M][ldx] / [N][ldw] or tile-major in 64 x 128-byte blocks), K is a
    // multiple of 128; out = acc * (xscale * wscale[n]) (+ bias): static per-tensor activation scale, per-output-channel
    // weight scale.  EPI_SILU_MUL can emit fp8 (the next GEMM's input): out byte = e4m3(value * out_fp8_inv), 0 = bf16 out.
    const float* wscale;
    float xscale;
    float out_fp8_inv;
    // split-K placement (gemm_launch sets it): 0 = splits in gridDim.y (every XCD then works on every K slice of its tiles, so
    // each of the 8 private L2s pulls the WHOLE X panel: measured 2.8x the algorithmic fetch on the skinny decode GEMMs,
    // profiles/r02e_pmc_fetch_*); n = 2 / 4 / 8: a 1-D grid in which an XCD only ever sees ONE K slice (8 / n XCDs share the
    // tiles of a slice), so X and W are each fetched into exactly one L2 per slice.  Speed only: block b runs on XCD b % 8
    // is an observation, not a contract, and nothing depends on it.
    int xcd_nsplit, xcd_per;
    // row-block placement (gemm_launch sets both from a request of xcd_maffine = -1; 1 / 2 / 4 / 8 m-blocks): the xcd_xps = 8 / mblocks XCDs
    // of group p work on m-block p alone -- all its n-blocks and K slices -- so that the 64 batch rows of an m-block are produced,
    // reduced (norm.h xcd_rows), attended (attn_decode.h xcd_rows) and consumed inside one group of L2s instead of crossing the fabric
    // at every kernel boundary; xcd_maffine = number of K slices.  Every group then streams the whole W (fetched from HBM once, from the
    // memory-side cache by the other groups).  Speed only, like xcd_nsplit: block b on XCD b % 8 is an observation, not a contract.
    int xcd_maffine, xcd_xps;
    long x_kt_stride;         // probe (tools/ubench_gemm.py --x-layout): > 0: X is stored K-TILE-MAJOR -- [K / 64][M] rows of 128 bytes (ldx = 64), this many
                              // bytes between consecutive K tiles of a row -- so that a block's X tile is one contiguous run; 0: row-major rows of ldx elements
    unsigned long long* tl;   // diagnostics (ntts_backbone_gemv_timeline on a large-batch engine): [workgroups][16] timestamps of wave 0 -- 0 entry,
                              // 1 first ring slots requested, 2 first tile landed (past the first barrier), 3 k-loop done, 4 epilogue issued, 5 stores drained
};

// position t of the grouped tile order (8 m-blocks x all n-blocks per group, m fastest) -> tile coordinates
NTTS_D void gemm_tile_from_linear(int t, int mblocks, int nblocks, int& mb, int& nb) {
    const int GM = 8;
    const int per_group = GM * nblocks;
    const int g = t / per_group;
    const int first_m = g * GM;
    const int gsz = (mblocks - first_m) < GM ? (mblocks - first_m) : GM;
    const int in = t - g * per_group;
    mb = first_m + in % gsz;
    nb = in / gsz;
}

NTTS_D void gemm_tile_coords(int bid, int mblocks, int nblocks, int& mb, int& nb) {
    const int ntiles = mblocks * nblocks;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective XCD chunking
    const int GM = 8;
    const int per_group = GM * nblocks;
    const int g = t / per_group;
    const int first_m = g * GM;
    const int gsz = (mblocks - first_m) < GM ? (mblocks - first_m) : GM;
    const int in = t - g * per_group;
    mb = first_m + in % gsz;
    nb = in / gsz;
}

NTTS_D float silu_f(float x) { return x / (1.0f + fexp(-x)); }
// x * sigmoid(x) without libm's expf (range checks) and without an IEEE division: t = exp(-|x|) by one v_exp_f32, 1 / (1 + t)
// by a refined v_rcp_f32; sigmoid(x) = 1 / (1 + t) for x >= 0 and t / (1 + t) for x < 0.  ~10 instructions instead of ~30: the
// SiLU epilogue is 10-25 % of the big gate/up and codec fc1 GEMMs (157 M / 268 M elements per launch).  |x| is clamped to 126
// (exp(-126) already underflows in the sum 1 + t, and the clamp keeps the exponent's argument finite).  On the backbone path the
// input is a bf16 value and the result is rounded to bf16: tests/test_gpu_kernels.py::test_silu_all_bf16_inputs checks EVERY
// bf16 input against torch's bf16 SiLU (hf:activations.py SiLUActivation -> torch.nn.functional.silu), so the two
// implementations are interchangeable bit for bit there; the codec's fc1 (fp32 input) is covered by its waveform tolerance.
NTTS_D float silu_fast(float x) {
    const float ax = __builtin_fminf(__builtin_fabsf(x), 126.0f);
    // exp(-|x|) is subnormal (flushed by v_exp_f32) for |x| > 87.3 while x * exp(-|x|) is still a normal number down to
    // x = -88.7: those three bf16 inputs (-87.5, -88, -88.5; results ~ -5e-37) take the division form -- a branch no
    // activation of a real model ever takes
    if (ax > 87.0f && x < 0.f) return silu_f(x);
    // exp(-|x|) as ONE v_exp_f32 of -|x| * log2(e) (round 3: fexp_neg's remainder correction, 4 more instructions per element, buys nothing
    // here -- the exhaustive test below passes with 0 differences of 65 280 either way; the unrefined v_rcp_f32 does NOT: 1 difference)
    const float t = fexp2(-ax * 1.44269504088896340736f);
    const float r = frcp_refined(1.0f + t);
    return x * (x >= 0.f ? r : t * r);
}
// Two values at once, branch-free, on the packed fp32 ALU (v_pk_mul / v_pk_add / v_pk_fma: IEEE-identical to the scalar forms, two lanes' worth
// per issue slot): the arithmetic of silu_fast operation for operation.  NOT valid for x < -87 (silu_fast's division form); callers test their
// inputs once per wave (silu2_needs_slow) and fall back to silu_fast for the whole fragment -- a branch no activation of a real model takes.
typedef __attribute__((ext_vector_type(2))) float f32x2;
NTTS_D f32x2 silu_fast2(f32x2 x) {
    f32x2 ax;
    ax[0] = __builtin_fminf(__builtin_fabsf(x[0]), 126.0f);
    ax[1] = __builtin_fminf(__builtin_fabsf(x[1]), 126.0f);
    const f32x2 e = ax * -1.44269504088896340736f;            // (-ax) * log2(e): the sign flip is exact
    f32x2 t;
    t[0] = fexp2(e[0]);
    t[1] = fexp2(e[1]);
    const f32x2 d = t + 1.0f;
    f32x2 r;
    r[0] = frcp_raw(d[0]);
    r[1] = frcp_raw(d[1]);
    r = __builtin_elementwise_fma(__builtin_elementwise_fma(-d, r, f32x2{1.0f, 1.0f}), r, r);   // frcp_refined
    const f32x2 tr = t * r;
    f32x2 sg;
    sg[0] = x[0] >= 0.f ? r[0] : tr[0];
    sg[1] = x[1] >= 0.f ? r[1] : tr[1];
    return x * sg;
}
NTTS_D float gemm_bias(const GemmArgs& p, int n) { return p.bias_f32 ? p.bias_f32[n] : (p.bias ? bf2f(p.bias[n]) : 0.f); }

// ---- epilogue shared by the GEMM kernels: lane owns token m (per a) x features nb16 .. nb16+15
//      (acc[a][j][r] <-> feature nb16 + j*4 + r); mrow0 = first row of this wave's tile, split = split-K slab index
template <int TM, int EPI, int WN, bool F8 = false, bool F16 = false>
NTTS_D void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[TM][4], int mrow0, int n0, int wn, int nb, int split) {
    const int lane = lane_id();
    const int g = lane >> 4, l15 = lane & 15;
    const int nb16 = n0 + wn * 64 + g * 16;
    float sc[4][4];        // F8: xscale * wscale[n] of this lane's 16 features (one product, then ONE multiply per accumulator)
    if constexpr (F8) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb16 + j * 4 + r;
                sc[j][r] = p.xscale * p.wscale[n < p.N ? n : p.N - 1];
            }
        if constexpr (EPI != EPI_BF16) {   // (EPI_BF16 applies the scale and the bias in one fma, below)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][j][r] *= sc[j][r];
        }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int m = mrow0 + a * 16 + l15;
        const bool mok = m < p.M;
        if constexpr (EPI == EPI_BF16 || EPI == EPI_BF16_SILU) {
            alignas(16) bf16_t o[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = nb16 + j * 4 + r;
                    float v = acc[a][j][r];
                    if constexpr (F8 && EPI == EPI_BF16) v = __builtin_fmaf(v, sc[j][r], n < p.N ? gemm_bias(p, n) : 0.f);
                    else if (n < p.N) v += gemm_bias(p, n);
                    if constexpr (EPI == EPI_BF16_SILU) v = silu_fast(v);
                    o[j * 4 + r] = f2op<F16>(v);          // (F16: the codec's fp16 operand rows)
                }
            if (mok) {
                bf16_t* dst = (bf16_t*)p.out + (long)m * p.ldo + nb16;
                if (nb16 + 16 <= p.N) {
                    *(u32x4*)dst = *(u32x4*)&o[0];
                    *(u32x4*)(dst + 8) = *(u32x4*)&o[8];
                } else {
                    for (int e = 0; e < 16; ++e)
                        if (nb16 + e < p.N) dst[e] = o[e];
                }
            }
        } else if constexpr (EPI == EPI_SILU_SPLIT3) {
            alignas(16) bf16_t hi[16], lo[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = nb16 + j * 4 + r;
                    float v = acc[a][j][r];
                    if (n < p.N) v += gemm_bias(p, n);
                    v = silu_fast(v);
                    hi[j * 4 + r] = f2bf(v);
                    lo[j * 4 + r] = f2bf(v - bf2f(hi[j * 4 + r]));
                }
            if (mok) {
                bf16_t* dst = (bf16_t*)p.out + (long)m * p.ldo + nb16;
                if (nb16 + 16 <= p.N) {
                    *(u32x4*)dst = *(u32x4*)&hi[0];
                    *(u32x4*)(dst + 8) = *(u32x4*)&hi[8];
                    *(u32x4*)(dst + p.N) = *(u32x4*)&lo[0];
                    *(u32x4*)(dst + p.N + 8) = *(u32x4*)&lo[8];
                    *(u32x4*)(dst + 2 * p.N) = *(u32x4*)&hi[0];
                    *(u32x4*)(dst + 2 * p.N + 8) = *(u32x4*)&hi[8];
                } else {
                    for (int e = 0; e < 16; ++e)
                        if (nb16 + e < p.N) { dst[e] = hi[e]; dst[p.N + e] = lo[e]; dst[2 * p.N + e] = hi[e]; }
                }
            }
        } else if constexpr (EPI == EPI_RESID) {
            // o = bf16(acc + bias) (the nn.Linear output), h = bf16(resid + o) (the residual add); in place is fine:
            // every element is read and written by the one lane that owns it
            if (mok) {
                const bf16_t* rs = p.resid_bf16 + (long)m * p.ldrb + nb16;
                bf16_t* dst = (bf16_t*)p.out + (long)m * p.ldo + nb16;
                if (nb16 + 16 <= p.N) {
                    alignas(16) bf16_t rr[16], o[16];
                    *(u32x4*)&rr[0] = *(const u32x4*)rs;
                    *(u32x4*)&rr[8] = *(const u32x4*)(rs + 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            o[j * 4 + r] = f2bf(bf2f(rr[j * 4 + r]) + rbf(acc[a][j][r] + gemm_bias(p, nb16 + j * 4 + r)));
                    *(u32x4*)dst = *(u32x4*)&o[0];
                    *(u32x4*)(dst + 8) = *(u32x4*)&o[8];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int n = nb16 + j * 4 + r;
                            if (n < p.N) dst[j * 4 + r] = f2bf(bf2f(rs[j * 4 + r]) + rbf(acc[a][j][r] + gemm_bias(p, n)));
                        }
                }
            }
        } else if constexpr (EPI == EPI_F32) {
            if (mok) {
                float* dst = (float*)p.out + (long)m * p.ldo + nb16;
                const float* rs = p.resid ? p.resid + (long)m * p.ldr + nb16 : nullptr;
                if (nb16 + 16 <= p.N) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 v = acc[a][j];
                        if (p.bias || p.bias_f32) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += gemm_bias(p, nb16 + j * 4 + r);
                        }
                        if (rs) {
                            const f32x4 q = ld16<f32x4>(rs + j * 4);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += q[r];
                        }
                        *(f32x4*)(dst + j * 4) = v;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int n = nb16 + j * 4 + r;
                            if (n < p.N) dst[j * 4 + r] = acc[a][j][r] + gemm_bias(p, n) + (rs ? rs[j * 4 + r] : 0.f);
                        }
                }
            }
        } else if constexpr (EPI == EPI_SILU_MUL) {
            // packed rows: j = 0,1 -> gate features fb + j*4 + r ; j = 2,3 -> up of the same features
            alignas(16) bf16_t o[8];
            // gate_proj / up_proj outputs (bf16), act_fn output (bf16), product (bf16): the roundings of hf:models/qwen2/modeling_qwen2.py:46-48, two
            // features per issue slot (silu_fast2): 900 -> 700 issue slots per thread of a 256 x 256 tile.  Decode step 1.600 -> 1.595 ms, prompt pass
            // unchanged (A/B in one call, profiles/r04l_ab_silu_packed.txt): the epilogue's time is its stores, not its arithmetic.
            float lowest = 0.f;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) lowest = __builtin_fminf(lowest, acc[a][jj][r]);
            if (__builtin_expect(any_lane(lowest < -86.5f), 0)) {   // (rounding to bf16 cannot carry a value above -86.5 below -87)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gt = rbf(acc[a][jj][r]);
                        const float up = rbf(acc[a][jj + 2][r]);
                        const float s = rbf(silu_fast(gt));
                        o[jj * 4 + r] = f2bf(s * up);
                    }
            } else {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 gt = rbf2(f32x2{acc[a][jj][r], acc[a][jj][r + 1]});
                        const f32x2 up = rbf2(f32x2{acc[a][jj + 2][r], acc[a][jj + 2][r + 1]});
                        const f32x2 sv = rbf2(silu_fast2(gt));
                        const f32x2 pr = sv * up;
                        o[jj * 4 + r] = f2bf(pr[0]);
                        o[jj * 4 + r + 1] = f2bf(pr[1]);
                    }
            }
            if (mok) {
                const int fb = ((n0 + wn * 64) >> 1) + g * 8;
                if (fb + 8 <= (p.N >> 1)) {
                    if (F8 && p.out_fp8_inv > 0.f) {   // the down_proj input of the fp8 model: e4m3(bf16 value / input scale)
                        alignas(8) unsigned short q[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) q[e] = f2fp8x2(bf2f(o[2 * e]) * p.out_fp8_inv, bf2f(o[2 * e + 1]) * p.out_fp8_inv);
                        *(u32x2*)((unsigned char*)p.out + (long)m * p.ldo + fb) = *(u32x2*)&q[0];
                    } else {
                        *(u32x4*)((bf16_t*)p.out + (long)m * p.ldo + fb) = *(u32x4*)&o[0];
                    }
                }
            }
        } else if constexpr (EPI == EPI_SPLITK) {
            if (mok && nb16 + 16 <= p.N) {
                float* dst = (float*)p.out + ((long)split * p.M + m) * p.ldo + nb16;
#pragma unroll
                for (int j = 0; j < 4; ++j) *(f32x4*)(dst + j * 4) = acc[a][j];
            }
        } else if constexpr (EPI == EPI_ARGMAX) {
            float best = -INFINITY;
            int bidx = 0x7fffffff;
            const int meos = (mok && p.mask_eos) ? p.mask_eos[m] : 0;   // eos id + 1, or 0
            alignas(16) bf16_t lo[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = nb16 + j * 4 + r;
                    float v = rbf(acc[a][j][r]);                  // lm_head output is bf16, then .float()
                    if (meos && n == (p.eos_col1 > 0 ? p.eos_col1 - 1 : meos - 1)) v = -INFINITY;
                    lo[j * 4 + r] = f2bf(v);
                    if (n < p.N) {
                        if (mok && p.logits) p.logits[(long)m * p.ld_logits + n] = v;
                        if (v > best) { best = v; bidx = n; }      // ascending n + strict '>' = first max wins
                    }
                }
            if (p.logits_bf16 && mok) {
                bf16_t* dst = p.logits_bf16 + (long)m * p.ld_logits_bf16 + nb16;
                if (nb16 + 16 <= p.N) {
                    *(u32x4*)dst = *(u32x4*)&lo[0];
                    *(u32x4*)(dst + 8) = *(u32x4*)&lo[8];
                } else {
                    for (int e = 0; e < 16; ++e)
                        if (nb16 + e < p.N) dst[e] = lo[e];
                }
            }
#pragma unroll
            for (int sh = 16; sh <= 32; sh <<= 1) {
                const float ov = shfl_xor(best, sh);
                const int oi = shfl_xor(bidx, sh);
                if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
            }
            if (mok && g == 0) {
                const long pi = (long)m * p.part_stride + nb * WN + wn;
                p.part_val[pi] = best;
                p.part_idx[pi] = bidx;
            }
        }
    }
}

// ---- epilogue of the natural-order tiles (TN != 4): the wave's W rows are features nw0 .. nw0 + TN*16 - 1 in order, so
//      acc[a][j][r] <-> token mrow0 + a*16 + l15, feature nw0 + j*16 + g*4 + r  (MFMA D: column = l15, rows = g*4 + r)
template <int TM, int TN, int EPI, int WN>
NTTS_D void gemm_epilogue_nat(const GemmArgs& p, f32x4 (&acc)[TM][TN], int mrow0, int nw0, int wn, int nb) {
    const int lane = lane_id();
    const int g = lane >> 4, l15 = lane & 15;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int m = mrow0 + a * 16 + l15;
        const bool mok = m < p.M;
        if constexpr (EPI == EPI_BF16) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n4 = nw0 + j * 16 + g * 4;
                alignas(8) bf16_t o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[a][j][r] + (n4 + r < p.N ? gemm_bias(p, n4 + r) : 0.f));
                if (mok) {
                    bf16_t* dst = (bf16_t*)p.out + (long)m * p.ldo + n4;
                    if (n4 + 4 <= p.N) *(u32x2*)dst = *(u32x2*)&o[0];
                    else for (int e = 0; e < 4; ++e) if (n4 + e < p.N) dst[e] = o[e];
                }
            }
        } else if constexpr (EPI == EPI_ARGMAX) {
            float best = -INFINITY;
            int bidx = 0x7fffffff;
            const int meos = (mok && p.mask_eos) ? p.mask_eos[m] : 0;   // eos id + 1, or 0
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                alignas(8) bf16_t lo[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = nw0 + j * 16 + g * 4 + r;
                    float v = rbf(acc[a][j][r]);                  // lm_head output is bf16, then .float()
                    if (meos && n == (p.eos_col1 > 0 ? p.eos_col1 - 1 : meos - 1)) v = -INFINITY;
                    lo[r] = f2bf(v);
                    if (n < p.N) {
                        if (mok && p.logits) p.logits[(long)m * p.ld_logits + n] = v;
                        if (v > best) { best = v; bidx = n; }      // ascending n + strict '>' = first max wins
                    }
                }
                if (p.logits_bf16 && mok) {
                    const int n4 = nw0 + j * 16 + g * 4;
                    bf16_t* dst = p.logits_bf16 + (long)m * p.ld_logits_bf16 + n4;
                    if (n4 + 4 <= p.N) *(u32x2*)dst = *(u32x2*)&lo[0];
                    else for (int e = 0; e < 4; ++e) if (n4 + e < p.N) dst[e] = lo[e];
                }
            }
#pragma unroll
            for (int sh = 16; sh <= 32; sh <<= 1) {
                const float ov = shfl_xor(best, sh);
                const int oi = shfl_xor(bidx, sh);
                if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
            }
            if (mok && g == 0) {
                const long pi = (long)m * p.part_stride + nb * WN + wn;
                p.part_val[pi] = best;
                p.part_idx[pi] = bidx;
            }
        } else {
            static_assert(EPI == EPI_ARGMAX || EPI == EPI_BF16, "epilogues of the natural-order tile");
        }
    }
}

// ABL (micro-benchmark ablation, always 0 in the product): 1 = no LDS reads / MFMA, 2 = no LDS-DMA, 4 = no stores
// BK = K extent of one ring slot (64 or 32).  A 256 x 256 block moves 64 KB per 64-wide K tile and one LDS-DMA round
// trip takes ~1.25 us whatever else the CU does, so a 2-slot ring of 64-wide tiles (all 128 KB a CU can spare) caps the
// load path at ~51 GB/s per CU -- below what the MFMAs of the tile need.  With BK = 32 the same 128 KB hold FOUR slots:
// three slices (96 KB) stay in flight under the MFMAs of the fourth.
// WNT: the W stream uses the non-temporal cache policy (decode step: every weight byte is read once per step by one
// workgroup, or by the few m-blocks of one XCD); X keeps the default policy (re-read by every n-block).
// F8: fp8 e4m3 operands on v_mfma_f32_16x16x32_fp8_fp8.  Everything about staging is byte-identical to the bf16 kernel -- an
// LDS row is still one 128-byte line, now 128 k-values instead of 64 -- so the ring, the DMA pieces and the swizzle are
// shared; a lane's 16-byte fragment read holds TWO 8-byte MFMA operands (the low and the high 8 of its 16 k-values; A and B
// use the same split, so the k order is consistent).  Half the weight AND activation bytes through the per-CU load path.
// F16: IEEE-half operands on v_mfma_f32_16x16x32_f16 (the codec's default: 3 more significant bits than bf16 at the same rate and the
// same bytes); the 16-bit outputs of EPI_BF16 / EPI_BF16_SILU are then halves too.  Staging is format-blind.
// TN: 16-column MFMA blocks per wave (the wave tile is TM*16 rows x TN*16 columns).  TN = 4 is the family described above
// (a lane owns 16 consecutive output features).  TN != 4 exists for the decode lm_head, whose GRID, not whose tile, was the
// problem (EPI_ARGMAX only; gemm_epilogue_nat): 850 tiles of 256 x 256 are 3.32 rounds of the 256 CUs -- the fourth round runs
// 82 tiles on an otherwise idle chip -- while 756 tiles of 256 x 288 (WN = 3, TN = 6, 12 waves) are 2.95: 129.5 -> 118.7-123.8 us.
// Not the 3 / 4 the round count promises: a tile's time follows its LDS-DMA bytes (X re-read from L2 + W from HBM come to
// 6.1 TB/s chip-wide = 24 GB/s per CU for either tile), so the 288-column tile costs 40 us where the 256-column one cost 32.
// The W rows sit in LDS in natural order (LDS row q <-> feature n0 + q), the loader may be uneven (NINST % NW != 0: the surplus
// instruction slots of the last waves are skipped, and the counted waits use each wave's own count).  The same machinery on the
// decode gate/up GEMM (128 x 80 as 244 workgroups instead of 128 x 128 as 152; 128 x 96 as 204) measured 13.7-15.5 vs 13.5 us
// and was removed again (profiles/r02k_sweep_lpt_head_gu_tiles.log), and so was the prefill gate/up GEMM on the 256 x 288 tile
// (34 column blocks instead of 38: 16.6 rounds instead of 18.55 per chunk, but the prompt pass got 12 % SLOWER, 32.4 vs 29.0 ms
// per chunk: in natural order the SiLU * up epilogue runs on half the lanes -- gate rows in lanes g < 2, their up rows in g >= 2
// -- and stores 8-byte pieces; profiles/r02k_sweep_pf_gu_nat.log).  The prefill QKV GEMM keeps it (N = 1152 = 4 x 288: 500 tiles
// instead of 625 with every fifth half empty; prompt pass -0.8 %).
template <int WM, int WN, int TM, int EPI, int NS, int ABL = 0, int BK = 64, bool WNT = false, bool F8 = false, int TN = 4, bool F16 = false>
NTTS_KERNEL(WM * WN * 64) void gemm_kernel(GemmArgs p) {
    static_assert(!F16 || (!F8 && TN == 4 && (EPI == EPI_BF16 || EPI == EPI_BF16_SILU || EPI == EPI_F32)), "fp16 operands: codec GEMMs");
    static_assert(BK == 64 || BK == 32, "ring slot K extent");
    static_assert(!F8 || BK == 64, "fp8: one ring slot = 128-byte rows");
    static_assert(TN == 4 || ((EPI == EPI_ARGMAX || EPI == EPI_BF16) && !F8 && BK == 64), "natural-order tile: lm_head / prefill QKV, bf16");
    constexpr int ESZ = F8 ? 1 : 2;            // bytes per operand element
    constexpr int CW = TN * 16;                // output columns per wave
    constexpr int BM = WM * TM * 16, BN = WN * CW, NW = WM * WN;
    constexpr int ROWS = BM + BN;              // LDS rows per buffer, BK bf16 each
    constexpr int KC = BK / 8;                 // 16-byte chunks per row
    constexpr int RPI = 64 / KC;               // rows one wave-instruction (64 lanes x 16 B = 1 KB) brings in
    static_assert(ROWS % RPI == 0, "whole loader instructions");
    constexpr int NINST = ROWS / RPI;          // wave-instructions per slot
    constexpr int SPT = 64 / BK;               // ring slots per 64-wide K tile (GemmArgs counts K in tiles of 64)
    constexpr int PER_WAVE = (NINST + NW - 1) / NW;
    constexpr bool EVEN = NINST % NW == 0;     // every wave issues PER_WAVE instructions per slot (else the last waves issue one fewer)
    static_assert(NS >= 2 && (NS - 2) * PER_WAVE <= 63, "vmcnt range");
    NTTS_SHARED bf16_t lds[NS * ROWS * BK];
    // bank-conflict-free ds_read_b128: 16 lanes read 16 consecutive rows; XOR the 16-byte chunk index with row bits so the
    // 16 accesses cover the 64 banks once (128-byte rows: bits 1..3; 64-byte rows: bits 2..3); applied on the DMA SOURCE
    // side (the destination is lane-linear) and again on the read side
    auto swz = [](int rho) { return BK == 64 ? (rho >> 1) & 7 : (rho >> 2) & 3; };

    const int lane = lane_id(), wave = wave_id();
    const int wm = wave / WN, wn = wave % WN;
    const int g = lane >> 4, l15 = lane & 15;
    int mb, nb, split = blockIdx.y;
    if (p.xcd_maffine > 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int wi = (xcd % p.xcd_xps) + p.xcd_xps * j;      // work item of m-block xcd / xps: (n-block, K slice)
        if (wi >= p.nblocks * p.xcd_maffine) return;           // padding block (block-uniform, before any barrier)
        mb = xcd / p.xcd_xps;
        nb = wi % p.nblocks;
        split = wi / p.nblocks;
    } else if (p.xcd_nsplit) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, xps = 8 / p.xcd_nsplit;   // XCDs per K slice
        split = xcd / xps;
        const int t = (xcd % xps) * p.xcd_per + j;
        if (t >= p.mblocks * p.nblocks) return;                // padding block (block-uniform, before any barrier)
        gemm_tile_from_linear(t, p.mblocks, p.nblocks, mb, nb);
    } else {
        gemm_tile_coords(blockIdx.x, p.mblocks, p.nblocks, mb, nb);
    }
    const long tlb = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 16;
    auto mark = [&](int slot) { if (p.tl && wave == 0 && lane == 0) p.tl[tlb + slot] = now_ticks(); };
    mark(0);
    const int m0 = mb * BM, n0 = nb * BN;
    const int ktiles = F8 ? p.K >> 7 : p.K >> 6;               // 128-byte K tiles
    const int kt0 = split * p.k_tiles_per_split * SPT;        // in ring slots from here on
    int nk = ktiles - split * p.k_tiles_per_split;
    if (nk > p.k_tiles_per_split) nk = p.k_tiles_per_split;
    nk *= SPT;

    // ---- loader set-up: which global row feeds each of this lane's LDS-DMA pieces
    const char* src[PER_WAVE];                          // byte addresses (an operand element is ESZ bytes)
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int inst = (EVEN || wave + i * NW < NINST) ? wave + i * NW : 0;   // (a surplus slot's address is never used)
        const int rho = inst * RPI + lane / KC;          // LDS row
        const int c = (lane % KC) ^ swz(rho);            // logical 16-byte chunk stored at physical lane % KC
        if (rho < BM) {
            int m = m0 + rho;
            if (m > p.M - 1) m = p.M - 1;
            src[i] = (const char*)p.X + (long)m * p.ldx * ESZ + c * 16;
        } else {
            const int q = rho - BM;                      // tile-major W row: q = wq*64 + j*16 + i16
            const int wq = q >> 6, j = (q >> 4) & 3, i16 = q & 15;
            int n = TN == 4 ? n0 + wq * 64 + (i16 >> 2) * 16 + j * 4 + (i16 & 3) : n0 + q;   // (TN != 4: natural order)
            if (n > p.N - 1) n = p.N - 1;
            src[i] = p.w_tile_major ? (const char*)p.W + (long)(n >> 6) * 64 * p.K * ESZ + (n & 63) * 128 + c * 16
                                    : (const char*)p.W + (long)n * p.ldw * ESZ + c * 16;
        }
    }
    const long xstep = p.x_kt_stride > 0 ? p.x_kt_stride : BK * 2;   // bytes between consecutive K tiles of a row (128; 64 for BK = 32)
    const long wstep = p.w_tile_major ? 8192 : xstep;   // tile-major: the next 64 x 128-byte block (BK = 64 only)
    auto stage = [&](int kt, int buf) {
        if constexpr (ABL & 2) return;
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int inst = wave + i * NW;
            if (!EVEN && inst >= NINST) continue;          // (wave-uniform) surplus slot of an uneven split
            const bool is_w = (inst * RPI) >= BM;          // this instruction's rows are W rows (wave-uniform)
            const char* g = src[i] + (long)(kt0 + kt) * (is_w ? wstep : xstep);
            bf16_t* l = lds + buf * (ROWS * BK) + inst * 512;
            if (WNT && is_w) glds16_nt(g, l); else glds16(g, l);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment read offsets (elements), swizzle involution on the read side
    int xoff[TM], woff[TN], xsw[TM], wsw[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int rho = wm * TM * 16 + a * 16 + l15;
        xoff[a] = rho * BK;
        xsw[a] = swz(rho);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int rho = BM + wn * CW + j * 16 + l15;
        woff[j] = rho * BK;
        wsw[j] = swz(rho);
    }
    // this wave's LDS-DMA instructions per slot (wave-uniform): the counted waits retire a wave's OWN requests
    const bool full_share = EVEN || wave + (PER_WAVE - 1) * NW < NINST;

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) stage(s, s);
    mark(1);
    int buf = 0;                  // ring slot of tile kt
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt must have landed; tiles kt+1 .. kt+NS-2 may stay in flight (none are left to wait on
        // at the tail, where the plain drain costs nothing extra)
        if (kt + NS - 2 < nk) {
            if (full_share) wait_vmem_le<(NS - 2) * PER_WAVE>(); else wait_vmem_le<(NS - 2) * (PER_WAVE - 1)>();
        } else wait_vmem();
        sync_keep_dma();  // tile kt landed for every wave; everyone is done reading the slot refilled below
        if (kt == 0) mark(2);
        if (kt + NS - 1 < nk) stage(kt + NS - 1, buf == 0 ? NS - 1 : buf - 1);
        const bf16_t* base = lds + buf * (ROWS * BK);
        buf = buf + 1 == NS ? 0 : buf + 1;
        if constexpr (ABL & 1) continue;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            const int c = ks * 4 + g;
            bf16x8 xb[TM], wa[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) xb[a] = ld16<bf16x8>(base + xoff[a] + ((c ^ xsw[a]) << 3));
#pragma unroll
            for (int j = 0; j < TN; ++j) wa[j] = ld16<bf16x8>(base + woff[j] + ((c ^ wsw[j]) << 3));
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (F8) {
                        const i64x2 w2 = __builtin_bit_cast(i64x2, wa[j]), x2 = __builtin_bit_cast(i64x2, xb[a]);
                        acc[a][j] = mfma16_fp8(w2[0], x2[0], acc[a][j]);
                        acc[a][j] = mfma16_fp8(w2[1], x2[1], acc[a][j]);
                    } else {
                        acc[a][j] = mfma16_op<F16>(wa[j], xb[a], acc[a][j]);
                    }
                }
        }
    }

    if constexpr (ABL & 4) {
        if (acc[0][0][0] != 12345.678f) return;   // keeps the accumulators live without storing
    }
    if (p.tl && wave == 0 && lane == 0) p.tl[tlb + 3] = now_ticks() + (acc[0][0][0] == 1.2345e30f ? 1 : 0);   // (after the k-loop's last MFMA)
    if constexpr (TN == 4) gemm_epilogue<TM, EPI, WN, F8, F16>(p, acc, m0 + wm * TM * 16, n0, wn, nb, split);
    else gemm_epilogue_nat<TM, TN, EPI, WN>(p, acc, m0 + wm * TM * 16, n0 + wn * CW, wn, nb);
    if (p.tl) { mark(4); wait_vmem(); mark(5); }
}


// Measured on MI355X and removed (profiles/r02i_ab_staged_epilogue.jsonl): the bf16-output epilogues of the 256 x 256 tile staged
// through the idle ring (swizzled 128 KB tile, then whole-row stores: every 128-byte line written once, in full, instead of
// 16-byte pieces 4 per line): prefill 122.2 -> 123.4 ms, A/B in one process -- the L2 merges the pieces as they are, and
// the LDS round trip with its two barriers costs more than the wider stores save.
// Measured on MI355X and removed (profiles/r02i_*): (a) an asymmetric ring for the 256 x 256 tile -- the HBM-side operand with
// three 32 KB slots (two k-tiles in flight), the L2-resident one with two, 160 KB in all: on constant-filled probe operands the
// lm_head went 136.8 -> 117.8 us and the prefill down_proj 230 -> 210 us, on the engine's random weights nothing moved (lm_head
// 132.2 vs 131.8-134.0 us, prefill 136.3 vs 137.2 ms, A/B in one process); (b) a decode GEMM with the weight slice of a feature
// wave requested into registers at kernel entry and the batch rows in an LDS ring (gemm_wreg.h in the history): bit-correct,
// gate/up 17.6 vs 13.6 us -- one CU needs 4.2 us to receive its 114 KB of HBM-cold weights however early they are requested,
// and the X ring then costs what the tile kernel's whole k-loop costs.
// Measured on MI355X and removed (profiles/r02a_*): a persistent variant of this kernel that requested the next tile's first
// stages before storing the current tile (prefill / codec GEMMs: 1116-1190 vs 1123-1188 TFLOP/s, no gain: the store tail
// is not what the counted waits were hiding), and an X-panel-resident variant with the RMSNorm fused into the QKV / gate-up
// prologues (2.33 vs 1.95 ms per decode step at batch 256: one 4-wave workgroup per CU cannot overlap its LDS-read -> MFMA
// chains).

// Measured on MI355X and removed (round 6; profiles/r06g_ubench_gemm_8phase.txt, the kernel is in the history: commit "Experiment: 256x256 GEMM on
// eight waves ..."): the 256 x 256 tile on EIGHT waves (128 x 64 per wave) in two groups one barrier apart that take turns on the matrix cores, a K tile
// computed in four phases of 16 matrix-core instructions and staged in four 16 KB units six phases ahead with counted waits that never drain the queue
// (the guide's "8-phase" schedule on this file's operand images and epilogues; bit-identical to the tile above at every size tried, 48 launches).  Its
// LDS-DMA stream alone runs 1.4x faster than this kernel's (8192^3: 535 vs 763 us with the matrix cores ablated) and its matrix cores alone slightly
// slower (545 vs 512 us), but together on random operands: 8192^3 1278-1287 vs 1206-1242 TFLOP/s, 4096^3 1171-1247 vs 1157-1197, codec fc1 965-981 vs
// 935-947, prefill down_proj 1023-1028 vs 990-1018, prefill gate/up 898-901 vs 939-952, prefill QKV 728-732 vs 776-779, lm_head (stores ablated) 100.6 vs
// 104.7 us.  Under the board's power limit the two halves do not overlap any better than before, and at K = 896 a quarter to a third of a tile's time is
// its epilogue on two waves per SIMD (gate/up: 620 -> 430 us with the stores ablated), which no main-loop schedule touches: +-5 % by shape, not adopted.

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
struct GemmShape { int BM, BN, WN; };

template <int WM, int WN, int TM, int EPI, int NS = 2, int ABL = 0, int BK = 64, bool WNT = false, bool F8 = false, int TN = 4, bool F16 = false>
inline void gemm_launch(GemmArgs p, int ksplit, hipStream_t s) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    p.mblocks = (p.M + BM - 1) / BM;
    p.nblocks = (p.N + BN - 1) / BN;
    const int ktiles = p.K / (F8 ? 128 : 64);
    if (ksplit < 1) ksplit = 1;
    if (ksplit > ktiles) ksplit = ktiles;
    p.k_tiles_per_split = (ktiles + ksplit - 1) / ksplit;
    const int nsplit = (ktiles + p.k_tiles_per_split - 1) / p.k_tiles_per_split;
    if constexpr (EPI == EPI_ARGMAX) p.part_stride = p.nblocks * WN;
    if (EPI == EPI_SPLITK && p.xcd_maffine == -1 && (p.mblocks == 1 || p.mblocks == 2 || p.mblocks == 4 || p.mblocks == 8)) {   // row-block placement requested
        p.xcd_maffine = nsplit;
        p.xcd_xps = 8 / p.mblocks;
        p.xcd_nsplit = 0;
        const int per_xcd = (p.nblocks * nsplit + p.xcd_xps - 1) / p.xcd_xps;
        NTTS_LAUNCH((gemm_kernel<WM, WN, TM, EPI, NS, ABL, BK, WNT, F8, TN, F16>), dim3(8 * per_xcd), dim3(WM * WN * 64), s, p);
        return;
    }
    p.xcd_maffine = 0;
    if (EPI == EPI_SPLITK && p.xcd_nsplit == -1 && (nsplit == 2 || nsplit == 4 || nsplit == 8)) {   // XCD-aware split-K placement requested
        p.xcd_nsplit = nsplit;
        const int xps = 8 / nsplit, tiles = p.mblocks * p.nblocks;
        p.xcd_per = (tiles + xps - 1) / xps;
        NTTS_LAUNCH((gemm_kernel<WM, WN, TM, EPI, NS, ABL, BK, WNT, F8, TN, F16>), dim3(8 * p.xcd_per), dim3(WM * WN * 64), s, p);
        return;
    }
    p.xcd_nsplit = 0;
    NTTS_LAUNCH((gemm_kernel<WM, WN, TM, EPI, NS, ABL, BK, WNT, F8, TN, F16>), dim3(p.mblocks * p.nblocks, nsplit), dim3(WM * WN * 64), s, p);
}

// tile families:  XL = 256x256 (4x4 waves = 1024 threads, 64x64 per wave, 128 KB LDS) -- big-M GEMMs (prefill, codec):
//                      half the bytes through LDS per FLOP of the L tile; measured 1.13-1.19 vs 0.89-0.96 PFLOP/s
//                      (profiles/r01_ubench_prefill.txt), the per-CU LDS-DMA fill rate being the limiter
//                 L = 128x128 (2x2 waves, 64x64 per wave)  -- medium M, lm_head
//                 S = 64x64   (4x1 waves, 16x64 per wave)  -- decode-batch skinny GEMMs (+ split-K)
#define NTTS_GEMM_XL(EPI, p, ks, s) ::ntts::gemm_launch<4, 4, 4, EPI, 2>(p, ks, s)
// big-M dispatch used by the prefill and codec paths
// by how many tiles the GEMM has (the measurements behind the thresholds: backbone.cpp gemm_large): 256 x 256 from 140 of those tiles, else
// 128 x 128 from 240 of THOSE, else the 64 x 64 skinny tile with its 4-slot ring (streaming codec passes, short batches: few rows)
#define NTTS_GEMM_BIG(EPI, p, s) do { \
    const long txl_ = (long)(((p).M + 255) / 256) * (((p).N + 255) / 256), tl_ = (long)(((p).M + 127) / 128) * (((p).N + 127) / 128); \
    if ((p).N >= 256 && txl_ >= 140) NTTS_GEMM_XL(EPI, p, 1, s); else if (tl_ >= 240) NTTS_GEMM_L(EPI, p, 1, s); else NTTS_GEMM_S(EPI, p, 1, s); } while (0)
#define NTTS_GEMM_L(EPI, p, ks, s) ::ntts::gemm_launch<2, 2, 4, EPI, 2>(p, ks, s)
// the same dispatch on fp16 operands (codec, precision = fp16)
#define NTTS_GEMM_BIG_F16(EPI, p, s) do { \
    const long txl_ = (long)(((p).M + 255) / 256) * (((p).N + 255) / 256), tl_ = (long)(((p).M + 127) / 128) * (((p).N + 127) / 128); \
    if ((p).N >= 256 && txl_ >= 140) ::ntts::gemm_launch<4, 4, 4, EPI, 2, 0, 64, false, false, 4, true>(p, 1, s); \
    else if (tl_ >= 240) ::ntts::gemm_launch<2, 2, 4, EPI, 2, 0, 64, false, false, 4, true>(p, 1, s); \
    else ::ntts::gemm_launch<4, 1, 1, EPI, 4, 0, 64, false, false, 4, true>(p, 1, s); } while (0)
#define NTTS_GEMM_S(EPI, p, ks, s) ::ntts::gemm_launch<4, 1, 1, EPI, 4>(p, ks, s)

// number of split-K slabs gemm_launch will produce for (K, ksplit); ktile = K extent of one 128-byte tile (64 bf16, 128 fp8)
inline int gemm_nsplit(int K, int ksplit, int ktile = 64) {
    const int ktiles = K / ktile;
    if (ksplit < 1) ksplit = 1;
    if (ksplit > ktiles) ksplit = ktiles;
    const int per = (ktiles + ksplit - 1) / ksplit;
    return (ktiles + per - 1) / per;
}

}  // namespace ntts
