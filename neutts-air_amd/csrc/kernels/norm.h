// norm.h -- residual add + RMSNorm (and the embedding gather), one wave64 per row.
//
// Replaces, with their bf16 rounding points:
//   embed_tokens(ids)                                   hf:models/qwen2/modeling_qwen2.py:356
//   hidden = residual + hidden                          hf:models/qwen2/modeling_qwen2.py:291,297
//   Qwen2RMSNorm: w * bf16(x32 * rsqrt(mean(x32^2)+eps)) hf:models/qwen2/modeling_qwen2.py:247-252
// and is the consumer that reduces a split-K GEMM's fp32 slabs (gemm.h EPI_SPLITK): the sum of
// the slabs (+bias) is rounded to bf16 once, exactly where the nn.Linear output would be.
//
// Memory-bound: every element is read/written once with 16-byte accesses (8 bf16 per lane per chunk).
#pragma once
#include <ntts/dev.h>

namespace ntts {

// batch row handled by workgroup `bid` of a one-row-per-workgroup grid of 64 * (8 / xps) rows when the rows of m-block p (64 rows)
// belong on the xps XCDs of group p (workgroup b runs on XCD b % 8 -- an observation, used for speed only): bijective
NTTS_HD int xcd_row(int bid, int xps) {
    const int x = bid & 7, j = bid >> 3;
    return (x / xps) * 64 + (x % xps) * (64 / xps) + j;
}

struct NormArgs {
    // --- the branch input "o" (exactly one of the three):
    const float* slabs;      // [nslab][slab_rows][H] fp32 split-K partials
    int nslab;
    long slab_rows;
    const bf16_t* o_bf16;    // [*, H] bf16 (already rounded nn.Linear output)
    const int* gather_ids;   // embedding mode: o = embed[ids[row]]  (then resid_in must be null)
    const bf16_t* embed;
    // --- residual stream
    const bf16_t* resid_in;  // [*, H] or null
    bf16_t* resid_out;       // [*, H] or null
    // --- norm
    const bf16_t* norm_w;    // [H] or null -> no normed output
    bf16_t* normed_out;      // [*, H]
    // --- row mapping (prefill tail: gather each prompt's last token into its decode-slot row)
    const int* in_rows;      // logical row r reads input row in_rows[r]   (null: r)
    const int* out_rows;     // logical row r writes output row out_rows[r] (null: r)
    int M, H;
    float eps;
    int xcd_rows;            // add_rmsnorm_row_kernel, M = 64 * mblocks: xps = 8 / mblocks (0 = off): workgroup b takes row xcd_row(b, xps), i.e. the
                             // rows of m-block p are normalised on XCD group p (whose L2s hold that m-block's split-K slabs: gemm.h xcd_maffine)
    float out_fp8_inv;       // > 0: normed_out holds e4m3 BYTES [*, H], value = bf16 result * out_fp8_inv (GEMM input of the fp8 model)
};

// One wave64 = one row.  `r` = logical row (already clamped into range), `rok` = its outputs are wanted (the clamped
// duplicates of a ragged last block keep every lane alive for the shuffles), `write_resid` = store the new residual row.
// `dst` != nullptr: the normalised row goes there (H contiguous bf16, e.g. an LDS panel: gemv.h's fused prologue) instead of
// p.normed_out.  This is THE arithmetic (and its order) of the residual add + RMSNorm on the path; every caller shares it.
// SG = slabs whose loads are in flight together (the additions always run in ascending slab order): 4 in the stand-alone
// kernels (register budget of 4 rows per workgroup), 16 = all of them for gemv.h's helper waves, whose whole job is this
// row and for whom every extra group is one more L2 round trip on the kernel's critical path.
// WIDE (gemv.h's helper waves, whose whole job is this row and for whom every memory round trip is on the kernel's critical path): the
//   slab loads of ALL the lane's chunks are requested together, branch-free (clamped addresses; a slab past nslab re-reads the last one
//   and is not added), SG slabs per group -- up to 8 slabs are ONE round trip.  The plain form walks chunk by chunk, and hipcc does not
//   hoist chunk 1's requests above chunk 0's sums (they sit behind its guards): two dependent round trips of ~1.4 us each, measured
//   with the phase timestamps of tools/gemv_timeline.py (profiles/r03g_gemv_timeline_*.txt).  Same additions in the same order.
//   (Measured and not kept: the row's "slab sum + residual" by all four helper waves with fully coalesced one-load-per-slab accesses, an
//   extra barrier, then this wave normalising from LDS -- batch 1 0.895 -> 0.890 ms, but batch 4 / 8 0.971 -> 1.011 / 1.143 -> 1.219: with
//   several rows the helpers already work side by side, one row each; profiles/r03g_ab_two_stage_prologue.txt.)
// issue_barrier (WIDE callers only): a workgroup barrier that does not wait for loads (sync_keep_dma) right after this row's requests are
//   out.  The GEMV kernels hold their feature waves at the matching barrier, so that the prologue's 36 KB go through the CU's load path
//   AHEAD of the 86-114 KB of weights instead of interleaved with them: a CU accepts HBM-cold requests at ~40 GB/s, in issue order, and
//   the row used to land when the weights did (3.7 us; 1.9 with the head start -- tools/gemv_timeline.py, profiles/r03g_*).
template <int NCH, int SG = 4, bool WIDE = false>  // NCH = 16-byte chunks per lane: H <= 512*NCH
NTTS_D void rmsnorm_row_wave(const NormArgs& p, int r, bool rok, bool write_resid, bf16_t* dst, bool issue_barrier = false) {
    const int lane = lane_id();
    const long ri = p.in_rows ? p.in_rows[r] : r;
    const long ro = p.out_rows ? p.out_rows[r] : r;
    const int nchunk = p.H >> 3;
    float v[NCH][8];
    float ss = 0.f;
    // every operand that does not depend on the slabs is requested first (clamped addresses, no branches): the residual
    // chunk and the norm weight would otherwise each cost one more round trip AFTER the slab sums
    bf16x8 rin[NCH], wv[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const long cc = (lane + 64 * i < nchunk) ? (long)(lane + 64 * i) * 8 : 0;
        if (p.resid_in) rin[i] = ld16<bf16x8>(p.resid_in + ri * p.H + cc);
        if (p.norm_w) wv[i] = ld16<bf16x8>(p.norm_w + cc);
    }
    float ow[NCH][8];          // WIDE: the rounded slab sums of every chunk
    if constexpr (WIDE) {
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) ow[i][e] = 0.f;
        if (p.slabs) {           // (wave-uniform)
            for (int s0 = 0; s0 < p.nslab; s0 += SG) {
                f32x4 a[NCH][SG], b[NCH][SG];
#pragma unroll
                for (int u = 0; u < SG; ++u) {
                    const int su = s0 + u < p.nslab ? s0 + u : p.nslab - 1;
#pragma unroll
                    for (int i = 0; i < NCH; ++i) {
                        const long cc = (lane + 64 * i < nchunk) ? (long)(lane + 64 * i) * 8 : 0;
                        const float* sp = p.slabs + ((long)su * p.slab_rows + ri) * p.H + cc;
                        a[i][u] = ld16<f32x4>(sp);
                        b[i][u] = ld16<f32x4>(sp + 4);
                    }
                }
                if (issue_barrier && s0 == 0) sync_keep_dma();          // (wave-uniform) the first group's requests are out
#pragma unroll
                for (int u = 0; u < SG; ++u) {
                    const bool use = s0 + u < p.nslab;                  // ascending slab order, as in the plain form
#pragma unroll
                    for (int i = 0; i < NCH; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ow[i][e] = use ? ow[i][e] + a[i][u][e] : ow[i][e];
                            ow[i][4 + e] = use ? ow[i][4 + e] + b[i][u][e] : ow[i][4 + e];
                        }
                }
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) ow[i][e] = rbf(ow[i][e]);
        } else if (issue_barrier) sync_keep_dma();                      // (embedding-gather / bf16-input rows: their few requests follow)
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ci = lane + 64 * i;
        const bool ok = ci < nchunk;
        const long col = (long)ci * 8;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
        if (ok) {
            if (p.slabs && WIDE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = ow[i][e];
            } else if (p.slabs) {
                // slabs summed in ascending order; loads issued four slabs at a time so they overlap
                for (int s0 = 0; s0 < p.nslab; s0 += SG) {
                    f32x4 a[SG], b[SG];
#pragma unroll
                    for (int u = 0; u < SG; ++u)
                        if (s0 + u < p.nslab) {
                            const float* sp = p.slabs + ((long)(s0 + u) * p.slab_rows + ri) * p.H + col;
                            a[u] = ld16<f32x4>(sp);
                            b[u] = ld16<f32x4>(sp + 4);
                        }
#pragma unroll
                    for (int u = 0; u < SG; ++u)
                        if (s0 + u < p.nslab) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { o[e] += a[u][e]; o[4 + e] += b[u][e]; }
                        }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rbf(o[e]);
            } else {
                const bf16_t* src = p.gather_ids ? p.embed + (long)p.gather_ids[ri] * p.H + col
                                                 : p.o_bf16 + ri * p.H + col;
                const bf16x8 t = ld16<bf16x8>(src);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = bf2f((bf16_t)t[e]);
            }
            if (p.resid_in) {
                const bf16x8 t = rin[i];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rbf(bf2f((bf16_t)t[e]) + o[e]);
            }
            if (p.resid_out && rok && write_resid) {
                bf16x8 t;
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (short)f2bf(o[e]);
                *(bf16x8*)(p.resid_out + ro * p.H + col) = t;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[i][e] = o[e];
            ss += o[e] * o[e];
        }
    }
    if (!p.norm_w) return;  // wave-uniform
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) ss += shfl_xor(ss, sh);
    const float inv = frsqrt_exact(ss / (float)p.H + p.eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ci = lane + 64 * i;
        if (ci < nchunk && rok) {
            const long col = (long)ci * 8;
            const bf16x8 w = wv[i];
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (short)f2bf(bf2f((bf16_t)w[e]) * rbf(v[i][e] * inv));
            if (p.out_fp8_inv > 0.f) {   // fp8 model: e4m3 BYTES, into the global row or (dst: gemv.h's panel of the fp8 GEMV kernels) at byte col of dst's row
                alignas(8) unsigned short q[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    q[e] = f2fp8x2(bf2f((bf16_t)t[2 * e]) * p.out_fp8_inv, bf2f((bf16_t)t[2 * e + 1]) * p.out_fp8_inv);
                *(u32x2*)(dst ? (unsigned char*)dst + col : (unsigned char*)p.normed_out + ro * p.H + col) = *(u32x2*)&q[0];
            } else {
                *(bf16x8*)(dst ? dst + col : p.normed_out + ro * p.H + col) = t;
            }
        }
    }
}

template <int NCH>
NTTS_KERNEL(256) void add_rmsnorm_kernel(NormArgs p) {
    const int row = blockIdx.x * 4 + wave_id();
    const bool rok = row < p.M;
    rmsnorm_row_wave<NCH>(p, rok ? row : p.M - 1, rok, true, nullptr);   // clamped duplicates keep every lane alive for the shuffles
}

// Decode-batch variant: ONE ROW PER WORKGROUP (2 waves, one 16-byte chunk per thread, H <= 1024), so that 256 rows occupy
// 256 CUs instead of 64 and every CU pulls a quarter of the bytes through its load path.  Same arithmetic in the same
// order as add_rmsnorm_kernel<2>: the sum of squares of lane l runs over chunk l and then chunk l + 64 element by element
// (wave 1 hands its values to wave 0 through LDS), then the 64-lane butterfly.
NTTS_D void add_rmsnorm_row_body(const NormArgs& p) {   // the whole kernel: also the second half of qkv_rope.h embed_norm_meta_kernel
    NTTS_SHARED float hand[64][8];
    NTTS_SHARED float inv_s;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int r = p.xcd_rows ? xcd_row((int)blockIdx.x, p.xcd_rows) : (int)blockIdx.x;
    const long ri = p.in_rows ? p.in_rows[r] : r;
    const long ro = p.out_rows ? p.out_rows[r] : r;
    const int nchunk = p.H >> 3;
    const int ci = tid;                           // wave 0: chunks 0..63, wave 1: chunks 64..127
    const bool ok = ci < nchunk;
    const long col = (long)ci * 8;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    bf16x8 wv = {0, 0, 0, 0, 0, 0, 0, 0}, rin = {0, 0, 0, 0, 0, 0, 0, 0};
    {   // operands that do not depend on the slabs: requested first (clamped address, no branch), see rmsnorm_row_wave
        const long cc = ok ? col : 0;
        if (p.resid_in) rin = ld16<bf16x8>(p.resid_in + ri * p.H + cc);
        if (p.norm_w) wv = ld16<bf16x8>(p.norm_w + cc);
    }
    if (ok) {
        if (p.slabs) {
            for (int s0 = 0; s0 < p.nslab; s0 += 4) {
                f32x4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (s0 + u < p.nslab) {
                        const float* sp = p.slabs + ((long)(s0 + u) * p.slab_rows + ri) * p.H + col;
                        a[u] = ld16<f32x4>(sp);
                        b[u] = ld16<f32x4>(sp + 4);
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (s0 + u < p.nslab) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o[e] += a[u][e]; o[4 + e] += b[u][e]; }
                    }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rbf(o[e]);
        } else {
            const bf16_t* src = p.gather_ids ? p.embed + (long)p.gather_ids[ri] * p.H + col : p.o_bf16 + ri * p.H + col;
            const bf16x8 t = ld16<bf16x8>(src);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = bf2f((bf16_t)t[e]);
        }
        if (p.resid_in) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rbf(bf2f((bf16_t)rin[e]) + o[e]);
        }
        if (p.resid_out) {
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (short)f2bf(o[e]);
            *(bf16x8*)(p.resid_out + ro * p.H + col) = t;
        }
    }
    if (!p.norm_w) return;  // block-uniform
    if (w == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) hand[lane][e] = o[e];
    }
    sync();
    if (w == 0) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += o[e] * o[e];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float x = hand[lane][e]; ss += x * x; }   // zeros beyond the row: + 0 is exact
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) ss += shfl_xor(ss, sh);
        if (lane == 0) inv_s = frsqrt_exact(ss / (float)p.H + p.eps);
    }
    sync();
    const float inv = inv_s;
    if (ok) {
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (short)f2bf(bf2f((bf16_t)wv[e]) * rbf(o[e] * inv));
        if (p.out_fp8_inv > 0.f) {
            alignas(8) unsigned short q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                q[e] = f2fp8x2(bf2f((bf16_t)t[2 * e]) * p.out_fp8_inv, bf2f((bf16_t)t[2 * e + 1]) * p.out_fp8_inv);
            *(u32x2*)((unsigned char*)p.normed_out + ro * p.H + col) = *(u32x2*)&q[0];
        } else {
            *(bf16x8*)(p.normed_out + ro * p.H + col) = t;
        }
    }
}

NTTS_KERNEL(128) void add_rmsnorm_row_kernel(NormArgs p) { add_rmsnorm_row_body(p); }

// dst[r][:] = src[rows[r]][:]  (bf16, 16-byte chunks; cols % 8 == 0) -- prefill's last layer: only each prompt's last position
// goes on to o_proj / the MLP / the lm_head, so its attention row and residual row are compacted first
NTTS_KERNEL(256) void gather_rows_kernel(const bf16_t* src, long ld_src, const int* rows, bf16_t* dst, long ld_dst, int cols) {
    const long r = blockIdx.x;
    const bf16_t* s = src + (long)rows[r] * ld_src;
    bf16_t* d = dst + r * ld_dst;
    for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) *(bf16x8*)(d + c) = ld16<bf16x8>(s + c);
}

inline void add_rmsnorm_launch(const NormArgs& p, hipStream_t s, bool row_per_block = false) {
    if (row_per_block && p.H > 512 && p.H <= 1024) {     // decode batch: one row per workgroup, all CUs pull
        NTTS_LAUNCH((add_rmsnorm_row_kernel), dim3(p.M), dim3(128), s, p);
        return;
    }
    const dim3 grid((p.M + 3) / 4), block(256);
    if (p.H <= 512) NTTS_LAUNCH((add_rmsnorm_kernel<1>), grid, block, s, p);
    else if (p.H <= 1024) NTTS_LAUNCH((add_rmsnorm_kernel<2>), grid, block, s, p);
    else NTTS_LAUNCH((add_rmsnorm_kernel<4>), grid, block, s, p);
}

}  // namespace ntts
