// attn_decode.h -- one decode step of GQA attention over the paged KV cache.
//
// Replaces, for q_len = 1 (hf:models/qwen2/modeling_qwen2.py):
//   eager_attention_forward :150-172  (bf16(QK^T) * scaling -> softmax fp32 -> bf16 -> PV, bf16 out)
//   the V half of DynamicCache.update hf:cache_utils.py:127-146 (the transposed V^T page; RoPE and the K append ride in the
//   QKV projection's epilogue, qkv_rope.h, which hands over rotated q heads and the v head as one bf16 row per sequence)
//
// One workgroup (4 waves) per (sequence, kv-head): the 7 query heads of a GQA group share every K/V
// byte that is read, so KV traffic is the algorithmic minimum  L * 2 * 64 * 2 B  per (seq, kv-head, layer).
// Layout in HBM (per layer):   K  [page][kv_head][32 tokens][64 d]     (a page-head is 4 KB contiguous)
//                              V^T[page][kv_head][64 d][32 token slots] (so PV's B-operand is k-contiguous); token t of
//                              a page sits at slot v_slot(t), which makes the 8 keys one lane feeds to the PV MFMA
//                              (t = 4g..4g+3 and 16+4g..16+4g+3) ONE 16-byte load
// Matrix-core mapping: S^T = K Q^T with A = K tile (16 keys x 32 d), B = Q^T (group heads padded to 16):
// the accumulator then holds 4 consecutive keys of ONE head per lane, which is already the A-operand
// shape PV needs (k-slot e<4 -> key g*4+e, e>=4 -> key 16+g*4+e-4 of the 32-key page) -- no cross-lane
// traffic between the two MFMAs; V^T is loaded with the same key permutation.
// Scores are rounded to bf16 (the eager contract) and parked in LDS, so K is streamed exactly once.
#pragma once
#include <ntts/dev.h>
#include "norm.h"

namespace ntts {

constexpr int kPage = 32;        // tokens per KV page  (== NTTS_PAGE_TOKENS)
constexpr int kAttnLMax = 2048;  // ref:neutts/neutts.py:85 max_context
// slot of token t (0..31) inside a V^T page row: [0-3,16-19 | 4-7,20-23 | 8-11,24-27 | 12-15,28-31]
NTTS_HD int v_slot(int t) { return ((t & 15) >> 2) * 8 + (t >> 4) * 4 + (t & 3); }
constexpr int kGroupMax = 8;     // query heads per kv head handled by one workgroup

struct AttnDecodeArgs {
    const bf16_t* qkv;     // [B][ld_qkv] q heads (rotated) | k heads (unused: already in their pages) | v heads, bf16, from qkv_rope.h
    long ld_qkv;
    bf16_t* out;           // [B][nh*64]
    long ld_out;
    float out_fp8_inv;     // > 0: `out` holds e4m3 BYTES, value = bf16 result * out_fp8_inv (the o_proj input of the fp8 model)
    bf16_t* kpool;         // this layer
    bf16_t* vpool;
    const int* block_table;  // [B][max_pages]
    int max_pages;
    const int* pos;        // [B] tokens already cached == position of the token being decoded
    const int* state;      // [B] 1 = running
    const bf16_t* rope_cos;  // [max_ctx][32] bf16 (cos(emb) rounded to bf16 like HF's cos.to(dtype))
    const bf16_t* rope_sin;
    int nh, nkv;
    long slab_rows;        // context-split form: rows per chunk slab of the partial outputs (AttnSplitArgs::oslabs)
    unsigned long long* tl;   // diagnostics: [B][nkv][4 waves][8] phase timestamps (now_ticks), null in the product path
    int nt_pages;             // K / V^T pages with the non-temporal load policy (kVar & 2).  Round 6, ONE 1024-row launch (2048 workgroups, 315 MB): 66.4 -> 60.4 us
                              // (4.75 -> 5.2 TB/s) and the QKV GEMM behind it 11.2 -> 10.1 us (its X / W stay in L2); kDepth 2, early V^T requests and 8-wave
                              // workgroups on top: nothing (profiles/r06b_sweep_wide_*).  At 256 rows, alone or four chains side by side: no gain (0.970 vs 0.965 ms)
    int xcd_rows;             // xps = 8 / (batch / 64), 0 = off: workgroup x takes sequence xcd_row(x, xps) (norm.h) -- the rows of m-block p on XCD group p,
                              // where the QKV GEMM left their split-K slabs and the o_proj GEMM will read their outputs (gemm.h xcd_maffine); speed only
};

// bf16 RoPE of one (x1 = x[i], x2 = x[i+32]) pair: q*cos + rotate_half(q)*sin, every op rounded
NTTS_D void rope_pair(float x1, float x2, float c, float s, float& o1, float& o2) {
    o1 = rbf(rbf(x1 * c) + rbf(-x2 * s));
    o2 = rbf(rbf(x2 * c) + rbf(x1 * s));
}

// No prologue: the rotated q heads are one bf16 row (qkv_rope.h), this step's K entry is already in its page, so the K pages are
// the kernel's first large requests.  The v row arrives as bf16 and is placed into the transposed page here (64 two-byte stores
// per workgroup, off the critical path; pass 2 takes it from LDS).  History of this kernel's prologue (sum the QKV GEMM's split-K
// slabs + bias + rounding + RoPE + KV append, ~3 us during which no workgroup streamed): DESIGN.md section 4a, git history.
// kVar & 4: the first V^T pages are requested right behind the first K pages instead of after the score pass (small batch: the
//   kernel is one chain of dependent round trips, this removes one; at batch 256 it costs 0.7 us).  kVar & 2: non-temporal page loads
//   (no gain).  Measured and not kept at batch 256: LDS-only barriers at the softmax merge (+1.1 us), deeper register rings for K
//   and V^T or V^T alone (kDepth 2 / 3: +0.2 / +1.2 us; V^T 2 / 3: +0.0 / +0.6, profiles/r03b_sweep_slab_store_policy_attn_vdepth.log),
//   block-table entries in registers (no effect), 8-wave workgroups (noise).  The K and V^T passes are HBM-bound as they are
//   (83 MB between 1 us and 14 us after entry = 6.4 TB/s); what is left is the launch gap, the first round trip, the 1.4 us merge
//   and ~2.5 us of exit skew between the 512 workgroups.
// NW = waves per workgroup: 4 at large batch; 8 with one page per wave in flight at small batch, where ONE workgroup pulls a whole
//   context through one CU's load path (batch 1, context 625: 4 waves x 2 pages 12.0 us, 8 x 2 11.1, 8 x 1 9.8, 16 x 2 12.5;
//   profiles/r03c_sweep_b1_fused_qkv_attn_waves.log).
// LMAX = longest context the instantiation can hold scores for: the score rows are most of the kernel's LDS (33 KB of 43 at 2048:
//   three workgroups per CU).  Engines created with max_context <= 1024 take the 1024 instantiation (16.6 KB of 27: the register
//   budget -- 102 -- then allows four), which matters where the grid is many rounds deep: batch 512 x 4 kv-heads = 2048 workgroups.
// DS = 4 (small batch; grid z = 4): the workgroup computes the scores and the softmax of ALL keys like DS = 1 but only 16 of the 64 output
//   dimensions -- V^T rows 16 z .. 16 z + 15, a quarter of the V^T bytes, one PV tile per page instead of four.  At batch 1 a (sequence, kv-head)
//   is then four workgroups of 100 KB each instead of one pulling 160 KB through a single CU's load path; P, and with it every output element,
//   is computed exactly as before (the other three quarters of the K stream are L2 hits).  No cross-workgroup exchange.
// HD = head_dim (round 6: 128 for Qwen3-style checkpoints, ref:neutts/neutts.py:164 dispatches through AutoModelForCausalLM): a K page row is HD values
//   (HD / 32 matrix-core k-steps per 16-key sub-tile), the V^T page has HD rows (HD / 16 PV tiles), scores = bf16(bf16(q . k) * HD^-0.5) -- for 64
//   the factor 2^-3 is exact and the second rounding a no-op; for 128 it is not (hf:models/qwen3/modeling_qwen3.py eager_attention_forward:
//   `torch.matmul(query, key.transpose(2, 3)) * scaling` rounds the product to bf16).  The HD = 64 instantiations are what they were, bit for bit.
NTTS_HD float attn_scale(int hd) { return hd == 64 ? 0.125f : hd == 128 ? 0.088388346f /* (float)(128 ** -0.5), torch's opmath scalar */ : 0.f; }
template <int kDepth, bool kTimeline = false, int kVar = 1, int NW = 4, int LMAX = kAttnLMax, int DS = 1, int HD = 64>
NTTS_KERNEL(NW * 64) void attn_decode_kernel(AttnDecodeArgs p) {
    static_assert(DS == 1 || DS == 2 || DS == 4, "output-dimension split");
    static_assert(HD == 64 || HD == 128, "head_dim");
    constexpr int NT = NW * 64;
    constexpr int KS = HD / 32;                         // matrix-core k-steps per key (32 of the HD values each)
    constexpr int NTL = (HD / 16) / DS;                 // PV tiles (16 output dimensions each) this workgroup computes
    const int nt0 = DS == 1 ? 0 : (int)blockIdx.z * NTL;   // ... starting at tile nt0
    NTTS_SHARED bf16_t sc[kGroupMax][LMAX + 16];        // rounded scores, 33 KB at 2048; +32 B/row de-aliases the LDS banks
    NTTS_SHARED bf16_t vnew[HD];
    NTTS_SHARED float wred[NW][kGroupMax];
    NTTS_SHARED float wsum[NW][kGroupMax];
    NTTS_SHARED float ored[NW][kGroupMax][HD];

    const int b = p.xcd_rows ? xcd_row((int)blockIdx.x, p.xcd_rows) : (int)blockIdx.x;
    const int kvh = blockIdx.y;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int g = lane >> 4, l15 = lane & 15;
    const int group = p.nh / p.nkv;
    const int* bt = p.block_table + (long)b * p.max_pages;
    auto mark = [&](int phase) {   // diagnostics instantiation only (ntts_backbone_attn_timeline); compiled out of the product kernel
        if constexpr (kTimeline) {
            if (lane == 0 && w < 4) p.tl[(((long)b * p.nkv + kvh) * 4 + w) * 8 + phase] = now_ticks();
        }
    };
    mark(0);
    // ---- Order of the first requests (a wave's vector loads return IN ORDER): the block-table entries of the first K pages, the
    //      slot's state and position, the q row and the v row; then the K pages.  Page indices past the context (or of a slot that
    //      turns out not to run) address some valid page of the pool and are never used: every use below is guarded by pg < npages.
    int bt0[kDepth];
#pragma unroll
    for (int j = 0; j < kDepth; ++j) bt0[j] = bt[w + NW * j < p.max_pages ? w + NW * j : 0];
    const int st = p.state[b];
    const int P = p.pos[b];
    // K fragments of a page: key row u * 16 + l15, k-step f = d values f * 32 + g * 8 .. + 7 (HD = 64: the two steps are the lane's 16 consecutive
    // values g * 16 .. + 15 split in halves -- the same pairing of A and B operand slots, the same bits)
    auto load_k_at = [&](long page, bf16x8 (&k)[2][KS]) {
        const bf16_t* kp = p.kpool + (page * p.nkv + kvh) * kPage * HD;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int f = 0; f < KS; ++f) {
                const bf16_t* kr = kp + (u * 16 + l15) * HD + (HD == 64 ? g * 16 + f * 8 : f * 32 + g * 8);
                if constexpr (kVar & 2) k[u][f] = ld16_nt<bf16x8>(kr); else k[u][f] = ld16<bf16x8>(kr);
            }
        }
    };
    auto load_v_at = [&](long page, bf16x8 (&v)[NTL]) {
        const bf16_t* vp = p.vpool + (page * p.nkv + kvh) * HD * kPage;
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            if constexpr (kVar & 2) v[nt] = ld16_nt<bf16x8>(vp + ((nt0 + nt) * 16 + l15) * kPage + g * 8);   // K/V pages: read once per step
            else v[nt] = ld16<bf16x8>(vp + ((nt0 + nt) * 16 + l15) * kPage + g * 8);
        }
    };
    auto load_k = [&](int pg, bf16x8 (&k)[2][KS]) { load_k_at(bt[pg], k); };
    auto load_v = [&](int pg, bf16x8 (&v)[NTL]) { load_v_at(bt[pg], v); };
    bf16x8 kq[kDepth][2][KS];  // register rings of kDepth pages per wave
    bf16x8 vq[kDepth][NTL];
    bf16x8 qB[KS];
    bf16_t vrow_new = 0;       // element tid of this step's v row / the page of position P (threads 0..HD-1)
    long vpage_new = 0;
    const int L = P + 1;
    const int npages = (L + kPage - 1) / kPage;
    const int last_page = npages - 1;
    {
        const bf16_t* qrow = p.qkv + (long)b * p.ld_qkv + (long)(kvh * group + (l15 < group ? l15 : 0)) * HD;
#pragma unroll
        for (int f = 0; f < KS; ++f) qB[f] = ld16<bf16x8>(qrow + (HD == 64 ? g * 16 + f * 8 : f * 32 + g * 8));
    }
    // this step's v row (bf16, from the fused QKV kernel) and the page it belongs in: requested BEFORE the K pages (a wave's loads
    // return in order), used only after pass 1 -- a wave that had to wait for them first would issue its next K page a whole memory
    // latency late and hold the other three up at the merge barrier (measured: +1 us per launch)
    if (tid < HD) vrow_new = p.qkv[(long)b * p.ld_qkv + (long)(p.nh + p.nkv + kvh) * HD + tid];
    if (st != 1) return;  // block-uniform
    mark(1);
    if (tid < HD) vpage_new = bt[P / kPage];
#pragma unroll
    for (int j = 0; j < kDepth; ++j) load_k_at(bt0[j], kq[j]);
    if constexpr (kVar & 4) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) load_v_at(bt0[j], vq[j]);
    }
    if (l15 >= group) {   // heads beyond the GQA group: zero columns of Q^T
#pragma unroll
        for (int f = 0; f < KS; ++f)
#pragma unroll
            for (int e = 0; e < 8; ++e) qB[f][e] = 0;
    }
    mark(2);
    // ---- pass 1: S^T = K Q^T per 16-key sub-tile, bf16-rounded scores -> LDS, with the softmax statistics carried
    //      online per lane (running max and sum of exp in fp32), so that one merge after the pass yields the row max
    //      and denominator: no separate pass over the stored scores.  Masked keys use a large finite score.
    constexpr float kMasked = -1.0e30f;
    float lmax = kMasked, lsum = 0.f;
    for (int pg0 = w; pg0 < npages; pg0 += NW * kDepth) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) {
            const int pg = pg0 + NW * j;
            if (pg < npages) {
                bf16x8 kc[2][KS];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int f = 0; f < KS; ++f) kc[u][f] = kq[j][u][f];
                if (pg + NW * kDepth < npages) load_k(pg + NW * kDepth, kq[j]);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int f = 0; f < KS; ++f) a = mfma16(kc[u][f], qB[f], a);
                    const int key0 = pg * kPage + u * 16 + g * 4;
                    bf16x4 sv;
                    float s4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float s = rbf(a[r]) * attn_scale(HD);   // matmul out (bf16) * scaling (bf16): the x 2^-3 of HD = 64 is exact,
                        if constexpr (HD != 64) s = rbf(s);     // the x 128^-0.5 rounds
                        if (key0 + r >= L) s = kMasked;
                        s4[r] = s;
                        sv[r] = (short)f2bf(s);
                    }
                    if (l15 < kGroupMax) *(bf16x4*)&sc[l15][key0] = sv;
                    const float mn = fmaxf(lmax, fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3])));
                    lsum = lsum * fexp_neg(lmax - mn) + fexp_neg(s4[0] - mn) + fexp_neg(s4[1] - mn) + fexp_neg(s4[2] - mn) +
                           fexp_neg(s4[3] - mn);
                    lmax = mn;
                }
                if (pg == w) mark(3);   // this wave's first page is through the matrix core
            }
        }
    }
    mark(4);
    if (tid < HD) {   // v row -> its slot of the transposed page, and LDS (pass 2 reads it there, behind the merge barrier)
        vnew[tid] = vrow_new;
        if (nt0 == 0)           // (DS = 4: one of the four workgroups appends; the others take the row from their own LDS copy like this one)
            p.vpool[(vpage_new * p.nkv + kvh) * HD * kPage + (long)tid * kPage + v_slot(P % kPage)] = vrow_new;
    }
    // ---- V^T pages are independent of the scores: (kVar & 4: already requested next to the K pages) else get the first
    //      ones in flight under the softmax reductions
    if constexpr (!(kVar & 4)) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j)
            if (w + NW * j < npages) load_v(w + NW * j, vq[j]);
    }

    // ---- merge the (max, sum) pairs: across the 4 key groups of a wave, then across the 4 waves (one barrier)
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {
        const float om = shfl_xor(lmax, sh), os = shfl_xor(lsum, sh);
        const float mn = fmaxf(lmax, om);
        lsum = lsum * fexp_neg(lmax - mn) + os * fexp_neg(om - mn);
        lmax = mn;
    }
    if (g == 0 && l15 < kGroupMax) { wred[w][l15] = lmax; wsum[w][l15] = lsum; }
    sync();
    float m_l = kMasked, sum_l = 1.f;
    if (l15 < kGroupMax) {
        m_l = wred[0][l15];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) m_l = fmaxf(m_l, wred[ww][l15]);
        sum_l = wsum[0][l15] * fexp_neg(wred[0][l15] - m_l);            // waves in ascending order (fp32 sum order is part of the result)
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) sum_l += wsum[ww][l15] * fexp_neg(wred[ww][l15] - m_l);
    }
    const float rs_l = frcp_refined(sum_l);
    mark(5);

    // ---- pass 2: O = P V with P = bf16(exp(s - m) / sum)
    f32x4 oacc[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) oacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int pg0 = w; pg0 < npages; pg0 += NW * kDepth) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) {
            const int pg = pg0 + NW * j;
            if (pg < npages) {
                bf16x8 vc[NTL];
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) vc[nt] = vq[j][nt];
                if (pg + NW * kDepth < npages) load_v(pg + NW * kDepth, vq[j]);
                bf16x8 pA;
#pragma unroll
                for (int e = 0; e < 8; ++e) pA[e] = 0;
                if (l15 < kGroupMax) {
                    const bf16x4 s0 = *(const bf16x4*)&sc[l15][pg * kPage + g * 4];
                    const bf16x4 s1 = *(const bf16x4*)&sc[l15][pg * kPage + 16 + g * 4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pA[e] = (short)f2bf(fdiv_r(fexp_neg(bf2f((bf16_t)s0[e]) - m_l), sum_l, rs_l));
                        pA[4 + e] = (short)f2bf(fdiv_r(fexp_neg(bf2f((bf16_t)s1[e]) - m_l), sum_l, rs_l));
                    }
                }
                if (pg == last_page) {  // new token's V from LDS; nothing beyond it may leak in (0 * garbage)
#pragma unroll
                    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int key = pg * kPage + (e < 4 ? g * 4 + e : 16 + g * 4 + e - 4);
                            short val = vc[nt][e];
                            if (key == P) val = (short)vnew[(nt0 + nt) * 16 + l15];
                            if (key > P) val = 0;
                            vc[nt][e] = val;
                        }
                }
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) oacc[nt] = mfma16(pA, vc[nt], oacc[nt]);
            }
        }
    }
    mark(6);
    // D: col = d (l15 within tile nt), row = head g*4 + r
    if (g < 2) {
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ored[w][g * 4 + r][(nt0 + nt) * 16 + l15] = oacc[nt][r];
    }
    sync();
    for (int t = tid; t < group * 16 * NTL; t += NT) {
        const int hh = t / (16 * NTL), d = nt0 * 16 + t % (16 * NTL);
        float o = ored[0][hh][d];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) o += ored[ww][hh][d];            // ascending wave order
        if (p.out_fp8_inv > 0.f) ((unsigned char*)p.out)[(long)b * p.ld_out + (kvh * group + hh) * HD + d] = f2fp8c(rbf(o) * p.out_fp8_inv);
        else p.out[(long)b * p.ld_out + (kvh * group + hh) * HD + d] = f2bf(o);
    }
    mark(7);
}

// ------------------------------------------------------------------------------------------------
// Context-split ("split-L") decode attention for SMALL batches: at batch 1 the
// kernel above is two workgroups, each pulling a whole context through one CU.  Here a (sequence, kv-head) is spread over
// NSPLIT workgroups, each owning a contiguous range of KV pages.  The eager contract rounds P = bf16(exp(s - m) / sum) with
// the GLOBAL row maximum and denominator, so the work is two launches with the softmax statistics in between:
//   attn_split_scores_kernel : prologue as above (slab reduce, RoPE, KV append by chunk 0), bf16 scores of the chunk's pages to a
//                              global scratch [B][nkv][8 heads][L], (max, sum of exp) per head and chunk
//   attn_split_pv_kernel     : merges the chunks' statistics in chunk order (every workgroup does, identically), P of its pages,
//                              P V into an fp32 partial output slab [chunk][B][nh * 64]
// and the o_proj GEMV's helper waves sum the chunk slabs in order and round ONCE to bf16 (gemv.h xslabs) -- the attention
// output's rounding point.  Same arithmetic per element as the single-workgroup kernel; the fp32 additions of the softmax
// denominator and of the output run in another order (chunk partials), which is the freedom the parity bars already allow.
struct AttnSplitArgs {
    AttnDecodeArgs a;        // a.out unused
    bf16_t* scores;          // [B * nkv][kGroupMax][ld_scores]
    long ld_scores;
    float* stats;            // [B * nkv][nsplit][kGroupMax][2]  (max, sum of exp)
    float* oslabs;           // [nsplit][B][nh * 64] fp32 partial outputs
    int nsplit;
};

NTTS_KERNEL(256) void attn_split_scores_kernel(AttnSplitArgs q) {
    constexpr int NW = 4;
    const AttnDecodeArgs& p = q.a;
    NTTS_SHARED float wred[NW][kGroupMax];
    NTTS_SHARED float wsum[NW][kGroupMax];
    const int b = blockIdx.x, kvh = blockIdx.y, ch = blockIdx.z;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int g = lane >> 4, l15 = lane & 15;
    const int group = p.nh / p.nkv;
    const int* bt = p.block_table + (long)b * p.max_pages;
    const int st = p.state[b];
    const int P = p.pos[b];
    const int L = P + 1;
    const int npages = (L + kPage - 1) / kPage;
    const int ppc = (npages + q.nsplit - 1) / q.nsplit;          // pages per chunk
    const int pg_lo = ch * ppc, pg_hi = (pg_lo + ppc < npages) ? pg_lo + ppc : npages;
    // rotated q rows + an appended K entry from the fused QKV kernel (qkv_rope.h)
    bf16x8 qB[2];
    const bf16_t* qrow = p.qkv + (long)b * p.ld_qkv + (long)(kvh * group + (l15 < group ? l15 : 0)) * 64 + g * 16;
    qB[0] = ld16<bf16x8>(qrow);
    qB[1] = ld16<bf16x8>(qrow + 8);
    if (st != 1) return;  // block-uniform
    if (l15 >= group) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { qB[0][e] = 0; qB[1][e] = 0; }
    }
    if (ch == 0 && tid < 64) {   // chunk 0 places this step's v row into its slot of the transposed page (read by the PV launch)
        const bf16_t v = p.qkv[(long)b * p.ld_qkv + (long)(p.nh + p.nkv + kvh) * 64 + tid];
        const long new_page = bt[P / kPage];
        p.vpool[(new_page * p.nkv + kvh) * 64 * kPage + (long)tid * kPage + v_slot(P % kPage)] = v;
    }
    constexpr float kMasked = -1.0e30f;
    float lmax = kMasked, lsum = 0.f;
    bf16_t* srow = q.scores + ((long)(b * p.nkv + kvh) * kGroupMax + (l15 < kGroupMax ? l15 : 0)) * q.ld_scores;
    for (int pg = pg_lo + w; pg < pg_hi; pg += NW) {
        const bf16_t* kp = p.kpool + ((long)bt[pg] * p.nkv + kvh) * kPage * 64;
        bf16x8 kc[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16_t* kr = kp + (u * 16 + l15) * 64 + g * 16;
            kc[u][0] = ld16<bf16x8>(kr);
            kc[u][1] = ld16<bf16x8>(kr + 8);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = mfma16(kc[u][0], qB[0], a);
            a = mfma16(kc[u][1], qB[1], a);
            const int key0 = pg * kPage + u * 16 + g * 4;
            bf16x4 sv;
            float s4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sc_ = rbf(a[r]) * 0.125f;
                if (key0 + r >= L) sc_ = kMasked;
                s4[r] = sc_;
                sv[r] = (short)f2bf(sc_);
            }
            if (l15 < kGroupMax) *(bf16x4*)(srow + key0) = sv;
            const float mn = fmaxf(lmax, fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3])));
            lsum = lsum * fexp_neg(lmax - mn) + fexp_neg(s4[0] - mn) + fexp_neg(s4[1] - mn) + fexp_neg(s4[2] - mn) + fexp_neg(s4[3] - mn);
            lmax = mn;
        }
    }
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {
        const float om = shfl_xor(lmax, sh), os = shfl_xor(lsum, sh);
        const float mn = fmaxf(lmax, om);
        lsum = lsum * fexp_neg(lmax - mn) + os * fexp_neg(om - mn);
        lmax = mn;
    }
    if (g == 0 && l15 < kGroupMax) { wred[w][l15] = lmax; wsum[w][l15] = lsum; }
    sync();
    if (w == 0 && g == 0 && l15 < kGroupMax) {
        float m_l = wred[0][l15];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) m_l = fmaxf(m_l, wred[ww][l15]);
        float sum_l = wsum[0][l15] * fexp_neg(wred[0][l15] - m_l);
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) sum_l += wsum[ww][l15] * fexp_neg(wred[ww][l15] - m_l);
        float* sd = q.stats + (((long)(b * p.nkv + kvh) * q.nsplit + ch) * kGroupMax + l15) * 2;
        sd[0] = m_l;
        sd[1] = sum_l;
    }
}

NTTS_KERNEL(256) void attn_split_pv_kernel(AttnSplitArgs q) {
    constexpr int NW = 4, NT = 256;
    const AttnDecodeArgs& p = q.a;
    NTTS_SHARED float ored[NW][kGroupMax][64];
    const int b = blockIdx.x, kvh = blockIdx.y, ch = blockIdx.z;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int g = lane >> 4, l15 = lane & 15;
    const int group = p.nh / p.nkv;
    const int* bt = p.block_table + (long)b * p.max_pages;
    if (p.state[b] != 1) return;  // block-uniform
    const int P = p.pos[b];
    const int L = P + 1;
    const int npages = (L + kPage - 1) / kPage;
    const int ppc = (npages + q.nsplit - 1) / q.nsplit;
    const int pg_lo = ch * ppc, pg_hi = (pg_lo + ppc < npages) ? pg_lo + ppc : npages;
    // ---- global softmax statistics: the chunks' (max, sum) merged in chunk order (chunks past the context wrote nothing: skipped)
    constexpr float kMasked = -1.0e30f;
    float m_l = kMasked, sum_l = 1.f;
    if (l15 < kGroupMax) {
        const float* sd = q.stats + ((long)(b * p.nkv + kvh) * q.nsplit * kGroupMax + l15) * 2;
        const int nch = (npages + ppc - 1) / ppc;                   // chunks that own pages
        for (int c = 0; c < nch; ++c) m_l = fmaxf(m_l, sd[(long)c * kGroupMax * 2]);
        sum_l = 0.f;
        for (int c = 0; c < nch; ++c) sum_l += sd[(long)c * kGroupMax * 2 + 1] * fexp_neg(sd[(long)c * kGroupMax * 2] - m_l);
    }
    const float rs_l = frcp_refined(sum_l);
    f32x4 oacc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) oacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* srow = q.scores + ((long)(b * p.nkv + kvh) * kGroupMax + (l15 < kGroupMax ? l15 : 0)) * q.ld_scores;
    for (int pg = pg_lo + w; pg < pg_hi; pg += NW) {
        const bf16_t* vp = p.vpool + ((long)bt[pg] * p.nkv + kvh) * 64 * kPage;
        bf16x8 vc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) vc[nt] = ld16<bf16x8>(vp + (nt * 16 + l15) * kPage + g * 8);
        bf16x8 pA;
#pragma unroll
        for (int e = 0; e < 8; ++e) pA[e] = 0;
        if (l15 < kGroupMax) {
            const bf16x4 s0 = *(const bf16x4*)(srow + pg * kPage + g * 4);
            const bf16x4 s1 = *(const bf16x4*)(srow + pg * kPage + 16 + g * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pA[e] = (short)f2bf(fdiv_r(fexp_neg(bf2f((bf16_t)s0[e]) - m_l), sum_l, rs_l));
                pA[4 + e] = (short)f2bf(fdiv_r(fexp_neg(bf2f((bf16_t)s1[e]) - m_l), sum_l, rs_l));
            }
        }
        if (pg == npages - 1) {   // the last page: nothing beyond position P may leak in (0 * garbage); P itself was appended by the scores kernel
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int key = pg * kPage + (e < 4 ? g * 4 + e : 16 + g * 4 + e - 4);
                    if (key > P) vc[nt][e] = 0;
                }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) oacc[nt] = mfma16(pA, vc[nt], oacc[nt]);
    }
    if (g < 2) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ored[w][g * 4 + r][nt * 16 + l15] = oacc[nt][r];
    }
    sync();
    float* os = q.oslabs + ((long)ch * p.slab_rows + b) * p.ld_out;
    for (int t = tid; t < group * 64; t += NT) {
        const int hh = t >> 6, d = t & 63;
        float o = ored[0][hh][d];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) o += ored[ww][hh][d];
        os[(kvh * group + hh) * 64 + d] = o;
    }
}

// the tile path's o_proj reads bf16 rows (LDS-DMA): there the chunk slabs are summed (in order, one rounding) by this pass
NTTS_KERNEL(256) void attn_split_combine_kernel(AttnSplitArgs q) {
    const AttnDecodeArgs& p = q.a;
    const int b = blockIdx.x;
    if (p.state[b] != 1) return;
    for (int c8 = threadIdx.x; c8 * 8 < p.ld_out; c8 += 256) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int ch = 0; ch < q.nsplit; ++ch) {
            const float* src = q.oslabs + ((long)ch * p.slab_rows + b) * p.ld_out + c8 * 8;
            const f32x4 u0 = ld16<f32x4>(src), u1 = ld16<f32x4>(src + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] += u0[e]; acc[4 + e] += u1[e]; }
        }
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (short)f2bf(acc[e]);
        *(bf16x8*)(p.out + (long)b * p.ld_out + c8 * 8) = t;
    }
}

inline void attn_split_launch(const AttnSplitArgs& q, int batch, hipStream_t s, bool combine = false) {
    const dim3 grid(batch, q.a.nkv, q.nsplit), block(256);
    NTTS_LAUNCH((attn_split_scores_kernel), grid, block, s, q);
    NTTS_LAUNCH((attn_split_pv_kernel), grid, block, s, q);
    if (combine) NTTS_LAUNCH((attn_split_combine_kernel), dim3(batch), block, s, q);
}

// large batch (tile path): one KV page per wave in flight, V^T pages requested after the score pass; engines with
// max_context <= 1024 take the instantiation with half the score rows
inline void attn_decode_launch(const AttnDecodeArgs& p, int batch, hipStream_t s, int max_ctx) {
    const dim3 grid(batch, p.nkv), block(256);
    // Few sequences (at most 256 workgroups after the split: batch 9 .. 32 with 2 kv-heads): the small-batch form -- 8 waves, V^T requested next to K,
    // four workgroups per (sequence, kv-head) with 16 output dimensions each (DS = 4).  Batch 16 / 32: 11.5 / 11.8 -> 8.1 / 8.4 us per launch, step
    // 1.212 / 1.259 -> 1.129 / 1.177 ms; from 512 workgroups on (batch 64) it is equal, at batch 128 twice as slow (one sweep, one box).
    if (!p.tl && batch * p.nkv * 4 <= 256) {
        NTTS_LAUNCH((attn_decode_kernel<1, false, 5, 8, kAttnLMax, 4>), dim3(batch, p.nkv, 4), dim3(512), s, p);
        return;
    }
    if (!p.tl && batch * p.nkv * 2 <= 256) {           // up to 256 workgroups with TWO per (sequence, kv-head), 32 dimensions each: batch 40 / 48 / 64
        NTTS_LAUNCH((attn_decode_kernel<1, false, 5, 8, kAttnLMax, 2>), dim3(batch, p.nkv, 2), dim3(512), s, p);   // step 1.263 / 1.277 / 1.304 -> 1.191 / 1.203 / 1.236 ms;
        return;                                                                                                 // batch 96 (384 workgroups): no gain, not used
    }
    if (!p.tl && batch * p.nkv <= 256) {               // at most one workgroup per CU: 8 waves (batch 96 / 128: 13.1 / 14.0 -> 10.3 / 11.6 us, step 1.322 / 1.384 ->
        NTTS_LAUNCH((attn_decode_kernel<1, false, 1, 8, kAttnLMax, 1>), grid, dim3(512), s, p);   // 1.269 / 1.345 ms; batch 192 / 256, two per CU: equal or slower)
        return;
    }
    if (p.tl) NTTS_LAUNCH((attn_decode_kernel<1, true, 1, 4, kAttnLMax>), grid, block, s, p);   // diagnostics: phase timestamps
    else if (max_ctx <= 1024 && p.nt_pages) NTTS_LAUNCH((attn_decode_kernel<1, false, 3, 4, 1024>), grid, block, s, p);
    else if (p.nt_pages) NTTS_LAUNCH((attn_decode_kernel<1, false, 3, 4, kAttnLMax>), grid, block, s, p);
    else if (max_ctx <= 1024) NTTS_LAUNCH((attn_decode_kernel<1, false, 1, 4, 1024>), grid, block, s, p);
    else NTTS_LAUNCH((attn_decode_kernel<1, false, 1, 4, kAttnLMax>), grid, block, s, p);
}
// head_dim 128 (round 6; the generic attention geometry): one instantiation per score-row length, 4 waves, one page per wave in flight
inline void attn_decode_launch_hd128(const AttnDecodeArgs& p, int batch, hipStream_t s, int max_ctx) {
    const dim3 grid(batch, p.nkv), block(256);
    if (max_ctx <= 1024) NTTS_LAUNCH((attn_decode_kernel<1, false, 1, 4, 1024, 1, 128>), grid, block, s, p);
    else NTTS_LAUNCH((attn_decode_kernel<1, false, 1, 4, kAttnLMax, 1, 128>), grid, block, s, p);
}
// small batch (gemv.h path): 8 waves, one page each in flight, V^T requested next to K (kVar 5): at batch 1 the kernel is one
// chain of dependent round trips through ONE CU's load path
inline void attn_decode_launch_small(const AttnDecodeArgs& p, int batch, hipStream_t s) {
    const dim3 grid(batch, p.nkv);
    if (p.tl) NTTS_LAUNCH((attn_decode_kernel<1, true, 5, 8>), grid, dim3(512), s, p);   // diagnostics: phase timestamps (waves 0..3)
    else NTTS_LAUNCH((attn_decode_kernel<1, false, 5, 8, kAttnLMax, 4>), dim3(batch, p.nkv, 4), dim3(512), s, p);   // four workgroups per (sequence, kv-head): 16 output dimensions each
}

}  // namespace ntts
