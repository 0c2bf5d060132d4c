// attn_decode.h -- one decode step of GQA attention over the paged KV cache, RoPE + cache append fused.
//
// Replaces, for q_len = 1 (hf:models/qwen2/modeling_qwen2.py):
//   apply_rotary_pos_emb :113-135   DynamicCache.update hf:cache_utils.py:127-146
//   eager_attention_forward :150-172  (bf16(QK^T) * scaling -> softmax fp32 -> bf16 -> PV, bf16 out)
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
constexpr int kAttnMaxSlabs = 4;      // split-K factor of the QKV GEMM this kernel can reduce in its prologue
constexpr int kAttnDepthDefault = 1;  // KV pages each wave keeps in flight (register ring)

struct AttnDecodeArgs {
    const bf16_t* qkv;     // kPre (tile path): [B][ld_qkv] q heads (rotated) | k heads (unused) | v heads, bf16, from qkv_rope.h
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
    // small-batch path (!kPre): the QKV GEMV's fp32 split-K slabs [nslab <= kAttnMaxSlabs][slab_rows][ld_qkv]; the kernel sums
    // them in slab order, adds the bias and applies the nn.Linear output rounding (one RNE to bf16), RoPE and the KV append itself
    const float* qkv_slabs;
    int nslab;
    long slab_rows;
    const bf16_t* qkv_bias;
    unsigned long long* tl;   // diagnostics: [B][nkv][4 waves][8] phase timestamps (now_ticks), null in the product path
    int xcd_rows;             // xps = 8 / (batch / 64), 0 = off: workgroup x takes sequence xcd_row(x, xps) (norm.h) -- the rows of m-block p on XCD group p,
                              // where the QKV GEMM left their split-K slabs and the o_proj GEMM will read their outputs (gemm.h xcd_maffine); speed only
};

// bf16 RoPE of one (x1 = x[i], x2 = x[i+32]) pair: q*cos + rotate_half(q)*sin, every op rounded
NTTS_D void rope_pair(float x1, float x2, float c, float s, float& o1, float& o2) {
    o1 = rbf(rbf(x1 * c) + rbf(-x2 * s));
    o2 = rbf(rbf(x2 * c) + rbf(x1 * s));
}

// kVar: 1 = the prologue's own operands (q/k/v row, RoPE row) are requested BEFORE the first K pages, 0 = after them.
//   A wave's vector loads return in order, so whatever is requested first is what the RoPE prologue ends up waiting for;
//   measured on MI355X at batch 256 (profiles/r01e_sweep_attn_variants.jsonl): 1 = -0.6 us per launch, -1.3 % per step.
//   Also measured there and NOT kept: LDS-only barriers (s_waitcnt lgkmcnt(0) + s_barrier, leaving the K / V^T prefetch in
//   flight) after the prologue (+3.5 us per launch) or at the softmax merge (+1.1 us), scheduling fences around the K
//   requests (no effect), deeper register rings (kDepth 2 / 3: +0.2 / +1.2 us), the RoPE row served from a per-slot copy
//   so that it does not hang off the load of the position (no effect).  Round 2: the wave's block-table entries held in
//   registers (one coalesced load + lane broadcasts instead of a dependent global load per page): 20.5-21.1 vs 20.8-21.0 us,
//   step 1.683 vs 1.685 ms (profiles/r02i_sweep_attn_bt_in_registers.log) -- the K and V^T passes are HBM-bound as they are
//   (83 MB between 1 us and 14 us after entry = 6.4 TB/s), what is left is the launch gap, the first microsecond and the exit skew.
//   8-wave workgroups at batch 256: 20.2-20.4 vs 20.6-20.9 us isolated, step 1.657-1.659 vs 1.660-1.661 ms (r02i_sweep_attn_8waves.log):
//   inside the noise, not instantiated.
// NW = waves per workgroup: 4.  16-wave workgroups (a whole 600-token context's pages requested at once by the one workgroup
//   a (sequence, kv-head) gets at batch 1) were measured and are not instantiated: 13.1-13.7 vs 10.5-12.0 us per launch --
//   the 12 extra waves' 192 KB of page requests queue on the CU's ~50 GB/s load path ahead of the prologue's RoPE row
//   (prologue 1.9 -> 6.4 us); letting them request only after the prologue moves the wait into the softmax merge
//   (14.8 us).  profiles/r02c_attn_timeline_b1.txt, r02f_sweep_b1_nw16_late_fw2.log.
// LMAX = longest context the instantiation can hold scores for: the score rows are most of the kernel's LDS (33 KB of 43 at 2048:
//   three workgroups per CU).  Engines created with max_context <= 1024 take the 1024 instantiation (16.6 KB of 27: the register
//   budget -- 102 -- then allows four), which matters where the grid is many rounds deep: batch 512 x 4 kv-heads = 2048 workgroups.
// kPre: the q heads arrive rotated and this step's K entry is already in its page -- the fused QKV kernel of the tile path did both
//   (qkv_rope.h): no prologue, the K pages are the kernel's first large requests.  The v row arrives as bf16 and is placed into the
//   transposed page here (64 two-byte stores per workgroup, off the critical path; pass 2 takes it from LDS).
template <int kDepth, bool kTimeline = false, int kVar = 1, int NW = 4, int LMAX = kAttnLMax, bool kPre = false>
NTTS_KERNEL(NW * 64) void attn_decode_kernel(AttnDecodeArgs p) {
    constexpr int NT = NW * 64;
    NTTS_SHARED bf16_t sc[kGroupMax][LMAX + 16];        // rounded scores, 33 KB at 2048; +32 B/row de-aliases the LDS banks
    NTTS_SHARED bf16_t qs[16][64];
    NTTS_SHARED bf16_t knew[64];
    NTTS_SHARED bf16_t vnew[64];
    NTTS_SHARED float wred[NW][kGroupMax];
    NTTS_SHARED float wsum[NW][kGroupMax];
    NTTS_SHARED float ored[NW][kGroupMax][64];

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
    // ---- Order of the first requests.  A wave's vector loads return IN ORDER: whatever is requested before the prologue's
    //      own operands sits on the prologue's critical path.  So (kVar & 1): (1) the block-table entries of the first K
    //      pages and this token's q/k/v values (needing nothing but the slot index) go first; (2) once the position is known,
    //      its RoPE row; (3) THEN the K pages, landing while the prologue computes.  Page indices past the context (or of a
    //      slot that turns out not to run) address some valid page of the pool and are never used: every use below is
    //      guarded by pg < npages.
    int bt0[kDepth];
#pragma unroll
    for (int j = 0; j < kDepth; ++j) bt0[j] = bt[w + NW * j < p.max_pages ? w + NW * j : 0];
    const int st = p.state[b];
    const int P = p.pos[b];
    auto load_k_at = [&](long page, bf16x8 (&k)[2][2]) {
        const bf16_t* kp = p.kpool + (page * p.nkv + kvh) * kPage * 64;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16_t* kr = kp + (u * 16 + l15) * 64 + g * 16;
            if constexpr (kVar & 2) { k[u][0] = ld16_nt<bf16x8>(kr); k[u][1] = ld16_nt<bf16x8>(kr + 8); }
            else { k[u][0] = ld16<bf16x8>(kr); k[u][1] = ld16<bf16x8>(kr + 8); }
        }
    };
    auto load_v_at = [&](long page, bf16x8 (&v)[4]) {
        const bf16_t* vp = p.vpool + (page * p.nkv + kvh) * 64 * kPage;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if constexpr (kVar & 2) v[nt] = ld16_nt<bf16x8>(vp + (nt * 16 + l15) * kPage + g * 8);   // K/V pages: read once per step
            else v[nt] = ld16<bf16x8>(vp + (nt * 16 + l15) * kPage + g * 8);
        }
    };
    auto load_k = [&](int pg, bf16x8 (&k)[2][2]) { load_k_at(bt[pg], k); };
    auto load_v = [&](int pg, bf16x8 (&v)[4]) { load_v_at(bt[pg], v); };
    // K pages do not depend on this step's q/k/v: they stream under the RoPE prologue (the slot of the token appended
    // below is overridden from LDS, whatever the page held).  Register ring of kDepth pages per wave.
    bf16x8 kq[kDepth][2][2];
    bf16x8 vq[kDepth][4];
    bf16x8 qB[2];
    bf16_t vrow_new = 0;       // kPre: element tid of this step's v row / the page of position P (threads 0..63)
    long vpage_new = 0;
    const int L = P + 1;
    const int npages = (L + kPage - 1) / kPage;
    const int last_page = npages - 1;
  if constexpr (kPre) {
    // ---- no prologue: the rotated q heads are one bf16 row (requested while the block-table entries are on their way), the
    //      K pages follow as soon as those entries are known
    {
        const bf16_t* qrow = p.qkv + (long)b * p.ld_qkv + (long)(kvh * group + (l15 < group ? l15 : 0)) * 64 + g * 16;
        qB[0] = ld16<bf16x8>(qrow);
        qB[1] = ld16<bf16x8>(qrow + 8);
    }
    // this step's v row (bf16, from the fused QKV kernel) and the page it belongs in: requested BEFORE the K pages (a wave's loads
    // return in order), used only after pass 1 -- a wave that had to wait for them first would issue its next K page a whole memory
    // latency late and hold the other three up at the merge barrier (measured: +1 us per launch)
    if (tid < 64) vrow_new = p.qkv[(long)b * p.ld_qkv + (long)(p.nh + p.nkv + kvh) * 64 + tid];
    if (st != 1) return;  // block-uniform
    mark(1);
    if (tid < 64) vpage_new = bt[P / kPage];
#pragma unroll
    for (int j = 0; j < kDepth; ++j) load_k_at(bt0[j], kq[j]);
    if constexpr (kVar & 4) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) load_v_at(bt0[j], vq[j]);
    }
    if (l15 >= group) {   // heads beyond the GQA group: zero columns of Q^T
#pragma unroll
        for (int e = 0; e < 8; ++e) { qB[0][e] = 0; qB[1][e] = 0; }
    }
    mark(2);
  } else {
    if constexpr (!(kVar & 1)) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) load_k_at(bt0[j], kq[j]);
    }
    // prologue work items: t = head * 32 + pair index; heads 0 .. group-1 are the q heads, item head == group is k (+ v)
    const int nitems = (group + 1) * 32;
    // element `col` of this sequence's q|k|v row as the bf16 nn.Linear output, rebuilt from the QKV GEMV's split-K slabs (all
    // loads independent; the additions follow the slab order, then + bias, then ONE rounding)
    // Branch-free on purpose: with a data-dependent slab count in the control flow the compiler waits for each element's
    // loads before it requests the next element's (a chain of round trips); here every request of the prologue is in
    // flight at once.  Absent slabs re-read the last present one and are dropped by a select.
    const float* sbase[kAttnMaxSlabs];
#pragma unroll
    for (int sl = 0; sl < kAttnMaxSlabs; ++sl) {
        const int su = sl < p.nslab ? sl : (p.nslab > 0 ? p.nslab - 1 : 0);
        sbase[sl] = p.qkv_slabs + ((long)su * p.slab_rows + b) * p.ld_qkv;
    }
    struct QkvReq { float part[kAttnMaxSlabs]; bf16_t bias; };
    auto qkv_request = [&](int col, QkvReq& q) {
#pragma unroll
        for (int sl = 0; sl < kAttnMaxSlabs; ++sl) q.part[sl] = sbase[sl][col];
        q.bias = p.qkv_bias[col];
    };
    auto qkv_value = [&](const QkvReq& q) -> bf16_t {
        float a = q.part[0];
#pragma unroll
        for (int sl = 1; sl < kAttnMaxSlabs; ++sl) a += sl < p.nslab ? q.part[sl] : 0.f;   // + 0.f: exact
        return f2bf(a + bf2f(q.bias));
    };
    constexpr int ITS = ((kGroupMax + 1) * 32 + NT - 1) / NT;   // prologue items per thread
    QkvReq qx1[ITS], qx2[ITS], qv1[ITS], qv2[ITS];
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int t = tid + it * NT;
        if (t < nitems) {
            const int hh = t >> 5, i = t & 31;
            const int c0 = hh < group ? (kvh * group + hh) * 64 : (p.nh + kvh) * 64;
            qkv_request(c0 + i, qx1[it]);
            qkv_request(c0 + i + 32, qx2[it]);
            if (hh == group) {
                const int v0 = (p.nh + p.nkv + kvh) * 64;
                qkv_request(v0 + i, qv1[it]);
                qkv_request(v0 + i + 32, qv2[it]);
            }
        }
    }
    if (st != 1) return;  // block-uniform
    mark(1);
    long new_page = 0;
    bf16_t rc[ITS], rs[ITS];
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int t = tid + it * NT;
        rc[it] = rs[it] = 0;
        if (t < nitems) {
            rc[it] = p.rope_cos[(long)P * 32 + (t & 31)];
            rs[it] = p.rope_sin[(long)P * 32 + (t & 31)];
            if ((t >> 5) == group) new_page = bt[P / kPage];
        }
    }
    if constexpr (kVar & 1) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) load_k_at(bt0[j], kq[j]);
    }
    // kVar & 4: the first V^T pages are requested right behind the first K pages instead of after the score pass -- at small
    // batch the kernel is one chain of dependent round trips (block table -> K -> scores -> V -> PV) and this removes one
    if constexpr (kVar & 4) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) load_v_at(bt0[j], vq[j]);   // (a page index past the context addresses a valid page; unused)
    }

    // ---- prologue: RoPE(q), RoPE(k) + append k, v to the cache (and keep them in LDS for this step)
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int t = tid + it * NT;
        if (t < nitems) {
            const int hh = t >> 5, i = t & 31;
            float o1, o2;
            rope_pair(bf2f(qkv_value(qx1[it])), bf2f(qkv_value(qx2[it])), bf2f(rc[it]), bf2f(rs[it]), o1, o2);
            if (hh < group) {
                qs[hh][i] = f2bf(o1);
                qs[hh][i + 32] = f2bf(o2);
            } else {
                const bf16_t k1 = f2bf(o1), k2 = f2bf(o2), v1 = qkv_value(qv1[it]), v2 = qkv_value(qv2[it]);
                knew[i] = k1; knew[i + 32] = k2; vnew[i] = v1; vnew[i + 32] = v2;
                const int slot = P % kPage;
                bf16_t* kd = p.kpool + ((new_page * p.nkv + kvh) * kPage + slot) * 64;
                kd[i] = k1; kd[i + 32] = k2;
                bf16_t* vd = p.vpool + (new_page * p.nkv + kvh) * 64 * kPage + v_slot(slot);
                vd[(long)i * kPage] = v1; vd[(long)(i + 32) * kPage] = v2;
            }
        }
    }
    for (int t = tid; t < (16 - group) * 64; t += NT) qs[group + t / 64][t % 64] = 0;
    sync();
    mark(2);

    qB[0] = ld16<bf16x8>(&qs[l15][g * 16]);
    qB[1] = ld16<bf16x8>(&qs[l15][g * 16 + 8]);

  }

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
                bf16x8 kc[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) { kc[u][0] = kq[j][u][0]; kc[u][1] = kq[j][u][1]; }
                if (pg + NW * kDepth < npages) load_k(pg + NW * kDepth, kq[j]);
                if constexpr (!kPre) {
                    if (pg == last_page) {  // the token appended this step comes from LDS, not from HBM
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            if (pg * kPage + u * 16 + l15 == P) {
                                kc[u][0] = ld16<bf16x8>(&knew[g * 16]);
                                kc[u][1] = ld16<bf16x8>(&knew[g * 16 + 8]);
                            }
                    }
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
                        float s = rbf(a[r]) * 0.125f;        // matmul out (bf16) * scaling (bf16): the x 2^-3 is exact
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
    if constexpr (kPre) {
        if (tid < 64) {   // v row -> its slot of the transposed page, and LDS (pass 2 reads it there, behind the merge barrier)
            vnew[tid] = vrow_new;
            p.vpool[(vpage_new * p.nkv + kvh) * 64 * kPage + (long)tid * kPage + v_slot(P % kPage)] = vrow_new;
        }
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
    f32x4 oacc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) oacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int pg0 = w; pg0 < npages; pg0 += NW * kDepth) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) {
            const int pg = pg0 + NW * j;
            if (pg < npages) {
                bf16x8 vc[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) vc[nt] = vq[j][nt];
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
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int key = pg * kPage + (e < 4 ? g * 4 + e : 16 + g * 4 + e - 4);
                            short val = vc[nt][e];
                            if (key == P) val = (short)vnew[nt * 16 + l15];
                            if (key > P) val = 0;
                            vc[nt][e] = val;
                        }
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) oacc[nt] = mfma16(pA, vc[nt], oacc[nt]);
            }
        }
    }
    mark(6);
    // D: col = d (l15 within tile nt), row = head g*4 + r
    if (g < 2) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ored[w][g * 4 + r][nt * 16 + l15] = oacc[nt][r];
    }
    sync();
    for (int t = tid; t < group * 64; t += NT) {
        const int hh = t >> 6, d = t & 63;
        float o = ored[0][hh][d];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) o += ored[ww][hh][d];            // ascending wave order
        if (p.out_fp8_inv > 0.f) ((unsigned char*)p.out)[(long)b * p.ld_out + (kvh * group + hh) * 64 + d] = f2fp8c(rbf(o) * p.out_fp8_inv);
        else p.out[(long)b * p.ld_out + (kvh * group + hh) * 64 + d] = f2bf(o);
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

template <bool kPre>
NTTS_KERNEL(256) void attn_split_scores_kernel(AttnSplitArgs q) {
    constexpr int NW = 4, NT = 256;
    const AttnDecodeArgs& p = q.a;
    NTTS_SHARED bf16_t qs[16][64];
    NTTS_SHARED bf16_t knew[64];
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
    const int last_page = npages - 1;
    const int ppc = (npages + q.nsplit - 1) / q.nsplit;          // pages per chunk
    const int pg_lo = ch * ppc, pg_hi = (pg_lo + ppc < npages) ? pg_lo + ppc : npages;
    bf16x8 qB[2];
  if constexpr (kPre) {   // rotated q rows + an appended K / V^T entry from the fused QKV kernel (qkv_rope.h)
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
  } else {
    // ---- prologue operands (every chunk rebuilds q; all of them need the new k for the page that holds position P)
    const int nitems = (group + 1) * 32;
    const float* sbase[kAttnMaxSlabs];
#pragma unroll
    for (int sl = 0; sl < kAttnMaxSlabs; ++sl) {
        const int su = sl < p.nslab ? sl : (p.nslab > 0 ? p.nslab - 1 : 0);
        sbase[sl] = p.qkv_slabs + ((long)su * p.slab_rows + b) * p.ld_qkv;
    }
    auto qkv_value = [&](int col) -> bf16_t {
        float part[kAttnMaxSlabs];
#pragma unroll
        for (int sl = 0; sl < kAttnMaxSlabs; ++sl) part[sl] = sbase[sl][col];
        float a = part[0];
#pragma unroll
        for (int sl = 1; sl < kAttnMaxSlabs; ++sl) a += sl < p.nslab ? part[sl] : 0.f;
        return f2bf(a + bf2f(p.qkv_bias[col]));
    };
    constexpr int ITS = ((kGroupMax + 1) * 32 + NT - 1) / NT;
    bf16_t x1[ITS], x2[ITS], v1[ITS], v2[ITS];
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int t = tid + it * NT;
        x1[it] = x2[it] = v1[it] = v2[it] = 0;
        if (t < nitems) {
            const int hh = t >> 5, i = t & 31;
            const int c0 = hh < group ? (kvh * group + hh) * 64 : (p.nh + kvh) * 64;
            x1[it] = qkv_value(c0 + i);
            x2[it] = qkv_value(c0 + i + 32);
            if (hh == group && ch == 0) {
                const int v0 = (p.nh + p.nkv + kvh) * 64;
                v1[it] = qkv_value(v0 + i);
                v2[it] = qkv_value(v0 + i + 32);
            }
        }
    }
    if (st != 1) return;  // block-uniform
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int t = tid + it * NT;
        if (t < nitems) {
            const int hh = t >> 5, i = t & 31;
            const float c = bf2f(p.rope_cos[(long)P * 32 + i]), sn = bf2f(p.rope_sin[(long)P * 32 + i]);
            float o1, o2;
            rope_pair(bf2f(x1[it]), bf2f(x2[it]), c, sn, o1, o2);
            if (hh < group) {
                qs[hh][i] = f2bf(o1);
                qs[hh][i + 32] = f2bf(o2);
            } else {
                const bf16_t k1 = f2bf(o1), k2 = f2bf(o2);
                knew[i] = k1; knew[i + 32] = k2;
                if (ch == 0) {                                     // chunk 0 appends the token to the cache
                    const long new_page = bt[P / kPage];
                    const int slot = P % kPage;
                    bf16_t* kd = p.kpool + ((new_page * p.nkv + kvh) * kPage + slot) * 64;
                    kd[i] = k1; kd[i + 32] = k2;
                    bf16_t* vd = p.vpool + (new_page * p.nkv + kvh) * 64 * kPage + v_slot(slot);
                    vd[(long)i * kPage] = v1[it]; vd[(long)(i + 32) * kPage] = v2[it];
                }
            }
        }
    }
    for (int t = tid; t < (16 - group) * 64; t += NT) qs[group + t / 64][t % 64] = 0;
    sync();
    qB[0] = ld16<bf16x8>(&qs[l15][g * 16]);
    qB[1] = ld16<bf16x8>(&qs[l15][g * 16 + 8]);
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
        if constexpr (!kPre) {
            if (pg == last_page) {  // the token appended this step comes from LDS (chunk 0's store may not have landed)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (pg * kPage + u * 16 + l15 == P) {
                        kc[u][0] = ld16<bf16x8>(&knew[g * 16]);
                        kc[u][1] = ld16<bf16x8>(&knew[g * 16 + 8]);
                    }
            }
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

inline void attn_split_launch(const AttnSplitArgs& q, int batch, hipStream_t s, bool combine = false, bool pre = false) {
    const dim3 grid(batch, q.a.nkv, q.nsplit), block(256);
    if (pre) NTTS_LAUNCH((attn_split_scores_kernel<true>), grid, block, s, q);
    else NTTS_LAUNCH((attn_split_scores_kernel<false>), grid, block, s, q);
    NTTS_LAUNCH((attn_split_pv_kernel), grid, block, s, q);
    if (combine) NTTS_LAUNCH((attn_split_combine_kernel), dim3(batch), block, s, q);
}

// tile path, behind the fused QKV kernel (qkv_rope.h): q rows rotated, the new K entry already in its page.  One KV page per wave
// in flight (deeper register rings: +0.2 / +1.2 us, r01d), V^T pages requested after the score pass (next to the K pages: 19.3 ->
// 20.0 us, profiles/r03a_sweep_qkv_fused.log); engines with max_context <= 1024 take the instantiation with half the score rows.
inline void attn_decode_launch_pre(const AttnDecodeArgs& p, int batch, hipStream_t s, int max_ctx) {
    const dim3 grid(batch, p.nkv), block(256);
    if (p.tl) NTTS_LAUNCH((attn_decode_kernel<1, true, 1, 4, kAttnLMax, true>), grid, block, s, p);   // diagnostics: phase timestamps
    else if (max_ctx <= 1024) NTTS_LAUNCH((attn_decode_kernel<1, false, 1, 4, 1024, true>), grid, block, s, p);
    else NTTS_LAUNCH((attn_decode_kernel<1, false, 1, 4, kAttnLMax, true>), grid, block, s, p);
}
// small-batch path (gemv.h): the QKV GEMV's fp32 split-K slabs are reduced, rotated and appended in this kernel's prologue; two pages
// per wave in flight, V^T requested next to K (kVar 7): at batch 1 the kernel is one chain of dependent round trips
inline void attn_decode_launch_small(const AttnDecodeArgs& p, int batch, hipStream_t s) {
    const dim3 grid(batch, p.nkv);
    if (p.tl) NTTS_LAUNCH((attn_decode_kernel<2, true, 7, 4>), grid, dim3(256), s, p);   // diagnostics: phase timestamps
    else NTTS_LAUNCH((attn_decode_kernel<2, false, 7, 4>), grid, dim3(256), s, p);
}

}  // namespace ntts
