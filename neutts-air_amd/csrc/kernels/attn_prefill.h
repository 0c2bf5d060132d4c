// attn_prefill.h -- prompt phase: RoPE + KV-page fill, then causal GQA attention over the pages.
//
// Replaces, for q_len = S (hf:models/qwen2/modeling_qwen2.py):
//   apply_rotary_pos_emb :113-135, DynamicCache.update hf:cache_utils.py:127-146,
//   eager_attention_forward :150-172 with the causal mask of create_causal_mask (:377-379).
// Same matrix-core mapping as attn_decode.h (S^T = K Q^T, accumulator == PV's A-operand); two sweeps
// over the keys (1: online row max / denominator, 2: normalised bf16 probabilities -> PV), because
// the eager contract rounds P = bf16(softmax) BEFORE the PV product.  K/V are read back from the pages
// the rope kernel just wrote (L2/MALL resident), so prefill exercises exactly the layout decode reads.
#pragma once
#include <ntts/dev.h>
#include <type_traits>
#include "attn_decode.h"

namespace ntts {

struct PrefillMeta {           // one entry per prompt of the packed batch (device arrays)
    const int* tok_base;       // [n] first packed row of prompt i
    const int* seq_len;        // [n] TOTAL context length after this pass (cached prefix + packed tokens)
    const int* pos0;           // [n] tokens of the prompt already in its KV pages (shared prefix; multiple of the page
                               //     size): the packed rows of prompt i are positions pos0[i] .. seq_len[i]-1
    const int* slot;           // [n] decode slot -> block_table row
    const int* tok_seq;        // [T] prompt index of packed row t
    const int* tile_seq;       // [n_tiles] work list of 64-row query tiles
    const int* tile_q0;        // [n_tiles] ABSOLUTE position of the tile's first query (pos0 + 64 k)
};

struct RopeWriteArgs {
    bf16_t* qkv;               // [T][ld_qkv]  q part rotated in place
    long ld_qkv;
    bf16_t* kpool;
    bf16_t* vpool;
    const int* block_table;
    int max_pages;
    PrefillMeta meta;
    const bf16_t* rope_cos;
    const bf16_t* rope_sin;
    int nh, nkv, T;
    int skip_q;                // rope_kv_write_vec_kernel: leave the q heads alone (attn_prefill_gqa_kernel rotates them as it loads them)
};

// 16-byte accesses: a work item = (token, head, 8 consecutive pairs) loads x[i0..i0+7], x[i0+32..i0+39] and
// the 8 cos / sin of its position as four 16-byte loads and stores the rotated halves as two (q in place, k into its page
// row); 4 tokens per workgroup.  Only the V^T scatter (a token's 64 values go to 64 rows of its page) stays element-wise.
// Same arithmetic per element (rope_pair) as the element-wise kernel it replaced (bit-identical results); 58 -> 30 us per launch
// at 32 000 tokens.  (ld_qkv % 8 == 0: the engine's QKV width is a multiple of 64.)
constexpr int kRopeTokPerBlock = 4;
NTTS_KERNEL(256) void rope_kv_write_vec_kernel(RopeWriteArgs p) {
    const int h_first = p.skip_q ? p.nh : 0;                     // skip_q: only the k and v heads have work here
    const int nheads = p.nh + 2 * p.nkv - h_first;
    const int per_tok = nheads * 4;                              // work items per token
    const int t0 = blockIdx.x * kRopeTokPerBlock;
    for (int x = threadIdx.x; x < per_tok * kRopeTokPerBlock; x += 256) {
        const int tt = x / per_tok, y = x - tt * per_tok;
        const int t = t0 + tt;
        if (t >= p.T) break;
        const int hh = h_first + (y >> 2), i0 = (y & 3) * 8;
        const int sq = p.meta.tok_seq[t];
        const int pos = p.meta.pos0[sq] + t - p.meta.tok_base[sq];
        bf16_t* h = p.qkv + (long)t * p.ld_qkv + hh * 64;
        const bf16x8 x1 = ld16<bf16x8>(h + i0), x2 = ld16<bf16x8>(h + i0 + 32);
        if (hh < p.nh + p.nkv) {
            const bf16x8 cv = ld16<bf16x8>(p.rope_cos + (long)pos * 32 + i0), sv = ld16<bf16x8>(p.rope_sin + (long)pos * 32 + i0);
            bf16x8 r1, r2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float o1, o2;
                rope_pair(bf2f((bf16_t)x1[e]), bf2f((bf16_t)x2[e]), bf2f((bf16_t)cv[e]), bf2f((bf16_t)sv[e]), o1, o2);
                r1[e] = (short)f2bf(o1);
                r2[e] = (short)f2bf(o2);
            }
            bf16_t* dst = h;
            if (hh >= p.nh) {
                const int* bt = p.block_table + (long)p.meta.slot[sq] * p.max_pages;
                const long pg = bt[pos / kPage];
                dst = p.kpool + ((pg * p.nkv + (hh - p.nh)) * kPage + pos % kPage) * 64;
            }
            *(bf16x8*)(dst + i0) = r1;
            *(bf16x8*)(dst + i0 + 32) = r2;
        } else {
            const int* bt = p.block_table + (long)p.meta.slot[sq] * p.max_pages;
            const long pg = bt[pos / kPage];
            bf16_t* vd = p.vpool + (pg * p.nkv + (hh - p.nh - p.nkv)) * 64 * kPage + v_slot(pos % kPage);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                vd[(long)(i0 + e) * kPage] = (bf16_t)x1[e];
                vd[(long)(i0 + 32 + e) * kPage] = (bf16_t)x2[e];
            }
        }
    }
}

// ---- The same step for the GENERAL attention geometry (round 6): head_dim HD = 64 or 128, optional per-head q / k RMSNorm (Qwen3-style
// `self_attn.q_norm` / `k_norm`, hf:models/qwen3/modeling_qwen3.py Qwen3Attention.forward: norm over head_dim on the projected heads BEFORE
// RoPE).  One wave per (row, head): lane l holds the head's values l and l + HD / 2 -- a RoPE pair (HD = 128; for 64 lanes 0..31 hold one pair
// each) -- so the norm is one cross-lane sum and the rotation needs no exchange.  q heads are normalised + rotated IN PLACE (the attention
// kernels then load them as they are), k heads go into their page row, v heads into the transposed page.  Used for the prompt pass (rows =
// packed prompt tokens, positions from PrefillMeta) and for the decode step (rows = decode slots, positions from the slot arrays: `dec_*`).
// Arithmetic per element = Qwen2RMSNorm (norm.h rmsnorm: fp32 mean of squares, rsqrt, ONE rounding, times the bf16 weight, one rounding) and
// rope_pair; nothing here is on the NeuTTS-Air path (head_dim 64, no qk-norm keeps the fused kernels above / qkv_rope.h).
struct RopeNormArgs {
    bf16_t* qkv;               // [rows][ld_qkv]
    long ld_qkv;
    bf16_t* kpool;
    bf16_t* vpool;
    const int* block_table;
    int max_pages;
    PrefillMeta meta;          // prompt pass (dec_pos == null)
    const int* dec_pos;        // decode step: [rows] position of the token being decoded (= tokens already cached), row = decode slot
    const int* dec_state;      //              [rows] 1 = running (others are skipped)
    const bf16_t* rope_cos;    // [max_ctx][HD / 2]
    const bf16_t* rope_sin;
    const bf16_t* q_norm;      // [HD] or null (no qk-norm)
    const bf16_t* k_norm;
    float eps;
    int nh, nkv, rows;
    int write_v;               // prompt pass: v heads into the transposed pages here; decode step: the attention kernel places its own v row (0)
};
template <int HD>
NTTS_KERNEL(256) void rope_norm_kv_write_kernel(RopeNormArgs p) {
    constexpr int HALF = HD / 2;
    const int lane = lane_id();
    const int nheads = p.nh + 2 * p.nkv;
    const long item = (long)blockIdx.x * 4 + wave_id();            // (row, head)
    if (item >= (long)p.rows * nheads) return;                       // wave-uniform
    const int t = (int)(item / nheads), hh = (int)(item % nheads);
    int pos, slot_row;
    if (p.dec_pos) {
        if (p.dec_state[t] != 1) return;
        pos = p.dec_pos[t];
        slot_row = t;
    } else {
        const int sq = p.meta.tok_seq[t];
        pos = p.meta.pos0[sq] + t - p.meta.tok_base[sq];
        slot_row = p.meta.slot[sq];
    }
    bf16_t* h = p.qkv + (long)t * p.ld_qkv + (long)hh * HD;
    const bool live = lane < HALF;                                  // HD = 64: half the wave idles (this is not the NeuTTS-Air path)
    const int i = live ? lane : 0;
    float x1 = bf2f(h[i]), x2 = bf2f(h[i + HALF]);
    const int* bt = p.block_table + (long)slot_row * p.max_pages;
    if (hh >= p.nh + p.nkv) {                                       // v head: transposed page, element-wise scatter
        if (p.write_v && live) {
            const long pg = bt[pos / kPage];
            bf16_t* vd = p.vpool + (pg * p.nkv + (hh - p.nh - p.nkv)) * HD * kPage + v_slot(pos % kPage);
            vd[(long)i * kPage] = h[i];
            vd[(long)(i + HALF) * kPage] = h[i + HALF];
        }
        return;
    }
    const bf16_t* nw = hh < p.nh ? p.q_norm : p.k_norm;
    if (nw) {                                                       // Qwen3RMSNorm over the head (wave-uniform branch)
        float ss = live ? x1 * x1 + x2 * x2 : 0.f;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) ss += shfl_xor(ss, sh);
        const float inv = frsqrt_exact(ss / (float)HD + p.eps);
        x1 = rbf(bf2f(nw[i]) * rbf(x1 * inv));
        x2 = rbf(bf2f(nw[i + HALF]) * rbf(x2 * inv));
    }
    float o1, o2;
    rope_pair(x1, x2, bf2f(p.rope_cos[(long)pos * HALF + i]), bf2f(p.rope_sin[(long)pos * HALF + i]), o1, o2);
    bf16_t* dst = h;
    if (hh >= p.nh) {
        const long pg = bt[pos / kPage];
        dst = p.kpool + ((pg * p.nkv + (hh - p.nh)) * kPage + pos % kPage) * HD;
    }
    if (live) { dst[i] = f2bf(o1); dst[i + HALF] = f2bf(o2); }
}

struct AttnPrefillArgs {
    const bf16_t* qkv;         // q already rotated
    long ld_qkv;
    bf16_t* out;               // [T][nh*64]
    long ld_out;
    float out_fp8_inv;         // > 0: `out` holds e4m3 BYTES, value = bf16 result * out_fp8_inv (the o_proj input of the fp8 model)
    const bf16_t* kpool;
    const bf16_t* vpool;
    const int* block_table;
    int max_pages;
    PrefillMeta meta;
    int nh, nkv;
    const bf16_t* rope_cos;    // attn_prefill_gqa_kernel: non-null = `qkv` holds the q heads as the QKV GEMM left them and the kernel applies
    const bf16_t* rope_sin;    // RoPE while it loads its Q fragments (rope_kv_write_vec_kernel then only handles k and v: skip_q)
    int q_cap;                 // attn_prefill_res_kernel: its queries (and, causally, its keys) are the positions below q_cap <= kPfResPages * 32
    int heads_per_wg;          // attn_prefill_res_kernel: query heads of the group one workgroup walks (grid z covers the rest)
    int only_last;             // attn_prefill_res_kernel: only the 16-query block that holds the prompt's last position is computed (last layer)
    int q_lo;                  // attn_prefill_deep_kernel: its queries are the positions q_lo <= q < q_cap (q_lo = the resident kernel's q_cap)
};

// ------------------------------------------------------------------------------------------------
// GQA-shared prefill attention: one workgroup per (64-query tile, kv-head).  The `group` query heads that share
// a kv-head (7 for NeuTTS-Air) are processed together, so every K / V^T page is brought into LDS ONCE per
// workgroup (LDS-DMA, double-buffered) and feeds 4 waves x GH heads, instead of being re-read from L2 by every
// (head, 16-query wave) (24x less load-path traffic than the first, per-head kernel).
// Eager contract: two sweeps over the keys, P = bf16(softmax) before PV.
//   LDS images (lane-linear LDS-DMA, swizzled on the SOURCE side):
//     K   page [32 keys][128 B]: 16-B chunk c of key r stored at chunk c ^ (r & 7)        (ds_read_b128 conflict-free)
//     V^T page [64 d][64 B]:     16-B unit  u of row d stored at unit  u ^ ((d >> 2) & 3) (ds_read_b128 conflict-free)
// Measured on MI355X and removed (profiles/r02k_sweep_pf_attn_8waves.log): the 7 heads split 4 + 3 over the two halves of an
// 8-wave workgroup (231 VGPRs, two waves per SIMD instead of one at 256 + 138 AGPRs; pages still staged once): prefill chunk
// 28.23-28.29 vs 28.31-28.42 ms -- the kernel is bound by the softmax arithmetic itself (two exp per score: the eager contract
// needs the global denominator before P is rounded), not by what one wave per SIMD cannot hide.
// HD = head_dim (64; 128 for Qwen3-style checkpoints, round 6: attn_decode.h).  The HD = 64 instantiations are what they were, bit for bit.
template <int GH, int HD = 64>
NTTS_KERNEL(256) void attn_prefill_gqa_kernel(AttnPrefillArgs p) {
    constexpr int KS = HD / 32, NTV = HD / 16, PB = kPage * HD;      // matrix-core k-steps per key, PV tiles, elements of one page image
    NTTS_SHARED bf16_t lds[2 * 2 * PB];            // [buf][K | V^T] 4 KB (8 KB at HD = 128) each
    const int lane = lane_id(), w = wave_id();
    const int g = lane >> 4, l15 = lane & 15;
    // grid = (kv-head, tile, pass): the kv-head is the FAST index, so that with the work list sorted by descending causal
    // depth (prefill_impl) the dispatch order is longest-first over all (tile, kv-head) pairs and the short tiles fill the tail
    const int tile = blockIdx.y;
    const int sq = p.meta.tile_seq[tile];
    const int S = p.meta.seq_len[sq];
    const int base = p.meta.tok_base[sq] - p.meta.pos0[sq];   // packed row of absolute position q is base + q
    const int kvh = blockIdx.x;
    const int group = p.nh / p.nkv;
    const int h0 = kvh * group + blockIdx.z * GH;            // first query head of this pass
    int nhd = group - blockIdx.z * GH;                        // heads handled here (<= GH)
    if (nhd > GH) nhd = GH;
    const int* bt = p.block_table + (long)p.meta.slot[sq] * p.max_pages;
    const int q0 = p.meta.tile_q0[tile];
    const int qw0 = q0 + w * 16;                              // first query of this wave
    const bool wave_live = qw0 < S;                           // dead waves still help loading and hit the barriers
    int qpos = qw0 + l15;
    if (qpos > S - 1) qpos = S - 1;
    const int qlast_blk = (q0 + 63 < S ? q0 + 63 : S - 1);
    const int npages = qlast_blk / kPage + 1;                 // pages the block needs (causal)
    const int qlast_w = (qw0 + 15 < S ? qw0 + 15 : S - 1);
    const int npages_w = wave_live ? qlast_w / kPage + 1 : 0; // pages this wave computes on

    // ---- Q fragments of all heads (B operand: column = query l15, k = d g*8.. within each 32-wide half)
    bf16x8 qB[GH][KS];
#pragma unroll
    for (int h = 0; h < GH; ++h) {
        const int hh = h < nhd ? h0 + h : h0;
        const bf16_t* qr = p.qkv + (long)(base + qpos) * p.ld_qkv + hh * HD;
#pragma unroll
        for (int f = 0; f < KS; ++f) qB[h][f] = ld16<bf16x8>(qr + (HD == 64 ? g * 16 + f * 8 : f * 32 + g * 8));
    }
    if constexpr (HD == 64) if (p.rope_cos) {
        // RoPE of the queries on the way in (hf:models/qwen2/modeling_qwen2.py:113-135, the same rope_pair arithmetic the rope kernel
        // applies: bit-identical q): lane (g, l15) holds d = 16 g .. 16 g + 15 of query qpos; the pair partner d +- 32 sits in lane g ^ 2
        const bf16_t* cr = p.rope_cos + (long)qpos * 32 + (g & 1) * 16;
        const bf16_t* sr = p.rope_sin + (long)qpos * 32 + (g & 1) * 16;
        bf16x8 cv[2], sv[2];
        cv[0] = ld16<bf16x8>(cr); cv[1] = ld16<bf16x8>(cr + 8);
        sv[0] = ld16<bf16x8>(sr); sv[1] = ld16<bf16x8>(sr + 8);
#pragma unroll
        for (int h = 0; h < GH; ++h)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const u32x4 own = __builtin_bit_cast(u32x4, qB[h][f]);
                u32x4 oth;
#pragma unroll
                for (int k = 0; k < 4; ++k) oth[k] = (unsigned int)shfl_xor((int)own[k], 32);
                const bf16x8 other = __builtin_bit_cast(bf16x8, oth);
                bf16x8 r;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float mine = bf2f((bf16_t)qB[h][f][e]), theirs = bf2f((bf16_t)other[e]);
                    float o1, o2;
                    rope_pair(g < 2 ? mine : theirs, g < 2 ? theirs : mine, bf2f((bf16_t)cv[f][e]), bf2f((bf16_t)sv[f][e]), o1, o2);
                    r[e] = (short)f2bf(g < 2 ? o1 : o2);
                }
                qB[h][f] = r;
            }
    }

    // ---- page loader: wave 0/1 bring K (2 KB each), wave 2/3 bring V^T; one LDS-DMA instruction covers 1 KB
    constexpr int KC = HD / 8;                                // 16-byte chunks per key row (8 / 16): chunk c of key r sits at c ^ (r & (KC - 1))
    auto stage = [&](int pg, int buf) {
        const long page = bt[pg];
        bf16_t* dst = lds + buf * (2 * PB);
        if (w < 2) {
            const bf16_t* kp = p.kpool + (page * p.nkv + kvh) * PB;
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                const int inst = w * KS + i;                   // 64 / KC keys per instruction (8; 4 at HD = 128)
                const int r = inst * (64 / KC) + lane / KC;
                const int c = (lane % KC) ^ (r & (KC - 1));
                glds16(kp + r * HD + c * 8, dst + inst * 512);
            }
        } else {
            const bf16_t* vp = p.vpool + (page * p.nkv + kvh) * PB;
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                const int inst = (w - 2) * KS + i;             // 16 d-rows per instruction (64 B each)
                const int d = inst * 16 + (lane >> 2);
                const int u = (lane & 3) ^ ((d >> 2) & 3);
                glds16(vp + d * kPage + u * 8, dst + PB + inst * 512);
            }
        }
    };
    // K fragments of a page for this lane: key row u*16 + l15, k-step f = logical chunk 2g + f (HD = 64) / 4f + g (HD = 128: d = 32 f + 8 g)
    auto load_k = [&](const bf16_t* kb, bf16x8 (&kf)[2][KS]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = u * 16 + l15;
#pragma unroll
            for (int f = 0; f < KS; ++f) {
                const int ch = HD == 64 ? 2 * g + f : 4 * f + g;
                kf[u][f] = ld16<bf16x8>(kb + r * HD + ((ch ^ (r & (KC - 1))) << 3));
            }
        }
    };
    // scores of one head against one page.  bf16(QK^T) * scaling: the product by 2^-3 is exact, one rounding suffices.
    // MASKED pages (those reaching past the wave's first query) apply the causal mask with a large finite value so the
    // fast exp never sees an infinity; pages entirely below the diagonal skip the compare/select.
    constexpr float kMasked = -1.0e30f;
    auto scores = [&](const bf16x8 (&kf)[2][KS], int h, int pg, float (&sc)[8], auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < KS; ++f) a = mfma16(kf[u][f], qB[h][f], a);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = rbf(a[r]) * attn_scale(HD);
                if constexpr (HD != 64) v = rbf(v);            // (x 128^-0.5 rounds; x 2^-3 is exact)
                if constexpr (MASKED) {
                    const int key = pg * kPage + u * 16 + g * 4 + r;
                    if (key > qpos) v = kMasked;               // causal (covers key >= S as qpos <= S-1)
                }
                sc[u * 4 + r] = v;
            }
        }
    };
    using Masked = std::integral_constant<bool, true>;
    using Clear = std::integral_constant<bool, false>;

    // ---- sweep 1: row max and softmax denominator per head (online, lane-local; merged across key groups after)
    float m[GH], sum[GH];
#pragma unroll
    for (int h = 0; h < GH; ++h) { m[h] = kMasked; sum[h] = 0.f; }
    auto sweep1_page = [&](const bf16x8 (&kf)[2][KS], int pg, auto masked_c) {
#pragma unroll
        for (int h = 0; h < GH; ++h) {
            float sc[8];
            scores(kf, h, pg, sc, masked_c);
            float tm = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
            const float mn = fmaxf(m[h], tm);
            float add = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) add += fexp_neg(sc[e] - mn);
            sum[h] = sum[h] * fexp_neg(m[h] - mn) + add;       // lanes whose keys are all masked carry junk that the
            m[h] = mn;                                         // merge below wipes out (their m stays at kMasked)
        }
    };
    stage(0, 0);
    for (int pg = 0; pg < npages; ++pg) {
        wait_vmem();
        sync();                                               // page pg landed; everyone done with the other buffer
        if (pg + 1 < npages) stage(pg + 1, (pg + 1) & 1);
        if (pg < npages_w) {
            bf16x8 kf[2][KS];
            load_k(lds + (pg & 1) * (2 * PB), kf);
            if (pg * kPage + kPage - 1 > qw0) sweep1_page(kf, pg, Masked{}); else sweep1_page(kf, pg, Clear{});
        }
    }
    float rs[GH];
#pragma unroll
    for (int h = 0; h < GH; ++h) {
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) {
            const float om = shfl_xor(m[h], sh), os = shfl_xor(sum[h], sh);
            const float mn = fmaxf(m[h], om);
            sum[h] = sum[h] * fexp_neg(m[h] - mn) + os * fexp_neg(om - mn);
            m[h] = mn;
        }
        rs[h] = frcp_refined(sum[h]);
    }

    // ---- sweep 2: P = bf16(exp(s - m) / sum), O += P V
    f32x4 oacc[GH][NTV];
#pragma unroll
    for (int h = 0; h < GH; ++h)
#pragma unroll
        for (int nt = 0; nt < NTV; ++nt) oacc[h][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto sweep2_page = [&](const bf16x8 (&kf)[2][KS], const bf16x8 (&vB)[NTV], int pg, auto masked_c) {
#pragma unroll
        for (int h = 0; h < GH; ++h) {
            float sc[8];
            scores(kf, h, pg, sc, masked_c);
            bf16x8 pA;
#pragma unroll
            for (int e = 0; e < 8; ++e) pA[e] = (short)f2bf(fdiv_r(fexp_neg(sc[e] - m[h]), sum[h], rs[h]));
#pragma unroll
            for (int nt = 0; nt < NTV; ++nt) oacc[h][nt] = mfma16(pA, vB[nt], oacc[h][nt]);
        }
    };
    sync();                                                   // sweep 1's last reads are done before buffer 0 is refilled
    stage(0, 0);
    for (int pg = 0; pg < npages; ++pg) {
        wait_vmem();
        sync();
        if (pg + 1 < npages) stage(pg + 1, (pg + 1) & 1);
        if (pg < npages_w) {
            const bf16_t* kb = lds + (pg & 1) * (2 * PB);
            const bf16_t* vb = kb + PB;
            bf16x8 kf[2][KS];
            load_k(kb, kf);
            bf16x8 vB[NTV];
            const bool tail = (pg + 1) * kPage > S;            // page holds slots past the prompt: mask them
#pragma unroll
            for (int nt = 0; nt < NTV; ++nt) {
                const int d = nt * 16 + l15;
                const int sw = (d >> 2) & 3;
                vB[nt] = ld16<bf16x8>(vb + d * kPage + ((g ^ sw) << 3));   // 16-B unit g of the row = keys 4g..+3, 16+4g..+3
                if (tail) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int key = pg * kPage + (e < 4 ? g * 4 + e : 16 + g * 4 + e - 4);
                        if (key >= S) vB[nt][e] = 0;
                    }
                }
            }
            if (pg * kPage + kPage - 1 > qw0) sweep2_page(kf, vB, pg, Masked{}); else sweep2_page(kf, vB, pg, Clear{});
        }
    }
    // D: col = d (l15 of tile nt), row = query qw0 + g*4 + r
    if (wave_live) {
#pragma unroll
        for (int h = 0; h < GH; ++h) {
            if (h < nhd) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qw0 + g * 4 + r;
                    if (q < S) {
                        if (p.out_fp8_inv > 0.f) {
                            unsigned char* o = (unsigned char*)p.out + (long)(base + q) * p.ld_out + (h0 + h) * HD + l15;
#pragma unroll
                            for (int nt = 0; nt < NTV; ++nt) o[nt * 16] = f2fp8c(rbf(oacc[h][nt][r]) * p.out_fp8_inv);
                        } else {
                            bf16_t* o = p.out + (long)(base + q) * p.ld_out + (h0 + h) * HD + l15;
#pragma unroll
                            for (int nt = 0; nt < NTV; ++nt) o[nt * 16] = f2bf(oacc[h][nt][r]);
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same attention for the queries at positions below 512, with the K and V^T pages RESIDENT in LDS and ONE exp per score.
// The two-sweep kernel above pays for the eager contract twice: P = bf16(exp(s - m) / sum) needs the row's maximum and denominator
// before any P is rounded, so it walks the keys twice and computes QK^T and exp(s - m) in both sweeps (~25 issue slots + two quarter-rate
// v_exp_f32 per score, one wave per SIMD at 256 + 134 registers, a third of the slots AGPR copies and MFMA hazards: 181-210 us per launch
// at 64 x 500 tokens).  Here one workgroup (8 waves, 256 queries of one prompt and one kv-head: two 16-query blocks per wave, dealt out from
// both ends of the prompt so that every wave and every workgroup carries the same causal depth) stages every page it can see ONCE (<= 16
// pages: 64 KB of K + 64 KB of V^T, LDS-DMA, swizzled for the ds_read_b128 lane groups: no bank conflict) and walks the group's query heads
// one after the other; per head a wave keeps the scores of its 16 queries against ALL keys in registers (16 pages x 8 per lane = 128 VGPRs):
//   pass A  raw QK^T per page (matrix core) into the registers                                        (no transcendental)
//   mask + row maximum
//   pass B  in place: e = exp((bf16(s) - bf16(max)) / 8), denominator                                  (the row's ONE exp per score)
//   pass C  P = bf16(e / sum) -> PV per page (matrix core)
// which is torch's own softmax order (max, exp, sum, divide: hf:models/qwen2/modeling_qwen2.py:150-172) with every rounding point of the
// eager contract kept: bf16(QK^T), the exact 2^-3 scaling, fp32 softmax, P rounded after the normalisation, fp32 PV accumulation in
// page order.  Per query the arithmetic does not depend on the workgroup it sits in or on what else is in the pass.  One barrier when the
// shallow blocks' pages are in, one when the rest is; two waves per SIMD (<= 256 registers, no AGPR copies).  103-107 us per launch
// (profiles/r04j_*).  Queries at positions >= 512 (they would need more than 16 pages) stay with the two-sweep kernel: the host splits the
// work list by POSITION (backbone.cpp prefill_impl), so which kernel computes a query depends on nothing but where it sits in its prompt.
constexpr int kPfResPages = 16;
constexpr int kPfResTile = 256;   // queries per work item: 8 waves x 2 blocks of 16
// LDS swizzles of the resident images, built for the lane groups a ds_read_b128 is served in (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} and the same + 32 -- one clock per group when its 16 lanes fall on 16 different 16-byte slots of the 256-byte bank row).
// K row r (128 B = 8 chunks): chunk c sits at c ^ k_swz(r).  A group reads chunk c of rows {0-3, 12-15} and chunk c + 2 of rows {4-11} (lane group g
// reads chunks 2g, 2g + 1): even / odd rows own the two halves of the bank row, (r >> 1) spreads a half's 8 rows, and the extra flip of bit 1 on
// rows 4-11 keeps chunk c + 2 of those rows off the slots of chunk c of the others.  V^T row d (64 B = 4 units): unit u sits at u ^ v_swz(d); a group
// reads unit g of rows {0-3, 12-15} and unit g + 1 of rows {4-11}: same construction one bit lower.
NTTS_D int k_swz(int r) { return ((r >> 1) & 7) ^ ((((r + 4) >> 3) & 1) << 1); }
NTTS_D int v_swz(int d) { return ((d >> 2) & 3) ^ (((d + 4) >> 3) & 1); }
template <int NP>
NTTS_KERNEL(512) void attn_prefill_res_kernel(AttnPrefillArgs p) {
    NTTS_SHARED bf16_t kres[NP * kPage * 64];    // [page][32 keys][128 B], chunk c of key r at c ^ k_swz(r)
    NTTS_SHARED bf16_t vres[NP * 64 * kPage];    // [page][64 d][64 B], 16-B unit u of row d at u ^ v_swz(d)
    const int lane = lane_id(), w = wave_id();
    const int g = lane >> 4, l15 = lane & 15;
    const int tile = blockIdx.y, kvh = blockIdx.x;
    const int sq = p.meta.tile_seq[tile];
    const int S = p.meta.seq_len[sq];
    const int Sq = S < p.q_cap ? S : p.q_cap;                 // this launch's queries end here
    const int pos0 = p.meta.pos0[sq];
    const int base = p.meta.tok_base[sq] - pos0;              // packed row of absolute position q is base + q
    const int group = p.nh / p.nkv;
    const int hb = kvh * group + blockIdx.z * p.heads_per_wg;
    int he = hb + p.heads_per_wg;
    if (he > (kvh + 1) * group) he = (kvh + 1) * group;
    const int* bt = p.block_table + (long)p.meta.slot[sq] * p.max_pages;
    // ---- the workgroup's queries: the prompt's 16-query blocks 0 .. nb - 1 (padded to a multiple of 16: nbp) are dealt out so that every workgroup
    //      and every wave gets the same causal depth -- workgroup k takes the 8 blocks from 8 k up and the 8 blocks from nbp - 1 - 8 k down, wave w
    //      block 8 k + w ("lo") and then block nbp - 1 - 8 k - w ("hi"): a shallow and a deep one, the same sum for every (k, w).  (A tile of 128
    //      consecutive queries gave waves of 13 .. 16 pages next to waves of 1 .. 4: SIMD slots idle behind the 128 KB of LDS, 1.4 of 2 waves resident.)
    const int nb = (Sq - pos0 + 15) >> 4;
    const int nbp = (nb + 15) & ~15;
    const int k8 = p.meta.tile_q0[tile] * 8;                  // tile_q0 = k
    const int blk_lo = k8 + w, blk_hi = nbp - 1 - k8 - w;
    int top_lo = k8 + 7;                                      // deepest live block of each half (workgroup-uniform)
    if (top_lo > nb - 1) top_lo = nb - 1;
    int top_hi = nbp - 1 - k8;
    if (top_hi > nb - 1) top_hi = nb - 1;
    auto pages_to = [&](int blk) { const int ql = pos0 + blk * 16 + 15; return (ql < Sq - 1 ? ql : Sq - 1) / kPage + 1; };
    const int np_lo = pages_to(top_lo);                       // pages the lo blocks can see (causal), and all blocks: <= NP
    const int npages = pages_to(top_hi > top_lo ? top_hi : top_lo);

    // ---- every page the workgroup can see, requested in page order: wave w brings 1 KB piece w of each (0-3: K, 8 keys each; 4-7: V^T, 16 d-rows
    //      each).  First the pages the lo blocks need, then the lo block's first q rows, then the rest: a wave's requests retire in order, so "at most
    //      npages - np_lo still in flight" below means its pieces of pages 0 .. np_lo - 1 and those q rows have landed, and the lo blocks run while
    //      the deep pages arrive (s_waitcnt takes an immediate: one case per count)
    auto stage = [&](int pg) {
        const long page = bt[pg];
        if (w < 4) {
            const int r = w * 8 + (lane >> 3);
            const int c = (lane & 7) ^ k_swz(r);
            glds16(p.kpool + ((page * p.nkv + kvh) * kPage + r) * 64 + c * 8, kres + pg * (kPage * 64) + w * 512);
        } else {
            const int d = (w - 4) * 16 + (lane >> 2);
            const int u = (lane & 3) ^ v_swz(d);
            glds16(p.vpool + (page * p.nkv + kvh) * 64 * kPage + d * kPage + u * 8, vres + pg * (64 * kPage) + (w - 4) * 512);
        }
    };
    // RoPE operands of a lane's query: d = 16 g .. 16 g + 15, the pair partner d +- 32 is the chunk of lane group g ^ 2 (read from the row directly:
    // L2 hits instead of cross-lane traffic); o = bf16(bf16(x c) + bf16(x' s')), s' = -s below d = 32 -- rope_pair, the one output this lane keeps
    bf16x8 qn[4];                                              // next head's raw chunks: own 2, partner 2
    auto fetch_q = [&](const bf16_t* qrow, int h) {
        const bf16_t* qr = qrow + h * 64;
        qn[0] = ld16<bf16x8>(qr + g * 16);
        qn[1] = ld16<bf16x8>(qr + g * 16 + 8);
        if (p.rope_cos) {
            qn[2] = ld16<bf16x8>(qr + (g ^ 2) * 16);
            qn[3] = ld16<bf16x8>(qr + (g ^ 2) * 16 + 8);
        }
    };
    auto qpos_of = [&](int blk) { const int q = pos0 + blk * 16 + l15; return q > Sq - 1 ? Sq - 1 : q; };
    for (int pg = 0; pg < np_lo; ++pg) stage(pg);
    const bool lo_live = blk_lo < nb && !(p.only_last && blk_lo != nb - 1);
    if (lo_live) fetch_q(p.qkv + (long)(base + qpos_of(blk_lo)) * p.ld_qkv, hb);
    for (int pg = np_lo; pg < npages; ++pg) stage(pg);
    switch (npages - np_lo) {
#define NTTS_PF_CASE(n) case n: wait_vmem_le<n>(); break;
        NTTS_PF_CASE(0) NTTS_PF_CASE(1) NTTS_PF_CASE(2) NTTS_PF_CASE(3) NTTS_PF_CASE(4) NTTS_PF_CASE(5) NTTS_PF_CASE(6) NTTS_PF_CASE(7)
        NTTS_PF_CASE(8) NTTS_PF_CASE(9) NTTS_PF_CASE(10) NTTS_PF_CASE(11) NTTS_PF_CASE(12) NTTS_PF_CASE(13) NTTS_PF_CASE(14) NTTS_PF_CASE(15)
#undef NTTS_PF_CASE
        default: wait_vmem(); break;
    }
    sync_keep_dma();

    constexpr float kMasked = -1.0e30f;
    for (int half = 0; half < 2; ++half) {
    if (half == 1) { wait_vmem(); sync_keep_dma(); }           // the rest of the pages (every wave passes here, live or not)
    const int blk = half == 0 ? blk_lo : blk_hi;
    if (blk >= nb || blk < 0 || (p.only_last && blk != nb - 1)) continue;   // padding block (no barrier is skipped: the one above precedes this test)
    const int qw0 = pos0 + blk * 16;                          // first query of this wave's block
    const int qpos = qpos_of(blk);
    const int qlast_w = (qw0 + 15 < Sq ? qw0 + 15 : Sq - 1);
    const int npw = qlast_w / kPage + 1;                      // pages this block computes on; the last one holds its diagonal
    const bf16_t* qrow = p.qkv + (long)(base + qpos) * p.ld_qkv;
    bf16x8 cv[2], sv[2];
    if (p.rope_cos) {
        const bf16_t* cr = p.rope_cos + (long)qpos * 32 + (g & 1) * 16;
        const bf16_t* sr = p.rope_sin + (long)qpos * 32 + (g & 1) * 16;
        cv[0] = ld16<bf16x8>(cr); cv[1] = ld16<bf16x8>(cr + 8);
        sv[0] = ld16<bf16x8>(sr); sv[1] = ld16<bf16x8>(sr + 8);
    }
    if (half == 1) fetch_q(qrow, hb);                          // (the lo block's first rows were requested ahead of the deep pages)
    const bool vtail = npw * kPage > S;                        // the block's last page holds slots past the prompt: their V^T entries are not data
    const int qrel = qpos - (npw - 1) * kPage;                 // this lane's query and the prompt's end relative to the block's last (diagonal) page
    const int srel = S - (npw - 1) * kPage;
    for (int h = hb; h < he; ++h) {
        bf16x8 qB[2];
        if (p.rope_cos) {
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float c = bf2f((bf16_t)cv[f][e]), s0 = bf2f((bf16_t)sv[f][e]);
                    const float a = rbf(bf2f((bf16_t)qn[f][e]) * c);
                    const float b = rbf(bf2f((bf16_t)qn[2 + f][e]) * (g < 2 ? -s0 : s0));
                    qB[f][e] = (short)f2bf(a + b);
                }
        } else {
            qB[0] = qn[0]; qB[1] = qn[1];
        }
        if (h + 1 < he) fetch_q(qrow, h + 1);                  // lands under this head's passes

        // (every pass takes its own opaque copy of the wave's page count: a page guard is then one s_cmp + s_cbranch where it stands, instead of a lane
        //  mask per (pass, page) computed ahead of the head loop and parked in spilled SGPR pairs)
        // ---- pass A: raw scores into registers, four pages per guard (8 LDS reads in flight ahead of 16 matrix-core ops; a page past the wave's last
        //      one is computed on whatever its LDS slot holds and never looked at again)
        float sc[NP * 8];
        {
            const int n = opaque_u(npw);
#pragma unroll
            for (int pq = 0; pq < NP; pq += 4) {
                if (pq < n) {
#pragma unroll
                    for (int pg = pq; pg < pq + 4; ++pg) {
                        const bf16_t* kb = kres + pg * (kPage * 64);
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int r = u * 16 + l15;
                            f32x4 a = {0.f, 0.f, 0.f, 0.f};
                            a = mfma16(ld16<bf16x8>(kb + r * 64 + (((2 * g) ^ k_swz(r)) << 3)), qB[0], a);
                            a = mfma16(ld16<bf16x8>(kb + r * 64 + (((2 * g + 1) ^ k_swz(r)) << 3)), qB[1], a);
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) sc[pg * 8 + u * 4 + rr] = a[rr];
                        }
                    }
                }
            }
        }
        // ---- the diagonal page (the wave's last): causal mask (covers key >= S as qpos <= S - 1); then the row maximum
        float mx = kMasked;
        {
            const int n = opaque_u(npw);
            const int rel = opaque(qrel);
#pragma unroll
            for (int pg = 0; pg < NP; ++pg) {
                if (pg < n) {
                    if (pg + 1 == n) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) sc[pg * 8 + e] = (e >> 2) * 16 + g * 4 + (e & 3) > rel ? kMasked : sc[pg * 8 + e];
                    }
                    mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(sc[pg * 8], sc[pg * 8 + 1]), fmaxf(sc[pg * 8 + 2], sc[pg * 8 + 3])),
                                         fmaxf(fmaxf(sc[pg * 8 + 4], sc[pg * 8 + 5]), fmaxf(sc[pg * 8 + 6], sc[pg * 8 + 7]))));
                }
            }
        }
        mx = fmaxf(mx, shfl_xor(mx, 16));
        mx = fmaxf(mx, shfl_xor(mx, 32));                      // row maximum of query l15 (key 0 is never masked: finite)
        const float mr = rbf(mx);                              // rounding is monotone: bf16(max) = max of the bf16 scores
        // ---- pass B: e = exp((bf16(s) - bf16(max)) / 8) in place; the 2^-3 scaling of the eager contract is exact, so it rides in fexp_neg8's constants
        float sum0 = 0.f, sum1 = 0.f;
        {
            const int n = opaque_u(npw);
#pragma unroll
            for (int pg = 0; pg < NP; ++pg) {
                if (pg < n) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const float e0 = fexp_neg8(rbf(sc[pg * 8 + e]) - mr), e1 = fexp_neg8(rbf(sc[pg * 8 + e + 1]) - mr);
                        sc[pg * 8 + e] = e0;
                        sc[pg * 8 + e + 1] = e1;
                        sum0 += e0;
                        sum1 += e1;
                    }
                }
            }
        }
        float sum = sum0 + sum1;
        sum += shfl_xor(sum, 16);
        sum += shfl_xor(sum, 32);
        const float rs = frcp_refined(sum);
        // ---- pass C: P = bf16(e / sum), O += P V (page order)
        f32x4 oacc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) oacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const int n = opaque_u(npw);
            const int lim = opaque(srel);
#pragma unroll
            for (int pg = 0; pg < NP; ++pg) {
                if (pg < n) {
                    const bf16_t* vb = vres + pg * (64 * kPage);
                    bf16x8 vB[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int d = nt * 16 + l15;
                        vB[nt] = ld16<bf16x8>(vb + d * kPage + ((g ^ v_swz(d)) << 3));   // 16-B unit g of the row = keys 4g..+3, 16+4g..+3
                    }
                    bf16x8 pA;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pA[e] = (short)f2bf(fdiv_r(sc[pg * 8 + e], sum, rs));
                    if (vtail && pg + 1 == n) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if ((e < 4 ? g * 4 + e : 16 + g * 4 + e - 4) >= lim) vB[nt][e] = 0;
                    }
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) oacc[nt] = mfma16(pA, vB[nt], oacc[nt]);
                }
            }
        }
        // D: col = d (l15 of tile nt), row = query qw0 + g*4 + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qw0 + g * 4 + r;
            if (q < Sq) {
                if (p.out_fp8_inv > 0.f) {
                    unsigned char* o = (unsigned char*)p.out + (long)(base + q) * p.ld_out + h * 64 + l15;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) o[nt * 16] = f2fp8c(rbf(oacc[nt][r]) * p.out_fp8_inv);
                } else {
                    bf16_t* o = p.out + (long)(base + q) * p.ld_out + h * 64 + l15;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) o[nt * 16] = f2bf(oacc[nt][r]);
                }
            }
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// The next tier: queries at positions 512 .. 1023 (17 .. 32 pages deep; NeuTTS prompts with a long reference clip reach them).  Same decomposition
// (256 queries per workgroup, 16-query blocks dealt out from both ends of the tier, one head at a time, passes A / max / B / C), cut to what a CU holds:
//   * the K pages stay RESIDENT (32 x 4 KB = 128 KB); the V^T pages stream through a ring of two 4-page slots (32 KB: 160 KB in all) during pass C,
//     one workgroup barrier per four pages;
//   * a block's scores against up to 1024 keys stay in registers as PACKED bf16 pairs (they ARE bf16 values: 32 pages x 4 = 128 VGPRs), so e =
//     exp((s - max) / 8) is formed twice from them -- once for the denominator, once for P -- instead of being kept (256 VGPRs of fp32 would leave
//     one wave per SIMD and AGPR copies, the two-sweep kernel's own ailment).  QK^T still runs once, two waves per SIMD, no AGPR.
// Rounding points and summation orders are the resident kernel's (bf16 scores, exact scaling, fp32 max / exp / sum in page order per lane then across
// the four key groups, P rounded after the normalisation, PV in page order).  Positions >= 1024 stay with the two-sweep kernel.
constexpr int kPfDeepPages = 32;
constexpr int kPfDeepGroup = 4;      // V^T pages per ring slot
template <int NP>
NTTS_KERNEL(512) void attn_prefill_deep_kernel(AttnPrefillArgs p) {
    NTTS_SHARED bf16_t kres[NP * kPage * 64];                       // [page][32 keys][128 B], chunk c of key r at c ^ k_swz(r)
    NTTS_SHARED bf16_t vring[2 * kPfDeepGroup * 64 * kPage];        // [slot][page of the group][64 d][64 B], unit u of row d at u ^ v_swz(d)
    const int lane = lane_id(), w = wave_id();
    const int g = lane >> 4, l15 = lane & 15;
    const int tile = blockIdx.y, kvh = blockIdx.x;
    const int sq = p.meta.tile_seq[tile];
    const int S = p.meta.seq_len[sq];
    const int Sq = S < p.q_cap ? S : p.q_cap;                 // this launch's queries end here
    const int pos0 = p.meta.pos0[sq];
    const int qbase = pos0 > p.q_lo ? pos0 : p.q_lo;          // ... and start here
    const int base = p.meta.tok_base[sq] - pos0;              // packed row of absolute position q is base + q
    const int group = p.nh / p.nkv;
    const int hb = kvh * group + blockIdx.z * p.heads_per_wg;
    int he = hb + p.heads_per_wg;
    if (he > (kvh + 1) * group) he = (kvh + 1) * group;
    const int* bt = p.block_table + (long)p.meta.slot[sq] * p.max_pages;
    const int nb = (Sq - qbase + 15) >> 4;                    // blocks of the tier, dealt out as in attn_prefill_res_kernel
    const int nbp = (nb + 15) & ~15;
    const int k8 = p.meta.tile_q0[tile] * 8;
    const int blk_lo = k8 + w, blk_hi = nbp - 1 - k8 - w;
    int top_lo = k8 + 7;
    if (top_lo > nb - 1) top_lo = nb - 1;
    int top_hi = nbp - 1 - k8;
    if (top_hi > nb - 1) top_hi = nb - 1;
    auto pages_to = [&](int blk) { const int ql = qbase + blk * 16 + 15; return (ql < Sq - 1 ? ql : Sq - 1) / kPage + 1; };
    const int np_lo = pages_to(top_lo);
    const int npages = pages_to(top_hi > top_lo ? top_hi : top_lo);   // <= NP
    // ---- K: every page the workgroup can see; wave w brings 1 KB piece w & 3 (8 keys) of the pages of its parity
    for (int pg = w >> 2; pg < npages; pg += 2) {
        const long page = bt[pg];
        const int r = (w & 3) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ k_swz(r);
        glds16(p.kpool + ((page * p.nkv + kvh) * kPage + r) * 64 + c * 8, kres + pg * (kPage * 64) + (w & 3) * 512);
    }
    // V^T group gi (pages 4 gi .. 4 gi + 3 below `lim`) into ring slot gi & 1: 16 pieces of 1 KB, two per wave
    auto stage_v = [&](int gi, int lim) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int i = w * 2 + i2, pg = gi * kPfDeepGroup + (i >> 2);
            if (pg < lim) {
                const long page = bt[pg];
                const int d = (i & 3) * 16 + (lane >> 2);
                const int u = (lane & 3) ^ v_swz(d);
                glds16(p.vpool + (page * p.nkv + kvh) * 64 * kPage + d * kPage + u * 8,
                       vring + ((gi & 1) * kPfDeepGroup + (i >> 2)) * (64 * kPage) + (i & 3) * 512);
            }
        }
    };
    bf16x8 qn[4];                                              // next head's raw chunks: own 2, partner 2 (RoPE as in attn_prefill_res_kernel)
    auto fetch_q = [&](const bf16_t* qrow, int h) {
        const bf16_t* qr = qrow + h * 64;
        qn[0] = ld16<bf16x8>(qr + g * 16);
        qn[1] = ld16<bf16x8>(qr + g * 16 + 8);
        if (p.rope_cos) {
            qn[2] = ld16<bf16x8>(qr + (g ^ 2) * 16);
            qn[3] = ld16<bf16x8>(qr + (g ^ 2) * 16 + 8);
        }
    };
    wait_vmem();
    sync();

    constexpr float kMasked = -1.0e30f;
    const unsigned int kMaskedBits = (unsigned int)f2bf(kMasked);   // bf16(-1e30): finite, exp -> 0
    auto lo16 = [](unsigned int u) { return __builtin_bit_cast(float, u << 16); };
    auto hi16 = [](unsigned int u) { return __builtin_bit_cast(float, u & 0xffff0000u); };
    for (int half = 0; half < 2; ++half) {
    const int blk = half == 0 ? blk_lo : blk_hi;
    const int npmax = half == 0 ? np_lo : npages;             // workgroup-uniform: the pages pass C walks for this half
    if (p.only_last) {                                        // workgroup-uniform: a half without the prompt's last block has nothing to do
        const int b = nb - 1;
        if (half == 0 ? !(b >= k8 && b <= k8 + 7) : !(b <= nbp - 1 - k8 && b >= nbp - 8 - k8)) continue;
    }
    const bool live = blk < nb && blk >= 0 && !(p.only_last && blk != nb - 1);   // dead waves stage and hit the barriers
    const int qw0 = qbase + blk * 16;                         // first query of this wave's block
    int qpos = qw0 + l15;
    if (qpos > Sq - 1) qpos = Sq - 1;
    if (qpos < 0) qpos = 0;
    const int qlast_w = (qw0 + 15 < Sq ? qw0 + 15 : Sq - 1);
    const int npw = live ? qlast_w / kPage + 1 : 0;           // pages this block computes on; the last one holds its diagonal
    const bf16_t* qrow = p.qkv + (long)(base + qpos) * p.ld_qkv;
    const bf16_t* cr = p.rope_cos + (long)qpos * 32 + (g & 1) * 16;
    const bf16_t* sr = p.rope_sin + (long)qpos * 32 + (g & 1) * 16;
    fetch_q(qrow, hb);
    const bool vtail = npw * kPage > S;
    const int qrel = qpos - (npw - 1) * kPage;
    const int srel = S - (npw - 1) * kPage;
    for (int h = hb; h < he; ++h) {
        bf16x8 qB[2];
        if (p.rope_cos) {
            bf16x8 cv[2], sv[2];                               // (L1 / L2 hits, re-read per head: 16 registers less across the passes)
            cv[0] = ld16<bf16x8>(cr); cv[1] = ld16<bf16x8>(cr + 8);
            sv[0] = ld16<bf16x8>(sr); sv[1] = ld16<bf16x8>(sr + 8);
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float c = bf2f((bf16_t)cv[f][e]), s0 = bf2f((bf16_t)sv[f][e]);
                    const float a = rbf(bf2f((bf16_t)qn[f][e]) * c);
                    const float b = rbf(bf2f((bf16_t)qn[2 + f][e]) * (g < 2 ? -s0 : s0));
                    qB[f][e] = (short)f2bf(a + b);
                }
        } else {
            qB[0] = qn[0]; qB[1] = qn[1];
        }
        if (h + 1 < he) fetch_q(qrow, h + 1);                  // lands under this head's passes

        // ---- pass A: bf16 scores, packed two per register (word 0 = the even element), four pages per guard
        unsigned int sp[NP * 4];
        {
            const int n = opaque_u(npw);
#pragma unroll
            for (int pq = 0; pq < NP; pq += 4) {
                if (pq < n) {
#pragma unroll
                    for (int pg = pq; pg < pq + 4; ++pg) {
                        const bf16_t* kb = kres + pg * (kPage * 64);
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int r = u * 16 + l15;
                            f32x4 a = {0.f, 0.f, 0.f, 0.f};
                            a = mfma16(ld16<bf16x8>(kb + r * 64 + (((2 * g) ^ k_swz(r)) << 3)), qB[0], a);
                            a = mfma16(ld16<bf16x8>(kb + r * 64 + (((2 * g + 1) ^ k_swz(r)) << 3)), qB[1], a);
                            sp[pg * 4 + u * 2] = pack_bf2(a[0], a[1]);
                            sp[pg * 4 + u * 2 + 1] = pack_bf2(a[2], a[3]);
                        }
                    }
                }
            }
        }
        // ---- the diagonal page: causal mask; then the row maximum (of bf16 values: no further rounding)
        float mr = kMasked;
        {
            const int n = opaque_u(npw);
            const int rel = opaque(qrel);
#pragma unroll
            for (int pg = 0; pg < NP; ++pg) {
                if (pg < n) {
                    if (pg + 1 == n) {
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) {
                            const int k0 = (e2 >> 1) * 16 + g * 4 + (e2 & 1) * 2;   // key of word 0; word 1 = k0 + 1
                            unsigned int v = sp[pg * 4 + e2];
                            if (k0 > rel) v = (v & 0xffff0000u) | kMaskedBits;
                            if (k0 + 1 > rel) v = (v & 0xffffu) | (kMaskedBits << 16);
                            sp[pg * 4 + e2] = v;
                        }
                    }
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) mr = fmaxf(mr, fmaxf(lo16(sp[pg * 4 + e2]), hi16(sp[pg * 4 + e2])));
                }
            }
        }
        mr = fmaxf(mr, shfl_xor(mr, 16));
        mr = fmaxf(mr, shfl_xor(mr, 32));
        // ---- pass B: the denominator
        float sum0 = 0.f, sum1 = 0.f;
        {
            const int n = opaque_u(npw);
#pragma unroll
            for (int pg = 0; pg < NP; ++pg) {
                if (pg < n) {
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        sum0 += fexp_neg8(lo16(sp[pg * 4 + e2]) - mr);
                        sum1 += fexp_neg8(hi16(sp[pg * 4 + e2]) - mr);
                    }
                }
            }
        }
        float sum = sum0 + sum1;
        sum += shfl_xor(sum, 16);
        sum += shfl_xor(sum, 32);
        const float rs = frcp_refined(sum);
        // ---- pass C: P = bf16(e / sum), O += P V (page order); V^T through the ring, four pages per barrier.  The first barrier keeps this head's
        //      first group off a slot another wave may still be reading for the previous head
        f32x4 oacc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) oacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const int n = opaque_u(npw);
            const int lim = opaque(srel);
            const int ngrp = opaque_u((npmax + kPfDeepGroup - 1) / kPfDeepGroup);
            sync_keep_dma();
            stage_v(0, npmax);
#pragma unroll
            for (int gi = 0; gi < NP / kPfDeepGroup; ++gi) {
                if (gi < ngrp) {
                    wait_vmem();
                    sync_keep_dma();                           // group gi landed for every wave; everyone is done with the other slot
                    if (gi + 1 < ngrp) stage_v(gi + 1, npmax);
#pragma unroll
                    for (int pi = 0; pi < kPfDeepGroup; ++pi) {
                        const int pg = gi * kPfDeepGroup + pi;
                        if (pg < n) {
                            const bf16_t* vb = vring + ((gi & 1) * kPfDeepGroup + pi) * (64 * kPage);
                            bf16x8 vB[4];
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt) {
                                const int d = nt * 16 + l15;
                                vB[nt] = ld16<bf16x8>(vb + d * kPage + ((g ^ v_swz(d)) << 3));
                            }
                            bf16x8 pA;
#pragma unroll
                            for (int e2 = 0; e2 < 4; ++e2) {
                                const float e0 = fexp_neg8(lo16(sp[pg * 4 + e2]) - mr), e1 = fexp_neg8(hi16(sp[pg * 4 + e2]) - mr);
                                pA[e2 * 2] = (short)f2bf(fdiv_r(e0, sum, rs));
                                pA[e2 * 2 + 1] = (short)f2bf(fdiv_r(e1, sum, rs));
                            }
                            if (vtail && pg + 1 == n) {
#pragma unroll
                                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                                    for (int e = 0; e < 8; ++e)
                                        if ((e < 4 ? g * 4 + e : 16 + g * 4 + e - 4) >= lim) vB[nt][e] = 0;
                            }
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt) oacc[nt] = mfma16(pA, vB[nt], oacc[nt]);
                        }
                    }
                }
            }
        }
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = qw0 + g * 4 + r;
                if (q < Sq) {
                    if (p.out_fp8_inv > 0.f) {
                        unsigned char* o = (unsigned char*)p.out + (long)(base + q) * p.ld_out + h * 64 + l15;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) o[nt * 16] = f2fp8c(rbf(oacc[nt][r]) * p.out_fp8_inv);
                    } else {
                        bf16_t* o = p.out + (long)(base + q) * p.ld_out + h * 64 + l15;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) o[nt * 16] = f2bf(oacc[nt][r]);
                    }
                }
            }
        }
    }
    }
}

inline void attn_prefill_deep_launch(AttnPrefillArgs p, int n_tiles, int q_lo, int q_cap, bool only_last, hipStream_t s) {
    const int group = p.nh / p.nkv;
    const long pairs = (long)p.nkv * n_tiles;
    int hps = group;
    if (pairs * group <= 256) hps = 1;
    else if (pairs * ((group + 1) / 2) <= 256) hps = 2;
    else if (pairs * ((group + 3) / 4) <= 256) hps = 4;
    p.q_lo = q_lo;
    p.q_cap = q_cap;
    p.heads_per_wg = hps;
    p.only_last = only_last ? 1 : 0;
    NTTS_LAUNCH((attn_prefill_deep_kernel<kPfDeepPages>), dim3(p.nkv, n_tiles, (group + hps - 1) / hps), dim3(512), s, p);
}

// host launcher of the resident kernel: short passes spread the group's heads over grid z while the grid stays within the CUs (as below)
inline void attn_prefill_res_launch(AttnPrefillArgs p, int n_tiles, int q_cap, bool only_last, hipStream_t s) {
    const int group = p.nh / p.nkv;
    const long pairs = (long)p.nkv * n_tiles;
    int hps = group;
    if (pairs * group <= 256) hps = 1;
    else if (pairs * ((group + 1) / 2) <= 256) hps = 2;
    else if (pairs * ((group + 3) / 4) <= 256) hps = 4;
    p.q_cap = q_cap;
    p.heads_per_wg = hps;
    p.only_last = only_last ? 1 : 0;
    NTTS_LAUNCH((attn_prefill_res_kernel<kPfResPages>), dim3(p.nkv, n_tiles, (group + hps - 1) / hps), dim3(512), s, p);
}

// host launcher: GH = 7 covers NeuTTS-Air's group in one pass; other group sizes run ceil(group / GH) passes
// A SHORT pass (one to a few prompts) is a handful of (tile, kv-head) pairs: a 500-token prompt is 16 workgroups, each walking its pages for all 7
// heads -- 54 us per launch, a third of the batch-1 prompt pass.  The heads of a group are independent, so while the grid stays within the CUs they are
// spread over more workgroups (GH = 1 or 2 heads each, grid z = passes over the group; every pass stages the pages again, from L2): same arithmetic per head.
inline void attn_prefill_launch(const AttnPrefillArgs& p, int n_tiles, hipStream_t s) {
    const int group = p.nh / p.nkv;
    const long pairs = (long)p.nkv * n_tiles;
    if (pairs * group <= 256) { NTTS_LAUNCH((attn_prefill_gqa_kernel<1>), dim3(p.nkv, n_tiles, group), dim3(256), s, p); return; }
    if (pairs * ((group + 1) / 2) <= 256 && group > 2) { NTTS_LAUNCH((attn_prefill_gqa_kernel<2>), dim3(p.nkv, n_tiles, (group + 1) / 2), dim3(256), s, p); return; }
    if (group > 4 && pairs * ((group + 3) / 4) <= 256) {   // (5 / 6 / 8 prompts of 500 tokens: prompt pass 5.81 / 5.95 / 7.06 -> 5.18 / 5.34 / 6.47 ms)
        NTTS_LAUNCH((attn_prefill_gqa_kernel<4>), dim3(p.nkv, n_tiles, (group + 3) / 4), dim3(256), s, p);
        return;
    }
    if (group <= 4) NTTS_LAUNCH((attn_prefill_gqa_kernel<4>), dim3(p.nkv, n_tiles, 1), dim3(256), s, p);
    else NTTS_LAUNCH((attn_prefill_gqa_kernel<7>), dim3(p.nkv, n_tiles, (group + 6) / 7), dim3(256), s, p);
}

// head_dim 128 (round 6): 1 / 2 / 4 heads of the group per workgroup (registers: 4 heads x 8 PV tiles are 128 accumulator registers)
inline void attn_prefill_launch_hd128(const AttnPrefillArgs& p, int n_tiles, hipStream_t s) {
    const int group = p.nh / p.nkv;
    const long pairs = (long)p.nkv * n_tiles;
    if (group == 1 || pairs * group <= 256) { NTTS_LAUNCH((attn_prefill_gqa_kernel<1, 128>), dim3(p.nkv, n_tiles, group), dim3(256), s, p); return; }
    if (group == 2 || pairs * ((group + 1) / 2) <= 256) { NTTS_LAUNCH((attn_prefill_gqa_kernel<2, 128>), dim3(p.nkv, n_tiles, (group + 1) / 2), dim3(256), s, p); return; }
    NTTS_LAUNCH((attn_prefill_gqa_kernel<4, 128>), dim3(p.nkv, n_tiles, (group + 3) / 4), dim3(256), s, p);
}

}  // namespace ntts
