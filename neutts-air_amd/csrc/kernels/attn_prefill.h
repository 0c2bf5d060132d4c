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
#include "attn_decode.h"

namespace ntts {

struct PrefillMeta {           // one entry per prompt of the packed batch (device arrays)
    const int* tok_base;       // [n] first packed row of prompt i
    const int* seq_len;        // [n]
    const int* slot;           // [n] decode slot -> block_table row
    const int* tok_seq;        // [T] prompt index of packed row t
    const int* tile_seq;       // [n_tiles] work list of 64-row query tiles
    const int* tile_q0;        // [n_tiles]
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
};

NTTS_KERNEL(256) void rope_kv_write_kernel(RopeWriteArgs p) {
    const int t = blockIdx.x;
    const int sq = p.meta.tok_seq[t];
    const int pos = t - p.meta.tok_base[sq];
    const int* bt = p.block_table + (long)p.meta.slot[sq] * p.max_pages;
    const long pg = bt[pos / kPage];
    const int slot = pos % kPage;
    bf16_t* row = p.qkv + (long)t * p.ld_qkv;
    const int npairs = (p.nh + 2 * p.nkv) * 32;
    for (int x = threadIdx.x; x < npairs; x += 256) {
        const int hh = x >> 5, i = x & 31;
        const float c = bf2f(p.rope_cos[(long)pos * 32 + i]), s = bf2f(p.rope_sin[(long)pos * 32 + i]);
        bf16_t* h = row + hh * 64;
        if (hh < p.nh) {
            float o1, o2;
            rope_pair(bf2f(h[i]), bf2f(h[i + 32]), c, s, o1, o2);
            h[i] = f2bf(o1);
            h[i + 32] = f2bf(o2);
        } else if (hh < p.nh + p.nkv) {
            const int kvh = hh - p.nh;
            float o1, o2;
            rope_pair(bf2f(h[i]), bf2f(h[i + 32]), c, s, o1, o2);
            bf16_t* kd = p.kpool + ((pg * p.nkv + kvh) * kPage + slot) * 64;
            kd[i] = f2bf(o1);
            kd[i + 32] = f2bf(o2);
        } else {
            const int kvh = hh - p.nh - p.nkv;
            bf16_t* vd = p.vpool + (pg * p.nkv + kvh) * 64 * kPage + slot;
            vd[(long)i * kPage] = h[i];
            vd[(long)(i + 32) * kPage] = h[i + 32];
        }
    }
}

struct AttnPrefillArgs {
    const bf16_t* qkv;         // q already rotated
    long ld_qkv;
    bf16_t* out;               // [T][nh*64]
    long ld_out;
    const bf16_t* kpool;
    const bf16_t* vpool;
    const int* block_table;
    int max_pages;
    PrefillMeta meta;
    int nh, nkv;
};

// grid (n_tiles, nh); 4 waves x 16 query rows
NTTS_KERNEL(256) void attn_prefill_kernel(AttnPrefillArgs p) {
    const int lane = lane_id(), w = wave_id();
    const int g = lane >> 4, l15 = lane & 15;
    const int sq = p.meta.tile_seq[blockIdx.x];
    const int S = p.meta.seq_len[sq];
    const int base = p.meta.tok_base[sq];
    const int h = blockIdx.y;
    const int kvh = h / (p.nh / p.nkv);
    const int* bt = p.block_table + (long)p.meta.slot[sq] * p.max_pages;
    const int qw0 = p.meta.tile_q0[blockIdx.x] + w * 16;   // first query position of this wave
    if (qw0 >= S) return;                                  // wave-uniform; no barriers in this kernel
    int qpos = qw0 + l15;                                  // this lane's query (B-operand column)
    const bool qok = qpos < S;
    if (!qok) qpos = S - 1;
    const int qlast = (qw0 + 15 < S ? qw0 + 15 : S - 1);
    const int npages = qlast / kPage + 1;

    const bf16_t* qr = p.qkv + (long)(base + qpos) * p.ld_qkv + h * 64 + g * 16;
    bf16x8 qB[2];
    qB[0] = ld16<bf16x8>(qr);
    qB[1] = ld16<bf16x8>(qr + 8);

    auto scores = [&](int pg, float (&s)[8]) {
        const bf16_t* kp = p.kpool + ((long)bt[pg] * p.nkv + kvh) * kPage * 64;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16_t* kr = kp + (u * 16 + l15) * 64 + g * 16;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = mfma16(ld16<bf16x8>(kr), qB[0], a);
            a = mfma16(ld16<bf16x8>(kr + 8), qB[1], a);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = pg * kPage + u * 16 + g * 4 + r;
                float v = rbf(rbf(a[r]) * 0.125f);
                if (key > qpos) v = -INFINITY;            // causal (covers key >= S as qpos <= S-1)
                s[u * 4 + r] = v;
            }
        }
    };

    // ---- sweep 1: row max and softmax denominator (online), lane-local then across the 4 key groups
    float m = -INFINITY, sum = 0.f;
    for (int pg = 0; pg < npages; ++pg) {
        float s[8];
        scores(pg, s);
        float tm = s[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) tm = fmaxf(tm, s[e]);
        const float mn = fmaxf(m, tm);
        if (mn != -INFINITY) {
            float add = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) add += fexp(s[e] - mn);
            sum = sum * fexp(m - mn) + add;
            m = mn;
        }
    }
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {
        const float om = shfl_xor(m, sh), os = shfl_xor(sum, sh);
        const float mn = fmaxf(m, om);
        if (mn != -INFINITY) {
            sum = (m == -INFINITY ? 0.f : sum * fexp(m - mn)) + (om == -INFINITY ? 0.f : os * fexp(om - mn));
            m = mn;
        }
    }

    // ---- sweep 2: P = bf16(exp(s - m) / sum), O += P V
    f32x4 oacc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) oacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int pg = 0; pg < npages; ++pg) {
        float s[8];
        scores(pg, s);
        bf16x8 pA;
#pragma unroll
        for (int e = 0; e < 8; ++e) pA[e] = (short)f2bf(fexp(s[e] - m) / sum);
        const bf16_t* vp = p.vpool + ((long)bt[pg] * p.nkv + kvh) * 64 * kPage;
        const bool tail = (pg + 1) * kPage > S;            // page holds slots past the prompt: mask them
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const bf16_t* vr = vp + (nt * 16 + l15) * kPage + g * 4;
            const bf16x4 v0 = ld16<bf16x4>(vr), v1 = ld16<bf16x4>(vr + 16);
            bf16x8 vB;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vB[e] = v0[e];
                vB[4 + e] = v1[e];
            }
            if (tail) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int key = pg * kPage + (e < 4 ? g * 4 + e : 16 + g * 4 + e - 4);
                    if (key >= S) vB[e] = 0;
                }
            }
            oacc[nt] = mfma16(pA, vB, oacc[nt]);
        }
    }
    // D: col = d (l15 of tile nt), row = query qw0 + g*4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = qw0 + g * 4 + r;
        if (q < S) {
            bf16_t* o = p.out + (long)(base + q) * p.ld_out + h * 64 + l15;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) o[nt * 16] = f2bf(oacc[nt][r]);
        }
    }
}

}  // namespace ntts
