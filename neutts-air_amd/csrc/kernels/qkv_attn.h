// qkv_attn.h -- the large-batch decode step's QKV projection AND its attention in ONE launch (round 4).
//
// Same arithmetic as qkv_rope.h + attn_decode.h (they stay: every other batch size / model variant takes them; this file restates their
// bodies step for step and cites them), replacing
//   q/k/v = nn.Linear(x), RoPE, cache.update           hf:models/qwen2/modeling_qwen2.py:206-214, hf:cache_utils.py:127-146
//   eager_attention_forward (q_len = 1)                  hf:models/qwen2/modeling_qwen2.py:150-172
// Why one launch.  A decode step's attention reads Sum_b L_b x 512 B of K / V^T per layer (83 MB at batch 256, context 625): none of it
// depends on the current step -- only q (and the new k / v row) does.  As two launches the 512 attention workgroups cannot request a
// byte before the QKV kernel has finished (6.4 us during which HBM idles) and a kernel boundary has passed (1.6 us), then stream for
// 13 us.  Here the QKV tiles are the FIRST workgroups of the grid (the PRODUCERS: qkv_rope_kernel's body) and the attention workgroups
// (the CONSUMERS) follow: a consumer requests ALL its K pages into registers at entry (a wave's pages w, w + 4, ...: RP x 16 VGPRs) and
// its first V^T pages, then waits for its own 288 values -- 7 q heads, the new k row, the new v row -- which the producers hand over
// as 8-byte {tag, 2 x bf16} GRANULES written by one write-through (sc1) store each: the data is the flag (cdna_hip_programming.md
// Guideline 16, recipe R2: no fence, no separate flag; the tag is unique per (step, layer), so nothing is zeroed).  One wave sweeps the
// granules (relaxed agent-scope loads) into LDS, the other waves wait at an LDS-only barrier with their K loads still in flight.
// Then: scores from the resident K (the new k row patched into its page's fragment), softmax statistics merged as in attn_decode.h,
// P V from the prefetched V^T pages and then the rest (requested in one go when the scores are done: the K registers are free).
// Results are bit-identical to the two-launch path: same operands, same order (tests: every decode golden runs through here at batch 256).
//
// Correctness does not depend on dispatch order, timing or XCD placement: a consumer that starts before its producers simply sweeps
// longer (bounded: the producers are the lowest block ids and need no resource a consumer holds -- see qkv_attn_launch); the K page of
// the current position may be read before or after the producer's append: that row is always overwritten from the granules.
#pragma once
#include <ntts/dev.h>
#include "attn_decode.h"
#include "qkv_rope.h"

namespace ntts {

struct QkvAttnArgs {
    QkvRopeArgs q;                 // producer side (q.q_out unused: q | k | v leave as granules; the K append into the page stays)
    AttnDecodeArgs a;              // consumer side (a.qkv unused)
    unsigned long long* hand;      // [M][N / 2] granules: {tag << 32 | bf16 pair} of features 2 j, 2 j + 1 of the q | k | v row
    const unsigned int* step_ctr;  // device counter bumped once per decode step (step_meta_row): tag = step * 32 + layer + 1
    int layer;
    int n_prod;                    // producer workgroups (a multiple of 8: keeps the consumers' workgroup -> XCD phase)
    int n_prod_real;               // ... of which this many hold a tile (the rest return)
    int batch;
    int x_sleep, x_defer;          // EXPERIMENT knobs: pauses of ~0.43 us before the first sweep; 1 = request K / V^T only after the hand-over
    unsigned int* cu_busy;         // [128] bitmap by cu_key(): CUs that currently run a producer (speed hint: a consumer on such a CU requests its pages after the hand-over)
    unsigned long long* tl;        // diagnostics: [workgroups][16] phase timestamps of wave 0 (100 MHz ticks), null in the product path
    unsigned int* err;             // set to 1 if a consumer gave up waiting (a bug, not a state: the launch's results are then invalid)
};

// ---- PRODUCER: qkv_rope_kernel<NS, F8, KS>'s body with granule outputs -----------------------------------------------------------
template <int NS, bool F8, int KS>
NTTS_D void qa_producer(const QkvAttnArgs& A, bf16_t* lds, int bid, unsigned int tag) {
    const QkvRopeArgs& p = A.q;
    constexpr int ESZ = F8 ? 1 : 2;
    constexpr int BM = 32, BN = 64, ROWS = BM + BN;
    constexpr int NINST = ROWS / 8;
    constexpr int PER_WAVE = NINST / 2;
    static_assert(KS == 2, "granule epilogue is written for two K slices (8 features per lane)");
    typedef f32x4 (*XchT)[2][4][64];
    XchT xch = (XchT)lds;
    auto swz = [](int rho) { return (rho >> 1) & 7; };
    const int lane = lane_id(), wave = wave_id();
    const int wm = wave & 1, kh = wave >> 1;
    const int g = lane >> 4, l15 = lane & 15;
    const int nblocks = p.N >> 6;
    int mb, nb;
    if (p.xcd_mpx > 0) {
        const int xcd = bid & 7, j = bid >> 3;
        mb = xcd * p.xcd_mpx + j % p.xcd_mpx;
        nb = j / p.xcd_mpx;
    } else {
        const int mblocks = (p.M + BM - 1) / BM;
        mb = bid % mblocks;
        nb = bid / mblocks;
    }
    if (nb >= nblocks) return;
    auto mark = [&](int slot) { if (A.tl && wave == 0 && lane == 0) A.tl[(long)blockIdx.x * 16 + slot] = now_ticks(); };
    mark(0);
    const int ck = cu_key();
    if (A.cu_busy && threadIdx.x == 0) atomic_or_global(A.cu_busy + (ck >> 5), 1u << (ck & 31));
    const int m0 = mb * BM, n0 = nb * BN;
    const int ktiles = F8 ? p.K >> 7 : p.K >> 6;
    int nk = ktiles - kh * p.kps;
    if (nk > p.kps) nk = p.kps;
    if (nk < 0) nk = 0;
    constexpr int NJ = 4 / KS;
    const int m = m0 + wm * 16 + l15;
    const int mc = m < p.M ? m : p.M - 1;
    const int e0 = kh * NJ * 4;
    const int nb16 = n0 + g * 16, nb16p = n0 + (g ^ 2) * 16;
    const u32x4 meta = ld16<u32x4>(p.meta + (long)mc * 4);
    bf16x4 cs[NJ], sn[NJ], bs[NJ], bsp[NJ];
    f32x4 wsc[NJ], wscp[NJ];
    {
        const bf16_t* rr = p.rope_rows + (long)mc * 64 + (g & 1) * 16 + e0;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            cs[jj] = *(const bf16x4*)(rr + jj * 4);
            sn[jj] = *(const bf16x4*)(rr + 32 + jj * 4);
            bs[jj] = *(const bf16x4*)(p.bias + nb16 + e0 + jj * 4);
            bsp[jj] = *(const bf16x4*)(p.bias + nb16p + e0 + jj * 4);
            if constexpr (F8) {
                wsc[jj] = ld16<f32x4>(p.wscale + nb16 + e0 + jj * 4);
                wscp[jj] = ld16<f32x4>(p.wscale + nb16p + e0 + jj * 4);
            }
        }
    }
    const char* src[PER_WAVE];
    bool is_w[PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int inst = wm + 2 * i;
        const int rho = inst * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz(rho);
        is_w[i] = inst * 8 >= BM;
        if (rho < BM) {
            int mr = m0 + rho;
            if (mr > p.M - 1) mr = p.M - 1;
            src[i] = (const char*)p.X + (long)mr * p.ldx * ESZ + c * 16;
        } else {
            const int q = rho - BM;
            const int j = (q >> 4) & 3, i16 = q & 15;
            const int n = n0 + (i16 >> 2) * 16 + j * 4 + (i16 & 3);
            src[i] = (const char*)p.W + (long)(n >> 6) * 64 * p.K * ESZ + (n & 63) * 128 + c * 16;
        }
    }
    const int kt0 = kh * p.kps;
    auto stage = [&](int kt, int slot) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int inst = wm + 2 * i;
            const char* gp = src[i] + (long)(kt0 + kt) * (is_w[i] ? 8192 : 128);
            glds16(gp, lds + ((slot * KS + kh) * ROWS) * 64 + inst * 512);
        }
    };
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int xrho = wm * 16 + l15;
    const int xoff = xrho * 64, xsw = swz(xrho);
    int woff[4], wsw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rho = BM + j * 16 + l15;
        woff[j] = rho * 64;
        wsw[j] = swz(rho);
    }
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) stage(s, s);
    int buf = 0;
    for (int kt = 0; kt < p.kps; ++kt) {
        if (kt < nk) {
            if (kt + NS - 2 < nk) wait_vmem_le<(NS - 2) * PER_WAVE>(); else wait_vmem();
        }
        sync_keep_dma();
        if (kt + NS - 1 < nk) stage(kt + NS - 1, buf == 0 ? NS - 1 : buf - 1);
        const bf16_t* base = lds + ((buf * KS + kh) * ROWS) * 64;
        buf = buf + 1 == NS ? 0 : buf + 1;
        if (kt >= nk) continue;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = ks * 4 + g;
            const bf16x8 xb = ld16<bf16x8>(base + xoff + ((c ^ xsw) << 3));
            bf16x8 wa[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wa[j] = ld16<bf16x8>(base + woff[j] + ((c ^ wsw[j]) << 3));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (F8) {
                    const i64x2 w2 = __builtin_bit_cast(i64x2, wa[j]), x2 = __builtin_bit_cast(i64x2, xb);
                    acc[j] = mfma16_fp8(w2[0], x2[0], acc[j]);
                    acc[j] = mfma16_fp8(w2[1], x2[1], acc[j]);
                } else {
                    acc[j] = mfma16(wa[j], xb, acc[j]);
                }
            }
        }
    }
    mark(1);
    sync();
#pragma unroll
    for (int j = 0; j < 4; ++j) xch[kh][wm][j][lane] = acc[j];
    sync();
    const int hd = n0 >> 6;
    const int st = (int)meta[0], page = (int)meta[2], slot = (int)meta[3];
    const bool mok = m < p.M, run = mok && st == 1;
    const bool rot = hd < p.nh + p.nkv;
    alignas(16) bf16_t out[NJ * 4];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = kh * NJ + jj;
        f32x4 so = xch[0][wm][j][lane], sp = xch[0][wm][j][lane ^ 32];
        if constexpr (F8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { so[r] *= p.xscale * wsc[jj][r]; sp[r] *= p.xscale * wscp[jj][r]; }
        }
#pragma unroll
        for (int q = 1; q < KS; ++q) {
            const f32x4 o = xch[q][wm][j][lane], op = xch[q][wm][j][lane ^ 32];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (F8) { so[r] += o[r] * (p.xscale * wsc[jj][r]); sp[r] += op[r] * (p.xscale * wscp[jj][r]); }
                else { so[r] += o[r]; sp[r] += op[r]; }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bf16_t vo = f2bf(so[r] + bf2f((bf16_t)bs[jj][r])), vp = f2bf(sp[r] + bf2f((bf16_t)bsp[jj][r]));
            const float xo = bf2f(vo), xp = bf2f(vp), c = bf2f((bf16_t)cs[jj][r]), sv = bf2f((bf16_t)sn[jj][r]);
            out[jj * 4 + r] = rot ? f2bf(rbf(xo * c) + (g < 2 ? rbf(-xp * sv) : rbf(xp * sv))) : vo;
        }
    }
    // the k head also goes into its page, for the steps to come (plain store: visible at the next kernel boundary)
    if (hd >= p.nh && hd < p.nh + p.nkv && run)
        *(u32x4*)(p.kpool + (((long)page * p.nkv + (hd - p.nh)) * kPage + slot) * 64 + g * 16 + e0) = *(u32x4*)&out[0];
    // every value leaves as a granule: features nb16 + e0 + 2 t, + 2 t + 1 of token m -> hand[m][(nb16 + e0) / 2 + t], ONE 8-byte
    // write-through store each (the consumer reads tag and data with one load: no tearing, no fence)
    if (mok) {
        unsigned long long* hg = A.hand + (long)m * (p.N >> 1) + ((nb16 + e0) >> 1);
#pragma unroll
        for (int t = 0; t < NJ * 2; ++t)
            granule_store(hg + t, ((unsigned long long)tag << 32) | ((unsigned long long)out[2 * t + 1] << 16) | (unsigned long long)out[2 * t]);
    }
    if (A.cu_busy) { sync(); if (threadIdx.x == 0) atomic_and_global(A.cu_busy + (ck >> 5), ~(1u << (ck & 31))); }
    if (A.tl) { wait_vmem(); mark(2); }
}

// ---- CONSUMER: attn_decode_kernel<1, false, 1, 4, LMAX, 1>'s arithmetic with the K pages resident in registers -----------------------
// RP = K pages per wave held in registers (contexts up to RP * 4 * 32 tokens), PV = V^T pages per wave requested at entry.
template <int RP, int PV, int LMAX>
NTTS_D void qa_consumer(const QkvAttnArgs& A, char* smem, int cid, unsigned int tag) {
    const AttnDecodeArgs& p = A.a;
    constexpr int NW = 4, NT = 256;
    static_assert(LMAX <= 1024 && RP * NW * kPage >= LMAX && PV <= RP, "resident pages cover the longest context");
    // LDS carve (all 16-byte aligned): rounded scores | new v row | per-wave softmax partials | per-wave output partials | handed-over row
    bf16_t (*sc)[LMAX + 16] = (bf16_t (*)[LMAX + 16])smem;
    bf16_t* vnew = (bf16_t*)(smem + 8 * (LMAX + 16) * 2);
    float (*wred)[kGroupMax] = (float (*)[kGroupMax])(vnew + 64);
    float (*wsum)[kGroupMax] = wred + NW;
    float (*ored)[kGroupMax][64] = (float (*)[kGroupMax][64])(wsum + NW);
    bf16_t* hq = (bf16_t*)(ored + NW);                 // [group * 64] rotated q heads | [64] rotated k row   (<= 576 + 64 values)
    const int b = p.xcd_rows ? xcd_row(cid % A.batch, p.xcd_rows) : cid % A.batch;
    const int kvh = cid / A.batch;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int g = lane >> 4, l15 = lane & 15;
    const int group = p.nh / p.nkv;
    const int* bt = p.block_table + (long)b * p.max_pages;
    const int st = p.state[b];
    const int P = p.pos[b];
    if (st != 1) return;   // block-uniform
    auto mark = [&](int slot) { if (A.tl && w == 0 && lane == 0) A.tl[(long)blockIdx.x * 16 + slot] = now_ticks(); };
    mark(0);
    const int L = P + 1;
    const int npages = (L + kPage - 1) / kPage;
    const int last_page = npages - 1;
    auto load_k_at = [&](long page, bf16x8 (&k)[2][2]) {
        const bf16_t* kp = p.kpool + (page * p.nkv + kvh) * kPage * 64;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16_t* kr = kp + (u * 16 + l15) * 64 + g * 16;
            k[u][0] = ld16<bf16x8>(kr); k[u][1] = ld16<bf16x8>(kr + 8);
        }
    };
    auto load_v_at = [&](long page, bf16x8 (&v)[4]) {
        const bf16_t* vp = p.vpool + (page * p.nkv + kvh) * 64 * kPage;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) v[nt] = ld16<bf16x8>(vp + (nt * 16 + l15) * kPage + g * 8);
    };
    // ---- everything that does not depend on this step: the block-table entries, then ALL K pages of this wave, then its first V^T pages
    int pgi[RP];
#pragma unroll
    for (int j = 0; j < RP; ++j) pgi[j] = bt[w + NW * j < npages ? w + NW * j : 0];
    bf16x8 kres[RP][2][2];
    bf16x8 vres[PV][4];
    // a producer on this CU?  Its LDS-DMA stream would queue behind this workgroup's 112 KB of page requests in the CU's load path (measured: the
    // QKV tiles then take 12-14 us instead of 6.4 and every consumer waits for them): such a consumer requests its pages after the hand-over
    bool defer = A.x_defer != 0;
    if (A.cu_busy && !defer) {
        const int ck = cu_key();
        defer = (relaxed_load_u32(A.cu_busy + (ck >> 5)) >> (ck & 31)) & 1u;
    }
    if (A.tl && w == 0 && lane == 0) A.tl[(long)blockIdx.x * 16 + 7] = defer ? 1 : 0;
    if (!defer) {
#pragma unroll
    for (int j = 0; j < RP; ++j)
        if (w + NW * j < npages) load_k_at(pgi[j], kres[j]);
#pragma unroll
    for (int j = 0; j < PV; ++j)
        if (w + NW * j < npages) load_v_at(pgi[j], vres[j]);
    }
    const long vpage_new = bt[P / kPage];
    mark(1);
    // ---- the hand-over: wave 0 sweeps this consumer's granules -- q heads of the group, the k row, the v row -- until every tag is
    //      this (step, layer)'s; relaxed agent-scope 8-byte loads, all of a pass in flight together, a short sleep between passes
    if (w == 0) {
        const int nq = group * 32;                           // q granules; then 32 k granules, 32 v granules
        const unsigned long long* hrow = A.hand + (long)b * ((p.nh + 2 * p.nkv) * 32);
        constexpr int NG = (kGroupMax * 32 + 64 + 63) / 64;  // granules per lane, upper bound
        unsigned int val[NG];
        for (int z = 0; z < A.x_sleep; ++z) __builtin_amdgcn_s_sleep(16);
        for (unsigned int spins = 0;; ++spins) {
            bool ok = true;
#pragma unroll
            for (int t = 0; t < NG; ++t) {
                const int gi = lane + 64 * t;
                if (gi < nq + 64) {
                    const long col = gi < nq ? (long)kvh * group * 32 + gi : gi < nq + 32 ? (long)(p.nh + kvh) * 32 + (gi - nq) : (long)(p.nh + p.nkv + kvh) * 32 + (gi - nq - 32);
                    const unsigned long long x = granule_load(hrow + col);
                    val[t] = (unsigned int)x;
                    ok = ok && (unsigned int)(x >> 32) == tag;
                }
            }
            if (ballot(!ok) == 0ull) break;
            if (spins > (1u << 22)) { if (lane == 0) *A.err = 1u; break; }   // (seconds: a producer that never ran -- never seen; fail loudly upstream)
            spin_pause();
        }
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            const int gi = lane + 64 * t;
            if (gi < nq + 32) *(unsigned int*)(hq + 2 * gi) = val[t];                  // q heads, then the k row, contiguous
            else if (gi < nq + 64) *(unsigned int*)(vnew + 2 * (gi - nq - 32)) = val[t];
        }
    }
    mark(2);
    sync_keep_dma();   // the handed-over row is in LDS; the other waves' K / V^T requests stayed in flight
    if (defer) {
#pragma unroll
    for (int j = 0; j < RP; ++j)
        if (w + NW * j < npages) load_k_at(pgi[j], kres[j]);
#pragma unroll
    for (int j = 0; j < PV; ++j)
        if (w + NW * j < npages) load_v_at(pgi[j], vres[j]);
    }
    bf16x8 qB[2];
    {
        const bf16_t* qrow = hq + (l15 < group ? l15 : 0) * 64 + g * 16;
        qB[0] = ld16<bf16x8>(qrow);
        qB[1] = ld16<bf16x8>(qrow + 8);
    }
    if (l15 >= group) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { qB[0][e] = 0; qB[1][e] = 0; }
    }
    // ---- pass 1: scores of the resident pages (attn_decode.h pass 1); the page of position P gets its new row from the hand-over
    constexpr float kMasked = -1.0e30f;
    float lmax = kMasked, lsum = 0.f;
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        const int pg = w + NW * j;
        if (pg < npages) {
            if (pg == last_page && l15 == (P & 15)) {
                const bf16_t* kn = hq + group * 64 + g * 16;
                const int u = (P >> 4) & 1;
                const bf16x8 k0 = ld16<bf16x8>(kn), k1 = ld16<bf16x8>(kn + 8);
                if (u == 0) { kres[j][0][0] = k0; kres[j][0][1] = k1; } else { kres[j][1][0] = k0; kres[j][1][1] = k1; }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                a = mfma16(kres[j][u][0], qB[0], a);
                a = mfma16(kres[j][u][1], qB[1], a);
                const int key0 = pg * kPage + u * 16 + g * 4;
                bf16x4 sv;
                float s4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = rbf(a[r]) * 0.125f;
                    if (key0 + r >= L) s = kMasked;
                    s4[r] = s;
                    sv[r] = (short)f2bf(s);
                }
                if (l15 < kGroupMax) *(bf16x4*)&sc[l15][key0] = sv;
                const float mn = fmaxf(lmax, fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3])));
                lsum = lsum * fexp_neg(lmax - mn) + fexp_neg(s4[0] - mn) + fexp_neg(s4[1] - mn) + fexp_neg(s4[2] - mn) + fexp_neg(s4[3] - mn);
                lmax = mn;
            }
        }
    }
    mark(3);
    // the K registers are free: request every V^T page that was not requested at entry
    bf16x8 vrest[RP - PV > 0 ? RP - PV : 1][4];
#pragma unroll
    for (int j = PV; j < RP; ++j)
        if (w + NW * j < npages) load_v_at(pgi[j], vrest[j - PV]);
    if (tid < 64) p.vpool[(vpage_new * p.nkv + kvh) * 64 * kPage + (long)tid * kPage + v_slot(P % kPage)] = vnew[tid];   // the V half of cache.update
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {
        const float om = shfl_xor(lmax, sh), os = shfl_xor(lsum, sh);
        const float mn = fmaxf(lmax, om);
        lsum = lsum * fexp_neg(lmax - mn) + os * fexp_neg(om - mn);
        lmax = mn;
    }
    if (g == 0 && l15 < kGroupMax) { wred[w][l15] = lmax; wsum[w][l15] = lsum; }
    sync_keep_dma();
    float m_l = kMasked, sum_l = 1.f;
    if (l15 < kGroupMax) {
        m_l = wred[0][l15];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) m_l = fmaxf(m_l, wred[ww][l15]);
        sum_l = wsum[0][l15] * fexp_neg(wred[0][l15] - m_l);
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) sum_l += wsum[ww][l15] * fexp_neg(wred[ww][l15] - m_l);
    }
    const float rs_l = frcp_refined(sum_l);
    mark(4);
    // ---- pass 2: O = P V (attn_decode.h pass 2), pages in the same ascending order per wave
    f32x4 oacc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) oacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        const int pg = w + NW * j;
        if (pg < npages) {
            bf16x8 vc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) vc[nt] = j < PV ? vres[j < PV ? j : 0][nt] : vrest[j >= PV ? j - PV : 0][nt];
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
            if (pg == last_page) {
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
    if (A.tl && w == 0 && lane == 0) A.tl[(long)blockIdx.x * 16 + 5] = now_ticks() + (oacc[0][0] == 1.2345e30f ? 1 : 0);
    if (g < 2) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ored[w][g * 4 + r][nt * 16 + l15] = oacc[nt][r];
    }
    sync();
    for (int t = tid; t < group * 64; t += NT) {
        const int hh = t / 64, d = t % 64;
        float o = ored[0][hh][d];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) o += ored[ww][hh][d];
        if (p.out_fp8_inv > 0.f) ((unsigned char*)p.out)[(long)b * p.ld_out + (kvh * group + hh) * 64 + d] = f2fp8c(rbf(o) * p.out_fp8_inv);
        else p.out[(long)b * p.ld_out + (kvh * group + hh) * 64 + d] = f2bf(o);
    }
    if (A.tl) { wait_vmem(); mark(6); }
}

template <bool F8, int RP, int PV, int LMAX>
NTTS_KERNEL2(256, 2) void qkv_attn_kernel(QkvAttnArgs A) {
    constexpr int NS = 3, KS = 2;
    constexpr int kProd = NS * KS * 96 * 64 * 2, kCons = 8 * (LMAX + 16) * 2 + 64 * 2 + 2 * 4 * kGroupMax * 4 + 4 * kGroupMax * 64 * 4 + (kGroupMax * 64 + 64) * 2;
    NTTS_SHARED char smem[kProd > kCons ? kProd : kCons];
    const unsigned int tag = *A.step_ctr * 32u + (unsigned int)A.layer + 1u;
    const int bid = blockIdx.x;
    if (bid < A.n_prod) {
        if (bid < A.n_prod_real) qa_producer<NS, F8, KS>(A, (bf16_t*)smem, bid, tag);
        return;
    }
    qa_consumer<RP, PV, LMAX>(A, smem, bid - A.n_prod, tag);
}

// grid = producers first (the lowest block ids are dispatched first; they wait for nobody and a CU always has room for one beside two
// consumers' worth of registers only if ... -- not relied upon: a consumer without a CU simply starts later), then batch x kv-heads consumers
template <bool F8>
inline void qkv_attn_launch(QkvAttnArgs A, bool xcd_place, int max_ctx, hipStream_t s) {
    constexpr int KS = 2;
    QkvRopeArgs& p = A.q;
    const int ktiles = p.K / (F8 ? 128 : 64);
    p.kps = (ktiles + KS - 1) / KS;
    const int mblocks = (p.M + 31) / 32, nblocks = p.N / 64;
    int grid = mblocks * nblocks;
    p.xcd_mpx = 0;
    if (xcd_place && p.M % 256 == 0) {
        p.xcd_mpx = p.M / 256;
        grid = 8 * p.xcd_mpx * nblocks;
    }
    A.n_prod_real = grid;
    A.n_prod = (grid + 7) / 8 * 8;
    const int total = A.n_prod + A.batch * A.a.nkv;
    if (max_ctx <= 768) NTTS_LAUNCH((qkv_attn_kernel<F8, 6, 2, 768>), dim3(total), dim3(256), s, A);
    else NTTS_LAUNCH((qkv_attn_kernel<F8, 8, 2, 1024>), dim3(total), dim3(256), s, A);
}

}  // namespace ntts
