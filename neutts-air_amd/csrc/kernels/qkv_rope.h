// qkv_rope.h -- the decode step's QKV projection with everything between the GEMM and the attention scores fused in:
//   q/k/v = nn.Linear(x) (+ bias, ONE rounding to bf16)          hf:models/qwen2/modeling_qwen2.py:206-208
//   q, k  = apply_rotary_pos_emb(q, k, cos, sin)                  hf:models/qwen2/modeling_qwen2.py:113-135,211
//   cache.update(k, v)                                            hf:models/qwen2/modeling_qwen2.py:214, hf:cache_utils.py:127-146
// The rotated q heads and the v heads leave as one bf16 row per sequence; the rotated k head goes straight into the sequence's
// K page ([page][kv_head][32][64], attn_decode.h).  The attention kernel behind it then starts with its K pages instead of a
// prologue (it used to sum this GEMM's fp32 split-K slabs, add the bias, round, rotate and append -- ~3 us during which none of
// its 512 workgroups streamed a byte); it still places its own v row into the transposed V^T page, off its critical path.
//
// Tile: 32 batch rows x 64 output columns (= ONE head: the RoPE partner of feature i is feature i + 32 of the same head, held by
// lane ^ 32 of the same wave) per workgroup of 2 row halves x KS K SLICES.  The K split lives inside the workgroup: every
// slice's tiles stream through the same LDS ring, each wave pair multiplies its own slice, and the fp32 partial sums meet in LDS
// (slice order; at KS = 2 that is the order the two split-K slabs were summed in, so the bits are those of the two-slab path this
// replaces) -- no slab round trip through HBM.  8 x 18 = 144 workgroups at batch 256.
#pragma once
#include <ntts/dev.h>
#include "attn_decode.h"
#include "gemv.h"

namespace ntts {

// ---- per-step row record: what the fused epilogue needs about each sequence, gathered ONCE per decode step (the 24 layers
//      share it) so that no layer's QKV kernel has a dependent load chain (position -> RoPE row / block-table entry)
struct StepMetaArgs {
    const int* pos;          // [M]
    const int* state;        // [M]
    const int* block_table;  // [M][max_pages]
    int max_pages, max_ctx, M;
    const bf16_t* rope_cos;  // [max_ctx][32]
    const bf16_t* rope_sin;
    int* meta;               // [M][4] = {state, pos, page of pos, slot of pos in its page}
    bf16_t* rope_rows;       // [M][64] = cos[pos][0..31] | sin[pos][0..31]
};
NTTS_D void step_meta_row(const StepMetaArgs& p, int row, int i) {   // one wave64 per batch row; i = lane
    int P = p.pos[row];
    if (P < 0) P = 0;
    if (P > p.max_ctx - 1) P = p.max_ctx - 1;      // (free slots hold stale positions: any valid row will do, nothing uses it)
    p.rope_rows[(long)row * 64 + i] = i < 32 ? p.rope_cos[(long)P * 32 + i] : p.rope_sin[(long)P * 32 + i - 32];
    if (i == 0) {
        int* m = p.meta + (long)row * 4;
        m[0] = p.state[row];
        m[1] = P;
        m[2] = p.block_table[(long)row * p.max_pages + P / kPage];
        m[3] = P % kPage;
    }
}
NTTS_KERNEL(256) void step_meta_kernel(StepMetaArgs p) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), i = threadIdx.x & 63;
    if (row >= p.M) return;
    step_meta_row(p, row, i);
}
// The large-batch decode step's first launch: row b's step record (wave 0) next to its embedding gather + RMSNorm (norm.h, one row per
// workgroup) -- two per-row jobs that both wait for the previous step's sample kernel, one launch instead of two (171 per step).
NTTS_KERNEL(128) void embed_norm_meta_kernel(NormArgs p, StepMetaArgs m) {
    if (threadIdx.x < 64 && (int)blockIdx.x < m.M) step_meta_row(m, (int)blockIdx.x, (int)threadIdx.x);
    add_rmsnorm_row_body(p);
}

struct QkvRopeArgs {
    const bf16_t* X;         // [M][K] normalised rows (bf16; fp8 model: e4m3 bytes)
    long ldx;
    const bf16_t* W;         // [N][K] tile-major (gemm.h GemmArgs::w_tile_major), rows q heads | k heads | v heads
    const bf16_t* bias;      // [N] (zeros for a model without attention bias)
    const float* wscale;     // fp8 model: per-output-channel weight scales, xscale = the static input scale
    float xscale;
    int M, N, K;
    int kps;                 // 128-byte K tiles per half (the launcher sets it)
    const int* meta;         // [M][4] step_meta_kernel
    const bf16_t* rope_rows; // [M][64]
    bf16_t* q_out;           // [M][ld_q] q|k|v row: the rotated q heads and the v heads (bf16); the k columns stay unwritten
    long ld_q;
    bf16_t* kpool;           // this layer's K pages
    int nh, nkv;
    unsigned long long* tl;  // diagnostics: [workgroups][16] timestamps of wave 0 (slots as gemm.h GemmArgs::tl; 3 = k-loop done, 4 = K slices met in LDS)
    int xcd_mpx;             // > 0: XCD x (workgroup b runs on XCD b % 8 -- an observation, speed only) takes the xcd_mpx 32-row blocks
                             // that hold batch rows [x * M / 8, (x + 1) * M / 8): the rows the attention workgroups of that XCD read
                             // (attn_decode.h xcd_rows) and whose o_proj tiles run there (gemm.h xcd_maffine)
};

// KS = K slices inside the workgroup (2 or 4): the workgroup is 2 row halves x KS slices = 2 * KS waves, a k-step brings in one
//   128-byte tile of EVERY slice (KS x 12 KB), so K = 896 is 7 steps at KS = 2 and 4 at KS = 4 (slices 4 + 4 + 3 + 3 tiles).
//   The partial sums meet in LDS and are added in slice order.
// TMQ = 16-row blocks per wave (1: the 32 x 64 tile of the <= 256-row decode batch; 2 / 4: 64 / 128 batch rows per workgroup for the WIDE
//   decode step -- 1024 rows, round 6: a W tile then enters LDS once per 64 / 128 rows instead of once per 32, 66 / 50 MB of LDS-DMA
//   traffic per launch instead of 99)
template <int NS, bool F8, int KS, int TMQ = 1>
NTTS_KERNEL(KS * 128) void qkv_rope_kernel(QkvRopeArgs p) {
    constexpr int ESZ = F8 ? 1 : 2;
    constexpr int BM = 32 * TMQ, BN = 64, ROWS = BM + BN;      // LDS rows (128 bytes each) per ring slot and K slice
    constexpr int NINST = ROWS / 8;                       // wave-instructions (8 rows x 128 B) per slot and slice: 12
    constexpr int PER_WAVE = NINST / 2;                   // the two waves of a K slice split them
    static_assert(KS == 2 || KS == 4, "K slices per workgroup");
    static_assert(NS >= 2 && (NS - 1) * PER_WAVE + 3 * TMQ * (4 / KS) + 4 * (4 / KS) + 1 <= 63, "vmcnt range (ring + the epilogue operands requested at entry)");
    NTTS_SHARED bf16_t lds[NS * KS * ROWS * 64];
    static_assert(sizeof(f32x4) * KS * 2 * TMQ * 4 * 64 <= sizeof(bf16_t) * NS * KS * ROWS * 64, "exchange area fits the ring");
    typedef f32x4 (*XchT)[2][TMQ][4][64];
    XchT xch = (XchT)lds;                                 // [slice][row half][row block][j][lane] partial sums (the ring, once everybody is done with it)
    auto swz = [](int rho) { return (rho >> 1) & 7; };

    const int lane = lane_id(), wave = wave_id();
    const int wm = wave & 1, kh = wave >> 1;
    const int g = lane >> 4, l15 = lane & 15;
    const int nblocks = p.N >> 6;
    int mb, nb;
    if (p.xcd_mpx > 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        mb = xcd * p.xcd_mpx + j % p.xcd_mpx;             // m fastest: co-resident workgroups share a W tile
        nb = j / p.xcd_mpx;
    } else if (p.xcd_mpx < 0) {
        // W-stationary (round 5): XCD x (workgroup b runs on XCD b % 8 -- an observation, speed only) takes the column blocks x, x + 8, x + 16 ...
        // with ALL their row blocks, so each private L2 pulls an eighth of W instead of all of it.  In natural order with 8 row blocks of
        // 32 rows (batch 256), b % 8 IS the row block: every XCD streamed the whole W -- the 5.7x over-fetch of round 4's counters.
        const int mblocks = (p.M + BM - 1) / BM;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        mb = j % mblocks;
        nb = xcd + 8 * (j / mblocks);
    } else {
        const int mblocks = (p.M + BM - 1) / BM;
        mb = blockIdx.x % mblocks;
        nb = blockIdx.x / mblocks;
    }
    if (nb >= nblocks) return;                            // (block-uniform, before any barrier)
    const long tlb = (long)blockIdx.x * 16;
    auto mark = [&](int slot) { if (p.tl && wave == 0 && lane == 0) p.tl[tlb + slot] = now_ticks(); };
    mark(0);
    const int m0 = mb * BM, n0 = nb * BN;
    const int ktiles = F8 ? p.K >> 7 : p.K >> 6;
    int nk = ktiles - kh * p.kps;                         // this slice's k-tiles (the later slices may be shorter, even empty)
    if (nk > p.kps) nk = p.kps;
    if (nk < 0) nk = 0;

    // ---- epilogue operands: requested before anything else, so they are the oldest entries of the in-order vector-memory counter
    //      and every later counted wait covers them.  The epilogue is shared out: wave (wm, kh) finishes NJ = 4 / KS of the four
    //      4-feature groups j of its row half -- features nb16 + e0 .. + 4 NJ - 1 of token m in lane (g, l15)
    constexpr int NJ = 4 / KS;
    const int e0 = kh * NJ * 4;
    const int nb16 = n0 + g * 16, nb16p = n0 + (g ^ 2) * 16;             // own features / their RoPE partners (i <-> i + 32)
    u32x4 meta[TMQ];
    bf16x4 cs[TMQ][NJ], sn[TMQ][NJ], bs[NJ], bsp[NJ];
    f32x4 wsc[NJ], wscp[NJ];
#pragma unroll
    for (int a = 0; a < TMQ; ++a) {                                     // this wave's row blocks: rows m0 + (wm * TMQ + a) * 16 + l15
        const int m = m0 + (wm * TMQ + a) * 16 + l15;
        const int mc = m < p.M ? m : p.M - 1;
        meta[a] = ld16<u32x4>(p.meta + (long)mc * 4);
        const bf16_t* rr = p.rope_rows + (long)mc * 64 + (g & 1) * 16 + e0;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            cs[a][jj] = *(const bf16x4*)(rr + jj * 4);
            sn[a][jj] = *(const bf16x4*)(rr + 32 + jj * 4);
        }
    }
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        bs[jj] = *(const bf16x4*)(p.bias + nb16 + e0 + jj * 4);
        bsp[jj] = *(const bf16x4*)(p.bias + nb16p + e0 + jj * 4);
        if constexpr (F8) {
            wsc[jj] = ld16<f32x4>(p.wscale + nb16 + e0 + jj * 4);
            wscp[jj] = ld16<f32x4>(p.wscale + nb16p + e0 + jj * 4);
        }
    }

    // ---- loader: the waves of K slice kh bring in that slice's tiles
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
            const int q = rho - BM;                       // LDS row j*16 + i16  <->  feature (i16>>2)*16 + j*4 + (i16&3) (gemm.h)
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

    f32x4 acc[TMQ][4];
#pragma unroll
    for (int a = 0; a < TMQ; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int xoff[TMQ], xsw[TMQ];
#pragma unroll
    for (int a = 0; a < TMQ; ++a) {
        const int xrho = (wm * TMQ + a) * 16 + l15;
        xoff[a] = xrho * 64;
        xsw[a] = swz(xrho);
    }
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
    mark(1);
    int buf = 0;
    for (int kt = 0; kt < p.kps; ++kt) {                  // (every slice runs kps rounds of the barrier)
        if (kt < nk) {
            if (kt + NS - 2 < nk) wait_vmem_le<(NS - 2) * PER_WAVE>(); else wait_vmem();
        }
        sync_keep_dma();
        if (kt == 0) mark(2);
        if (kt + NS - 1 < nk) stage(kt + NS - 1, buf == 0 ? NS - 1 : buf - 1);
        const bf16_t* base = lds + ((buf * KS + kh) * ROWS) * 64;
        buf = buf + 1 == NS ? 0 : buf + 1;
        if (kt >= nk) continue;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = ks * 4 + g;
            bf16x8 xb[TMQ], wa[4];
#pragma unroll
            for (int a = 0; a < TMQ; ++a) xb[a] = ld16<bf16x8>(base + xoff[a] + ((c ^ xsw[a]) << 3));
#pragma unroll
            for (int j = 0; j < 4; ++j) wa[j] = ld16<bf16x8>(base + woff[j] + ((c ^ wsw[j]) << 3));
#pragma unroll
            for (int a = 0; a < TMQ; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (F8) {
                        const i64x2 w2 = __builtin_bit_cast(i64x2, wa[j]), x2 = __builtin_bit_cast(i64x2, xb[a]);
                        acc[a][j] = mfma16_fp8(w2[0], x2[0], acc[a][j]);
                        acc[a][j] = mfma16_fp8(w2[1], x2[1], acc[a][j]);
                    } else {
                        acc[a][j] = mfma16(wa[j], xb[a], acc[a][j]);
                    }
                }
        }
    }

    // ---- the K slices meet in LDS (every wave parks its partial sums; the ring is free once everybody is past the k-loop), and
    //      every wave finishes its share: sums in slice order + bias -> ONE rounding (the nn.Linear output), then RoPE against the
    //      partner feature i +- 32, which sits in the same table at lane ^ 32 -- no shuffles
    if (p.tl && wave == 0 && lane == 0) p.tl[tlb + 3] = now_ticks() + (acc[0][0][0] == 1.2345e30f ? 1 : 0);
    sync();
#pragma unroll
    for (int a = 0; a < TMQ; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) xch[kh][wm][a][j][lane] = acc[a][j];
    sync();
    mark(4);
    const int hd = n0 >> 6;                               // which head this workgroup's 64 columns are
    const bool rot = hd < p.nh + p.nkv;                   // a q or k head: rotate (rope_pair: every op rounded to bf16)
#pragma unroll
    for (int a = 0; a < TMQ; ++a) {
    const int m = m0 + (wm * TMQ + a) * 16 + l15;
    const int st = (int)meta[a][0], page = (int)meta[a][2], slot = (int)meta[a][3];
    const bool mok = m < p.M, run = mok && st == 1;
    alignas(16) bf16_t out[NJ * 4];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = kh * NJ + jj;
        f32x4 so = xch[0][wm][a][j][lane], sp = xch[0][wm][a][j][lane ^ 32];
        if constexpr (F8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { so[r] *= p.xscale * wsc[jj][r]; sp[r] *= p.xscale * wscp[jj][r]; }
        }
#pragma unroll
        for (int q = 1; q < KS; ++q) {
            const f32x4 o = xch[q][wm][a][j][lane], op = xch[q][wm][a][j][lane ^ 32];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (F8) { so[r] += o[r] * (p.xscale * wsc[jj][r]); sp[r] += op[r] * (p.xscale * wscp[jj][r]); }
                else { so[r] += o[r]; sp[r] += op[r]; }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bf16_t vo = f2bf(so[r] + bf2f((bf16_t)bs[jj][r])), vp = f2bf(sp[r] + bf2f((bf16_t)bsp[jj][r]));
            const float xo = bf2f(vo), xp = bf2f(vp), c = bf2f((bf16_t)cs[a][jj][r]), sv = bf2f((bf16_t)sn[a][jj][r]);
            // g < 2: o1 = x1*c - x2*s with x1 = own, x2 = partner;  g >= 2: o2 = x2*c + x1*s with x2 = own, x1 = partner
            out[jj * 4 + r] = rot ? f2bf(rbf(xo * c) + (g < 2 ? rbf(-xp * sv) : rbf(xp * sv))) : vo;
        }
    }
    // q heads and v heads leave as bf16 columns of the q|k|v row; the k head goes into its page.  The v head's page is TRANSPOSED
    // (64 rows of 2 bytes per token): that scatter costs nothing when each attention workgroup does it for its own (sequence,
    // kv-head) off its critical path, while here it would be 2048 two-byte stores from each of the few workgroups that own a v head
    bf16_t* dst = nullptr;
    if (hd < p.nh || hd >= p.nh + p.nkv) { if (mok) dst = p.q_out + (long)m * p.ld_q + nb16 + e0; }
    else if (run) dst = p.kpool + (((long)page * p.nkv + (hd - p.nh)) * kPage + slot) * 64 + g * 16 + e0;
    if (dst) {
        if constexpr (NJ == 2) *(u32x4*)dst = *(u32x4*)&out[0];
        else *(u32x2*)dst = *(u32x2*)&out[0];
    }
    }   // row blocks
    if (p.tl) { wait_vmem(); mark(5); }
}

// 3 ring slots, 2 K slices per workgroup (MI355X, batch 256: 4 slices 6.56 vs 6.41 us, 2 slots 7.5, 4 / 6 slots 6.4-8.3:
// profiles/r03a_sweep_qkv_fused*.log; KS = 2 also keeps the summation order of the two-slab path this kernel replaced)
// the wide decode step (>= 512 rows): 64 or 128 batch rows per workgroup, 2-slot ring (64 rows: two workgroups per CU)
template <bool F8, int TMQ>
inline void qkv_rope_launch_wide(QkvRopeArgs p, hipStream_t s) {
    constexpr int KS = 2, NS = TMQ == 2 ? 2 : 3;
    const int ktiles = p.K / (F8 ? 128 : 64);
    p.kps = (ktiles + KS - 1) / KS;
    const int mblocks = (p.M + 32 * TMQ - 1) / (32 * TMQ), nblocks = p.N / 64;
    p.xcd_mpx = 0;
    NTTS_LAUNCH((qkv_rope_kernel<NS, F8, KS, TMQ>), dim3(mblocks * nblocks), dim3(KS * 128), s, p);
}

template <bool F8>
inline void qkv_rope_launch(QkvRopeArgs p, bool xcd_place, hipStream_t s, bool w_stationary = false) {
    constexpr int KS = 2, NS = 3;
    const int ktiles = p.K / (F8 ? 128 : 64);
    p.kps = (ktiles + KS - 1) / KS;
    const int mblocks = (p.M + 31) / 32, nblocks = p.N / 64;
    int grid = mblocks * nblocks;
    p.xcd_mpx = 0;
    if (xcd_place && p.M % 256 == 0) {                    // whole 32-row blocks per XCD
        p.xcd_mpx = p.M / 256;
        grid = 8 * p.xcd_mpx * nblocks;
    } else if (w_stationary) {
        p.xcd_mpx = -1;
        grid = 8 * mblocks * ((nblocks + 7) / 8);         // (column blocks padded to whole rounds of the 8 XCDs: the surplus workgroups return at once)
    }
    NTTS_LAUNCH((qkv_rope_kernel<NS, F8, KS>), dim3(grid), dim3(KS * 128), s, p);
}

// ------------------------------------------------------------------------------------------------
// The same fusion for the SMALL-batch step (M <= 16 rows; gemv.h's regime): nothing is tiled, the point is that as many CUs as
// possible each stream a short, private piece of W once.  A workgroup owns 16 output features -- 8 RoPE pairs of one head,
// features {8q .. 8q+7} and {32+8q .. 32+8q+7} -- over the WHOLE K: its 4 feature waves each take a K slice (their weight
// fragments go HBM -> VGPR, all requested at entry, as in gemv.h), the 4 helper waves build the normalised X panel in LDS
// meanwhile (gemv.h's fused prologue: sum the previous down_proj's split-K slabs + residual + RMSNorm), the four partial sums
// meet in LDS in slice order (the order the attention prologue used to add the four slabs in: same bits), and wave 0 finishes:
// bias, ONE rounding, RoPE against the partner rows (lane ^ 32), q / v columns to the bf16 row, the k pair into its page.
// 72 workgroups at N = 1152 -- as many as the split-K GEMV it replaces had -- and no fp32 slabs for the attention kernel to reduce.
struct GemvQkvArgs {
    NormArgs pro;            // the fused prologue's operands (gemv.h PRO)
    const bf16_t* W;         // [N][K] tile-major
    const bf16_t* bias;      // [N]
    int M, N, K;
    int kps;                 // k-tiles per K slice (launcher)
    const int* meta;         // [M][4] step_meta_kernel
    const bf16_t* rope_rows; // [M][64]
    bf16_t* q_out;           // [M][ld_q] q|k|v row (q rotated, k columns unwritten, v as is)
    long ld_q;
    bf16_t* kpool;
    int nh, nkv;
    const float* wscale;     // F8 kernel (fp8 model; gemv.h GemvArgs): per-output-channel weight scales, xscale = the static input scale;
    float xscale;            //   pro.out_fp8_inv quantises the panel
    unsigned long long* tl;  // diagnostics (ntts_backbone_gemv_timeline, which = 1): [workgroups][16] phase timestamps as in gemv.h; slot 5 =
                             // the K slices have met in LDS, slot 6 = wave 0's stores done
};

template <int KT, bool F8 = false>
NTTS_KERNEL(512) void gemv_qkv_rope_kernel(GemvQkvArgs p) {
    constexpr int ESZ = F8 ? 1 : 2;
    NTTS_SHARED bf16_t xs[kGemvRows * kGemvXld];
    NTTS_SHARED f32x4 red[4][64];
    const int lane = lane_id(), w = wave_id();
    const int g = lane >> 4, l15 = lane & 15;
    const long tlb = (long)blockIdx.x * 16;
    auto mark = [&](int slot) { if (p.tl && lane == 0) p.tl[tlb + slot] = now_ticks(); };
    if (w >= 4) {            // helper waves: the normalised X panel (block 0 also writes the new residual stream); SG = 10: the ten
        if (w == 4) mark(8);  // down_proj slabs are ONE round trip (with 8 + 2 the panel was ready 0.7 us later: profiles/r03g_gemv_timeline_b1.txt)
        bool first = true;    // (the issue barrier of norm.h / gemv.h: the prologue's requests ahead of the weights)
        for (int m = w - 4; m < p.M; m += 4) { rmsnorm_row_wave<2, 10, true>(p.pro, m, true, blockIdx.x == 0, xs + m * kGemvXld, first); first = false; }
        if (first) sync_keep_dma();
        if (w == 4) mark(9);
        sync();
        if (w == 4) mark(10);
        sync();
        return;
    }
    if (w == 0) mark(0);
    const int hd = blockIdx.x >> 2, q = blockIdx.x & 3;
    const int m = l15, mc = m < p.M ? m : p.M - 1;
    // wave 0's epilogue operands first (tiny; a wave's loads return in order)
    const int i0 = q * 8 + (g & 1) * 4;                            // RoPE index of this lane's 4 features
    const int n0 = hd * 64 + (g >= 2 ? 32 : 0) + i0;               // their columns
    u32x4 meta = {0, 0, 0, 0};
    bf16x4 cs = {0, 0, 0, 0}, sn = {0, 0, 0, 0}, bs = {0, 0, 0, 0};
    if (w == 0) {
        meta = ld16<u32x4>(p.meta + (long)mc * 4);
        cs = *(const bf16x4*)(p.rope_rows + (long)mc * 64 + i0);
        sn = *(const bf16x4*)(p.rope_rows + (long)mc * 64 + 32 + i0);
        bs = *(const bf16x4*)(p.bias + n0);
    }
    // this wave's K slice of the workgroup's 16 weight rows: MFMA row l15 <-> feature 8q + l15 (l15 < 8) / 32 + 8q + l15 - 8
    const int ktiles = F8 ? p.K >> 7 : p.K >> 6;                  // 128-byte k-tiles
    const int kt0 = w * p.kps;
    int nk = ktiles - kt0;
    if (nk > p.kps) nk = p.kps;
    if (nk < 0) nk = 0;
    const int frow = l15 < 8 ? q * 8 + l15 : 32 + q * 8 + (l15 - 8);
    const bf16_t* wbase = (const bf16_t*)((const char*)p.W + (long)hd * 64 * p.K * ESZ) + frow * 64 + g * 16;
    bf16x8 wa[KT][2];
    sync_keep_dma();                                              // the helpers' requests go first
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const int jj = j < nk ? j : (nk > 0 ? nk - 1 : 0);
        const int kt = kt0 + jj < ktiles ? kt0 + jj : ktiles - 1;
        const bf16_t* src = wbase + (long)kt * 4096;
        wa[j][0] = ld16<bf16x8>(src);
        wa[j][1] = ld16<bf16x8>(src + 8);
    }
    if (w == 0) mark(1);
    if (p.tl) { wait_vmem(); if (w == 0) mark(2); }
    sync();                                                       // the panel is complete
    if (w == 0) mark(3);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bf16_t* xrow = xs + l15 * kGemvXld + g * 16;
    bf16x8 xq[KT][2];                                             // the slice's X fragments, all read before the matrix-core chain (gemv.h)
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const int jj = j < nk ? j : (nk > 0 ? nk - 1 : 0);
        const int kt = kt0 + jj < ktiles ? kt0 + jj : ktiles - 1;
        const bf16_t* xp = xrow + kt * 64;
        xq[j][0] = ld16<bf16x8>(xp);
        xq[j][1] = ld16<bf16x8>(xp + 8);
    }
    sched_fence();
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const short keep = j < nk ? (short)-1 : (short)0;         // surplus tiles contribute 0 * x
        if constexpr (F8) {
            const i64x2 a0 = __builtin_bit_cast(i64x2, wa[j][0] & keep), a1 = __builtin_bit_cast(i64x2, wa[j][1] & keep);
            const i64x2 b0 = __builtin_bit_cast(i64x2, xq[j][0]), b1 = __builtin_bit_cast(i64x2, xq[j][1]);
            acc = mfma16_fp8(a0[0], b0[0], acc);
            acc = mfma16_fp8(a0[1], b0[1], acc);
            acc = mfma16_fp8(a1[0], b1[0], acc);
            acc = mfma16_fp8(a1[1], b1[1], acc);
        } else {
            acc = mfma16(wa[j][0] & keep, xq[j][0], acc);
            acc = mfma16(wa[j][1] & keep, xq[j][1], acc);
        }
    }
    if (p.tl && w == 0 && lane == 0) p.tl[tlb + 4] = now_ticks() + (acc[0] == 1.2345e30f ? 1 : 0);
    red[w][lane] = acc;
    sync();
    if (w != 0) return;
    mark(5);
    // lane (g, m): rows g*4 + r of the 16 = features n0 + r of token m; slices added in order
    f32x4 sum = red[0][lane];
    f32x4 sc = {1.f, 1.f, 1.f, 1.f};
    if constexpr (F8) {                                           // every slice's partial sum times (input scale x weight scale of column n0 + r), as qkv_rope_kernel<F8>
        const f32x4 ws = ld16<f32x4>(p.wscale + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { sc[r] = p.xscale * ws[r]; sum[r] *= sc[r]; }
    }
#pragma unroll
    for (int sl = 1; sl < 4; ++sl) {
        const f32x4 o = red[sl][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (F8) sum[r] += o[r] * sc[r]; else sum[r] += o[r];
        }
    }
    alignas(8) bf16_t val[4], out[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) val[r] = f2bf(sum[r] + bf2f((bf16_t)bs[r]));      // the nn.Linear output: one rounding
    const bool rot = hd < p.nh + p.nkv;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        const int own = (int)val[2 * k2] | ((int)val[2 * k2 + 1] << 16);
        const int par = shfl_xor(own, 32);                         // lanes g < 2 hold x[i], lanes g >= 2 hold x[i + 32]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = 2 * k2 + h;
            const float xo = bf2f((bf16_t)((own >> (16 * h)) & 0xffff)), xp = bf2f((bf16_t)((par >> (16 * h)) & 0xffff));
            const float c = bf2f((bf16_t)cs[r]), sv = bf2f((bf16_t)sn[r]);
            out[r] = rot ? f2bf(rbf(xo * c) + (g < 2 ? rbf(-xp * sv) : rbf(xp * sv))) : val[r];
        }
    }
    const int st = (int)meta[0], page = (int)meta[2], slot = (int)meta[3];
    const bool mok = m < p.M;
    bf16_t* dst = nullptr;
    if (hd < p.nh || hd >= p.nh + p.nkv) { if (mok) dst = p.q_out + (long)m * p.ld_q + n0; }
    else if (mok && st == 1) dst = p.kpool + (((long)page * p.nkv + (hd - p.nh)) * kPage + slot) * 64 + (n0 & 63);
    if (dst) *(u32x2*)dst = *(u32x2*)&out[0];
    if (p.tl) { wait_vmem(); mark(6); }
}

inline void gemv_qkv_rope_launch(GemvQkvArgs p, hipStream_t s) {
    if (p.wscale) {                                               // fp8 model: 128 k-values per k-tile, <= 2 per slice for K <= 1024
        p.kps = (p.K / 128 + 3) / 4;
        NTTS_LAUNCH((gemv_qkv_rope_kernel<2, true>), dim3(p.N / 16), dim3(512), s, p);
        return;
    }
    const int ktiles = p.K / 64;
    p.kps = (ktiles + 3) / 4;                                     // <= 4 for K <= 1024
    NTTS_LAUNCH((gemv_qkv_rope_kernel<4>), dim3(p.N / 16), dim3(512), s, p);
}

}  // namespace ntts
