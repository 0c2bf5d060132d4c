// codec.h -- NeuCodec decoder kernels (everything that is not a GEMM).
//
// Replaces NeuCodec.decode_code (ref:neutts/neutts.py:288-291) as restated by transformers' xcodec2
// (hf:models/xcodec2/modeling_xcodec2.py): FSQ de-index + project_out + fc (:806-809, :839), Conv1d k7 / k3
// (as GEMMs over overlapping rows, gemm.h), GroupNorm(32)+SiLU (:650-659), RMSNorm / LayerNorm (:320-325,
// :862), full (non-causal) attention (:268-308), ISTFT head (:762-796).
//
// Row layout ("packed padded rows"): utterance b owns rows [off[b], off[b+1]) = its lens[b] frames between kPadRows pad rows on
// either side; frame t sits at row off[b] + kPadRows + t.  A ragged batch therefore costs its OWN frames, not B x the longest
// utterance (150-350-frame batches: 1.4x fewer GEMM rows); equal lengths give the layout b * (T + 2 kPadRows) of the earlier rounds.
// Pad rows of every bf16 GEMM-input buffer are ZERO, which is exactly Conv1d's zero padding (6 zero rows between two utterances),
// so a k-tap convolution is one GEMM with lda = C and K = k*C over overlapping rows.
// The residual stream is fp32 (the reference codec runs in fp32); only GEMM operands are bf16.
#pragma once
#include <ntts/dev.h>
#include "attn_decode.h"
#include "gemm.h"   // silu_fast

namespace ntts {

constexpr int kPadRows = 3;  // covers the k=7 stem (padding 3) and the k=3 ResNet convs (padding 1)

struct CodecRows {
    const int* lens;   // [B] valid frames of each utterance
    const int* off;    // [B + 1] first row of utterance b (its leading pad rows); off[B] = rows
    int B, Tp;         // Tp = longest utterance + 2 * kPadRows (grid sizing only)
    long rows;         // off[B]
};
NTTS_D long codec_row0(const CodecRows& R, int b) { return (long)R.off[b] + kPadRows; }   // row of frame 0 of utterance b

NTTS_D bool codec_row(const CodecRows& R, long r, int& b, int& t) {   // which (utterance, frame) is row r: binary search in off[]
    int lo = 0, hi = R.B - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((long)R.off[mid] <= r) lo = mid; else hi = mid - 1;
    }
    b = lo;
    t = (int)(r - R.off[lo]) - kPadRows;
    return t >= 0 && t < R.lens[b];
}

// ---- precision = high: SPLIT bf16 GEMM operands -------------------------------------------------------------------------------------
// A bf16 operand row of C values is written as [hi | lo | hi] (3 C columns), hi = bf16(x), lo = bf16(x - hi): x to ~16 mantissa bits.
// Against a weight row [wh | wh | wl] (codec.cpp Finalizer::op) the GEMM's ordinary K-loop forms xh wh + xl wh + xh wl in its fp32
// accumulator: the operand-rounding error of a bf16 GEMM (2^-9 relative per operand -- 7e-3 of the waveform over ~60 GEMMs in series,
// DESIGN.md section 2) drops to ~2^-17 at three times the matrix-core work.  The ISTFT head's DFT operand has always been built this way
// (istft_prep_kernel).  `split` = 0: the plain bf16 row of C columns.
// ---- precision = fp16 (the default, `split` = kOpF16): the plain row of C columns as IEEE HALVES for v_mfma_f32_16x16x32_f16 -- 11 significant
// bits per operand at the bf16 rate and the bf16 bytes: 8.1e-4 relative on the waveform from the GEMM operands (tools/codec_operand_sim.py predicts 8.0e-4; bf16 7.4e-3), 9.5e-4 with the ISTFT as ONE fp16 term per bin (istft_prep_kernel: 37.6 -> 37.0 ms per 256 x 250 frames).
constexpr int kOpBf16 = 0, kOpSplit = 1, kOpF16 = 2;
NTTS_D void put_op(bf16_t* y, long r, int C, int ch, float v, int split) {
    if (split == kOpF16) { y[r * C + ch] = f2h(v); return; }
    const bf16_t hi = f2bf(v);
    if (!split) { y[r * C + ch] = hi; return; }
    bf16_t* row = y + r * 3 * C;
    row[ch] = hi; row[C + ch] = f2bf(v - bf2f(hi)); row[2 * C + ch] = hi;
}
NTTS_D void put_op4(bf16_t* y, long r, int C, int ch, const float (&v)[4], int split) {      // 4 consecutive channels, ch % 4 == 0
    bf16x4 hi, lo;
    if (split == kOpF16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) hi[e] = (short)f2h(v[e]);
        *(bf16x4*)(y + r * C + ch) = hi;
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { hi[e] = (short)f2bf(v[e]); lo[e] = (short)f2bf(v[e] - bf2f((bf16_t)hi[e])); }
    if (!split) { *(bf16x4*)(y + r * C + ch) = hi; return; }
    bf16_t* row = y + r * 3 * C;
    *(bf16x4*)(row + ch) = hi; *(bf16x4*)(row + C + ch) = lo; *(bf16x4*)(row + 2 * C + ch) = hi;
}

// ---- FSQ de-index + (project_out o fc) folded into one affine  8 -> H --------------------------------
struct CodecEmbedArgs {
    const int* codes;      // packed, utterance b starts at code_off[b]
    const int* code_off;
    const float* wf;       // [H][nq] folded weight (fp32)
    const float* bf;       // [H]
    bf16_t* out;           // [rows][H]  (split: [rows][3 H], put_op)
    CodecRows R;
    int H, nq;
    int levels[8];
    int split;
};
// kEmbedRows rows per workgroup: a thread keeps the folded weights of its channels (H / 256 channels x nq <= 8 weights) in
// registers across the rows (one row per workgroup was 65 536 launches' worth of tiny workgroups at 256 x 250 frames, each
// re-reading its weights: 0.9 ms per pass)
constexpr int kEmbedRows = 16, kEmbedChMax = 8;   // channels per thread held in registers: H <= 2048
NTTS_KERNEL(256) void codec_embed_kernel(CodecEmbedArgs p) {
    const long r0 = (long)blockIdx.x * kEmbedRows;
    const long rows = p.R.rows;
    float wreg[kEmbedChMax][8], breg[kEmbedChMax];
#pragma unroll
    for (int j = 0; j < kEmbedChMax; ++j) {
        const int c = threadIdx.x + 256 * j;
        breg[j] = c < p.H ? p.bf[c] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) wreg[j][i] = (c < p.H && i < p.nq) ? p.wf[(long)c * p.nq + i] : 0.f;
    }
    for (int rr = 0; rr < kEmbedRows; ++rr) {
        const long r = r0 + rr;
        if (r >= rows) break;                                   // block-uniform
        int b, t;
        const bool ok = codec_row(p.R, r, b, t);
        float val[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) val[i] = 0.f;
        if (ok) {
            int code = p.codes[p.code_off[b] + t];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i < p.nq) {
                    const int L = p.levels[i], d = code % L, half = L / 2;   // hf:...modeling_xcodec2.py:680-690
                    code /= L;
                    val[i] = (float)(d - half) / (float)half;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kEmbedChMax; ++j) {
            const int c = threadIdx.x + 256 * j;
            if (c < p.H) {
                float acc = 0.f;
                if (ok) {
                    acc = breg[j];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i < p.nq) acc += wreg[j][i] * val[i];
                }
                put_op(p.out, r, p.H, c, acc, p.split);
            }
        }
    }
}

// ---- GroupNorm(32 groups, eps) + SiLU : fp32 [rows][C] -> bf16 [rows][C], pad rows zeroed ---------------
struct GroupNormArgs {
    const float* x;
    bf16_t* y;
    const float* gamma;
    const float* beta;
    CodecRows R;
    int C;
    float eps;
    int split;             // y is a split operand [rows][3 C] (put_op)
};
// grid (B, 32); block 256 = (256/cg row lanes) x (cg channels)
NTTS_KERNEL(256) void groupnorm_silu_kernel(GroupNormArgs p) {
    NTTS_SHARED float red[2][4];
    const int b = blockIdx.x, grp = blockIdx.y, tid = threadIdx.x;
    const int cg = p.C / 32, nrl = 256 / cg;
    const int ch = grp * cg + tid % cg, rl = tid / cg;
    const int T = p.R.lens[b];
    const long row0 = codec_row0(p.R, b);
    float s = 0.f, ss = 0.f;
    for (int t = rl; t < T; t += nrl) {
        const float v = p.x[(row0 + t) * p.C + ch];
        s += v;
        ss += v * v;
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) { s += shfl_xor(s, sh); ss += shfl_xor(ss, sh); }
    if (lane_id() == 0) { red[0][wave_id()] = s; red[1][wave_id()] = ss; }
    sync();
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    ss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const float n = (float)T * (float)cg;
    const float mean = s / n;
    float var = ss / n - mean * mean;
    if (var < 0.f) var = 0.f;
    const float rstd = frsqrt_exact(var + p.eps);
    const float ga = p.gamma[ch], be = p.beta[ch];
    for (int t = rl - kPadRows; t < T + kPadRows; t += nrl) {   // all rows of the utterance, pads included
        float o = 0.f;
        if (t >= 0 && t < T) {
            const float v = (p.x[(row0 + t) * p.C + ch] - mean) * rstd * ga + be;
            o = silu_fast(v);
        }
        put_op(p.y, row0 + t, p.C, ch, o, p.split);
    }
}

// The same for utterances of up to 256 frames with 4-channel (16-byte) accesses and the utterance's slice held in REGISTERS between
// the statistics and the normalisation: every fp32 input is read once, all of a thread's loads are in flight together (the
// kernel above reads each value twice, 4 bytes at a time, one dependent iteration after the other: 205 us per launch at
// 256 x 250 frames against ~70 us of bytes).  Same arithmetic per element; the sums run in another order.
// grid (B, 32); block 256 = (256 / (cg/4) row lanes) x (cg/4 float4 lanes); needs cg % 4 == 0, 256 % (cg/4) == 0, T <= 8 * row lanes
// kGnRegIters = row iterations per thread the instantiation holds in registers: 8 (256 frames at 32 row lanes) or 16 (512 frames)
template <int kGnRegIters>
NTTS_KERNEL(256) void groupnorm_silu_reg_kernel(GroupNormArgs p) {
    NTTS_SHARED float red[2][4];
    const int b = blockIdx.x, grp = blockIdx.y, tid = threadIdx.x;
    const int cg = p.C / 32, vl = cg >> 2, nrl = 256 / vl;
    const int ch = grp * cg + (tid % vl) * 4, rl = tid / vl;
    const int T = p.R.lens[b];
    const long row0 = codec_row0(p.R, b);
    f32x4 v[kGnRegIters];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < kGnRegIters; ++i) {
        const int t = rl + i * nrl;
        v[i] = ld16<f32x4>(p.x + (row0 + (t < T ? t : (T > 0 ? T - 1 : 0))) * p.C + ch);      // clamped address, no branch around the load
    }
#pragma unroll
    for (int i = 0; i < kGnRegIters; ++i) {
        if (rl + i * nrl < T) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { s += v[i][e]; ss += v[i][e] * v[i][e]; }
        }
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) { s += shfl_xor(s, sh); ss += shfl_xor(ss, sh); }
    if (lane_id() == 0) { red[0][wave_id()] = s; red[1][wave_id()] = ss; }
    sync();
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    ss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const float n = (float)T * (float)cg;
    const float mean = s / n;
    float var = ss / n - mean * mean;
    if (var < 0.f) var = 0.f;
    const float rstd = frsqrt_exact(var + p.eps);
    const f32x4 ga = ld16<f32x4>(p.gamma + ch), be = ld16<f32x4>(p.beta + ch);
#pragma unroll
    for (int i = 0; i < kGnRegIters; ++i) {
        const int t = rl + i * nrl;
        if (t < T) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = silu_fast((v[i][e] - mean) * rstd * ga[e] + be[e]);
            put_op4(p.y, row0 + t, p.C, ch, o, p.split);
        }
    }
    // pad rows of the utterance: zero, as the Conv1d padding wants them
    const float z[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = rl - kPadRows; t < T + kPadRows; t += nrl)
        if (t < 0 || t >= T) put_op4(p.y, row0 + t, p.C, ch, z, p.split);
}
inline void groupnorm_silu_launch(const GroupNormArgs& p, int Tmax, hipStream_t s) {
    const int cg = p.C / 32, vl = cg / 4;
    if (cg % 4 == 0 && vl >= 1 && 256 % vl == 0 && Tmax <= 8 * (256 / vl)) NTTS_LAUNCH((groupnorm_silu_reg_kernel<8>), dim3(p.R.B, 32), dim3(256), s, p);
    else if (cg % 4 == 0 && vl >= 1 && 256 % vl == 0 && Tmax <= 16 * (256 / vl)) NTTS_LAUNCH((groupnorm_silu_reg_kernel<16>), dim3(p.R.B, 32), dim3(256), s, p);
    else NTTS_LAUNCH((groupnorm_silu_kernel), dim3(p.R.B, 32), dim3(256), s, p);
}

// ---- RMSNorm / LayerNorm over a row: fp32 -> bf16, one wave per row -------------------------------------
struct RowNormArgs {
    const float* x;
    bf16_t* y;
    const float* w;
    const float* bias;  // nullptr -> RMSNorm, else LayerNorm
    long rows;
    int C;
    float eps;
    int split;          // y is a split operand [rows][3 C] (put_op)
};
template <int NV>  // float4 per lane: C <= 256 * NV
NTTS_KERNEL(256) void rownorm_kernel(RowNormArgs p) {
    const int lane = lane_id();
    long row = (long)blockIdx.x * 4 + wave_id();
    const bool rok = row < p.rows;
    if (!rok) row = p.rows - 1;
    const int nvec = p.C >> 2;
    f32x4 v[NV];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (vi < nvec) v[i] = ld16<f32x4>(p.x + row * p.C + vi * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s += v[i][e]; ss += v[i][e] * v[i][e]; }
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) { s += shfl_xor(s, sh); ss += shfl_xor(ss, sh); }
    const float n = (float)p.C;
    float mean = 0.f, inv;
    if (p.bias) {
        mean = s / n;
        float var = ss / n - mean * mean;
        if (var < 0.f) var = 0.f;
        inv = frsqrt_exact(var + p.eps);
    } else {
        inv = frsqrt_exact(ss / n + p.eps);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < nvec && rok) {
            const f32x4 w = ld16<f32x4>(p.w + vi * 4);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y = (v[i][e] - mean) * inv * w[e];
                if (p.bias) y += p.bias[vi * 4 + e];
                o[e] = y;
            }
            put_op4(p.y, row, p.C, vi * 4, o, p.split);
        }
    }
}
inline void rownorm_launch(const RowNormArgs& p, hipStream_t s) {
    const dim3 grid((unsigned)((p.rows + 3) / 4)), block(256);
    if (p.C <= 256) NTTS_LAUNCH((rownorm_kernel<1>), grid, block, s, p);
    else if (p.C <= 1024) NTTS_LAUNCH((rownorm_kernel<4>), grid, block, s, p);
    else NTTS_LAUNCH((rownorm_kernel<8>), grid, block, s, p);
}

// ---- V -> V^T pages: qkv[rows][3C] (v part) -> vt[b][head][page][64 d][32 frames], frames >= T zeroed ------
struct VTransposeArgs {
    const bf16_t* qkv;
    bf16_t* vt;
    CodecRows R;
    int C, nh, npages;
};
// grid (B * npages, nh)
NTTS_KERNEL(256) void v_transpose_kernel(VTransposeArgs p) {
    NTTS_SHARED bf16_t tile[kPage][64 + 2];
    const int b = blockIdx.x / p.npages, pg = blockIdx.x % p.npages, h = blockIdx.y, tid = threadIdx.x;
    const int T = p.R.lens[b];
    const long row0 = codec_row0(p.R, b) + (long)pg * kPage;
    for (int x = tid; x < kPage * 64; x += 256) {
        const int tk = x >> 6, d = x & 63;
        const int t = pg * kPage + tk;
        tile[tk][d] = (t < T) ? p.qkv[(row0 + tk) * (3L * p.C) + 2 * p.C + h * 64 + d] : (bf16_t)0;
    }
    sync();
    bf16_t* dst = p.vt + (((long)b * p.nh + h) * p.npages + pg) * 64 * kPage;
    for (int x = tid; x < 64 * kPage; x += 256) {
        const int d = x >> 5, tk = x & 31;
        dst[d * kPage + tk] = tile[tk][d];
    }
}

// ---- full (non-causal) multi-head attention within each utterance ----------------------------------------
struct AttnFullArgs {
    const bf16_t* qkv;   // [rows][3C]: q | k | v
    const bf16_t* vt;    // from v_transpose_kernel
    bf16_t* out;         // [rows][C]  (split: [rows][3 C], put_op)
    CodecRows R;
    int C, nh, npages, qtiles;
    int split;
};
// grid (B * qtiles, nh); 4 waves x 16 queries.  Same matrix-core mapping as attn_prefill.h: the workgroup's K page
// (gathered from the qkv rows) and V^T page are staged ONCE in LDS by LDS-DMA (double-buffered) and shared by the 4
// waves, instead of every wave pulling them through its own load path.  Two sweeps (max / denominator, then PV).
//   K image [32 keys][128 B], chunk c of key r at c ^ (r & 7);  V^T image [64 d][64 B], 16-B unit u of row d at
//   u ^ ((d >> 2) & 3)  -- V^T pages come from v_transpose_kernel in plain token order, so a lane's 8 keys are the
//   two 8-byte halves (units g>>1 and 2 + (g>>1)).
template <bool F16>
NTTS_KERNEL(256) void attn_full_kernel(AttnFullArgs p) {
    NTTS_SHARED bf16_t lds[2 * 2 * kPage * 64];
    const int lane = lane_id(), w = wave_id();
    const int g = lane >> 4, l15 = lane & 15;
    const int b = blockIdx.x / p.qtiles, qt = blockIdx.x % p.qtiles, h = blockIdx.y;
    const int T = p.R.lens[b];
    if (qt * 64 >= T) return;                                  // block-uniform
    const int qw0 = qt * 64 + w * 16;
    const bool wave_live = qw0 < T;
    const long row0 = codec_row0(p.R, b);
    const long ld = 3L * p.C;
    int qi = qw0 + l15;
    if (qi > T - 1) qi = T - 1;
    const bf16_t* qr = p.qkv + (row0 + qi) * ld + h * 64 + g * 16;
    bf16x8 qB[2];
    qB[0] = ld16<bf16x8>(qr);
    qB[1] = ld16<bf16x8>(qr + 8);
    const int npg = (T + kPage - 1) / kPage;

    auto stage = [&](int pg, int buf) {
        bf16_t* dst = lds + buf * (2 * kPage * 64);
        if (w < 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int inst = w * 2 + i;                    // 8 keys per instruction
                const int r = inst * 8 + (lane >> 3);
                int kt = pg * kPage + r;
                if (kt > T - 1) kt = T - 1;
                const int c = (lane & 7) ^ (r & 7);
                glds16(p.qkv + (row0 + kt) * ld + p.C + h * 64 + c * 8, dst + inst * 512);
            }
        } else {
            const bf16_t* vp = p.vt + (((long)b * p.nh + h) * p.npages + pg) * 64 * kPage;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int inst = (w - 2) * 2 + i;              // 16 d-rows per instruction
                const int d = inst * 16 + (lane >> 2);
                const int u = (lane & 3) ^ ((d >> 2) & 3);
                glds16(vp + d * kPage + u * 8, dst + kPage * 64 + inst * 512);
            }
        }
    };
    constexpr float kMasked = -1.0e30f;
    auto scores = [&](const bf16_t* kb, int pg, float (&s)[8]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = u * 16 + l15;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = mfma16_op<F16>(ld16<bf16x8>(kb + r * 64 + (((2 * g) ^ (r & 7)) << 3)), qB[0], a);
            a = mfma16_op<F16>(ld16<bf16x8>(kb + r * 64 + (((2 * g + 1) ^ (r & 7)) << 3)), qB[1], a);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int key = pg * kPage + u * 16 + g * 4 + rr;
                s[u * 4 + rr] = key < T ? a[rr] * 0.125f : kMasked;
            }
        }
    };
    float m = kMasked, sum = 0.f;
    stage(0, 0);
    for (int pg = 0; pg < npg; ++pg) {
        wait_vmem();
        sync();
        if (pg + 1 < npg) stage(pg + 1, (pg + 1) & 1);
        if (wave_live) {
            float s[8];
            scores(lds + (pg & 1) * (2 * kPage * 64), pg, s);
            const float tm = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
            const float mn = fmaxf(m, tm);
            float add = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) add += fexp_neg(s[e] - mn);
            sum = sum * fexp_neg(m - mn) + add;
            m = mn;
        }
    }
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {
        const float om = shfl_xor(m, sh), os = shfl_xor(sum, sh);
        const float mn = fmaxf(m, om);
        sum = sum * fexp_neg(m - mn) + os * fexp_neg(om - mn);
        m = mn;
    }
    f32x4 oacc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) oacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float rs = 1.0f / sum;
    sync();
    stage(0, 0);
    for (int pg = 0; pg < npg; ++pg) {
        wait_vmem();
        sync();
        if (pg + 1 < npg) stage(pg + 1, (pg + 1) & 1);
        if (wave_live) {
            const bf16_t* kb = lds + (pg & 1) * (2 * kPage * 64);
            const bf16_t* vb = kb + kPage * 64;
            float s[8];
            scores(kb, pg, s);
            bf16x8 pA;
#pragma unroll
            for (int e = 0; e < 8; ++e) pA[e] = (short)f2op<F16>(fexp_neg(s[e] - m) * rs);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int d = nt * 16 + l15;
                const int sw = (d >> 2) & 3;
                const bf16x4 v0 = ld16<bf16x4>(vb + d * kPage + (((g >> 1) ^ sw) << 3) + (g & 1) * 4);
                const bf16x4 v1 = ld16<bf16x4>(vb + d * kPage + (((2 + (g >> 1)) ^ sw) << 3) + (g & 1) * 4);
                bf16x8 vB;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vB[e] = v0[e]; vB[4 + e] = v1[e]; }
                oacc[nt] = mfma16_op<F16>(pA, vB, oacc[nt]);
            }
        }
    }
    if (wave_live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qw0 + g * 4 + r;
            if (q < T) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) put_op(p.out, row0 + q, p.C, h * 64 + l15 + nt * 16, oacc[nt][r], p.split);
            }
        }
    }
}

// ---- the same attention for utterances of up to 256 frames (5 s of audio; the benchmark's 250), with K and V^T RESIDENT ----------
// grid (B, nh): one workgroup per (utterance, head) stages that head's whole K (LDS-DMA, 8 pages) and V^T (transposed on the
// way in: no v_transpose_kernel pass, no V^T round trip through HBM) ONCE -- 64 KB -- and its 8 waves then walk the 16-query
// tiles with no barrier in the loop.  ONE sweep over the keys (online softmax: running maximum m, P = bf16(exp(s - m)) unnormalised,
// accumulator and denominator rescaled by exp(m_old - m_new) per page, one division at the end) instead of two: QK^T is
// computed once and every score costs one exp instead of two.  Allowed here because the codec's reference is fp32 with a
// waveform tolerance -- the backbone's eager contract (P rounded AFTER the global normalisation) is not.
// The rounding of P to bf16 before the PV product is the same 2^-9 relative perturbation as in attn_full_kernel.
// Replaces hf:models/xcodec2/modeling_xcodec2.py:242-331 (Xcodec2Attention, non-causal) like attn_full_kernel.
constexpr int kAttnResPages = 16;  // resident pages at most: 512 frames (NP = 8 / 12 / 16 instantiations: 64 / 96 / 128 KB of LDS)
// 8 waves: the per-page chain K-MFMA -> row maximum (two cross-lane steps) -> exp -> PV-MFMA is serial inside a wave, and the 64 KB of
// LDS allow two workgroups per CU -- with 4 waves each that was 2 waves per SIMD and the kernel ran at the latency of that chain
// (289 us per launch; trimming its arithmetic changed nothing); 8 waves halve the query tiles per wave and double the waves per SIMD.
// NP = pages the instantiation can hold (the launcher picks the smallest that fits the batch's longest utterance): 8 pages = 64 KB, two workgroups
// per CU; 12 / 16 pages (384 / 512 frames: a ragged batch of 150-350-frame utterances used to fall to the paged two-sweep kernel as a whole) one.
template <int NP, bool F16>
NTTS_KERNEL(512) void attn_full_resident_kernel(AttnFullArgs p) {
    NTTS_SHARED bf16_t kres[NP * kPage * 64];    // [page][32 keys][128 B], chunk c of key r at c ^ (r & 7)
    NTTS_SHARED bf16_t vres[NP * 64 * kPage];    // [page][64 d][64 B], 16-B unit u of row d at u ^ ((d >> 2) & 3)
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int g = lane >> 4, l15 = lane & 15;
    const int b = blockIdx.x, h = blockIdx.y;
    const int T = p.R.lens[b];
    if (T < 1) return;                                         // block-uniform
    const long row0 = codec_row0(p.R, b);
    const long ld = 3L * p.C;
    const int npg = (T + kPage - 1) / kPage;                  // <= NP (launcher)
    // ---- K: LDS-DMA, all pages requested at once (frames past T re-read the last one; masked below)
    for (int inst = w; inst < npg * 4; inst += 8) {           // one instruction = 8 keys x 128 B
        const int pg = inst >> 2, r = (inst & 3) * 8 + (lane >> 3);
        int kt = pg * kPage + r;
        if (kt > T - 1) kt = T - 1;
        const int c = (lane & 7) ^ (r & 7);
        glds16(p.qkv + (row0 + kt) * ld + p.C + h * 64 + c * 8, kres + pg * (kPage * 64) + (inst & 3) * 512);
    }
    // ---- V -> V^T: a work item = (frame, 8 d); consecutive lanes = consecutive frames, so the 2-byte LDS stores of a wave
    //      fall on consecutive addresses of one V^T row
    for (int x = tid; x < npg * kPage * 8; x += 512) {
        const int t = x % (npg * kPage), dc = x / (npg * kPage);          // frame, d chunk (8 values)
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0;
        if (t < T) v = ld16<bf16x8>(p.qkv + (row0 + t) * ld + 2 * p.C + h * 64 + dc * 8);
        const int pg = t >> 5, tk = t & 31;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = dc * 8 + e;
            vres[pg * (64 * kPage) + d * kPage + (((tk >> 3) ^ ((d >> 2) & 3)) << 3) + (tk & 7)] = (bf16_t)v[e];
        }
    }
    wait_vmem();
    sync();

    constexpr float kMasked = -1.0e30f;
    const int ntile = (T + 15) >> 4;
    for (int qt = w; qt < ntile; qt += 8) {                    // wave-uniform: no barrier below
        const int qw0 = qt * 16;
        int qi = qw0 + l15;
        if (qi > T - 1) qi = T - 1;
        const bf16_t* qr = p.qkv + (row0 + qi) * ld + h * 64 + g * 16;
        bf16x8 qB[2];
        qB[0] = ld16<bf16x8>(qr);
        qB[1] = ld16<bf16x8>(qr + 8);
        float m = kMasked, lsum = 0.f;                          // running maximum of query l15 (same in its 4 lanes), lane-partial denominator
        f32x4 oacc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) oacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int pg = 0; pg < npg; ++pg) {
            const bf16_t* kb = kres + pg * (kPage * 64);
            const bf16_t* vb = vres + pg * (64 * kPage);
            float s[8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = u * 16 + l15;
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                a = mfma16_op<F16>(ld16<bf16x8>(kb + r * 64 + (((2 * g) ^ (r & 7)) << 3)), qB[0], a);
                a = mfma16_op<F16>(ld16<bf16x8>(kb + r * 64 + (((2 * g + 1) ^ (r & 7)) << 3)), qB[1], a);
                // scores stay in RAW units (q.k, not yet times 1/8): the scale rides in the exponent's constant below.  Only the
                // last page can hold frames past T (wave-uniform test: full pages skip the compare / select)
                if (pg + 1 == npg) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) s[u * 4 + rr] = pg * kPage + u * 16 + g * 4 + rr < T ? a[rr] : kMasked;
                } else {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) s[u * 4 + rr] = a[rr];
                }
            }
            float tm = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
            tm = fmaxf(tm, shfl_xor(tm, 16));
            tm = fmaxf(tm, shfl_xor(tm, 32));                  // this page's maximum for query l15
            const float mn = fmaxf(m, tm);
            // exp((s - m) / 8) = exp2((s - m) * log2(e) / 8): one subtract, one multiply, one v_exp_f32 per score (~2e-6 relative:
            // the result is rounded to bf16 for the PV product, and the codec's bar is a waveform tolerance against an fp32
            // reference -- the backbone's attention keeps the ~1.5-ulp fexp_neg)
            constexpr float kC = 0.125f * 1.44269504088896340736f;
            const float alpha = fexp2((m - mn) * kC);   // first page: exp2(-huge) = 0 on zeroed accumulators
            m = mn;
            bf16x8 pA;
            float add = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pe = fexp2((s[e] - mn) * kC);
                add += pe;
                pA[e] = (short)f2op<F16>(pe);
            }
            lsum = lsum * alpha + add;
            // accumulator rows are queries g*4 + r: their factors live in the lanes whose l15 is that query
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ar = shfl(alpha, g * 4 + r);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) oacc[nt][r] *= ar;
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int d = nt * 16 + l15;
                const int sw = (d >> 2) & 3;
                const bf16x4 v0 = ld16<bf16x4>(vb + d * kPage + (((g >> 1) ^ sw) << 3) + (g & 1) * 4);
                const bf16x4 v1 = ld16<bf16x4>(vb + d * kPage + (((2 + (g >> 1)) ^ sw) << 3) + (g & 1) * 4);
                bf16x8 vB;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vB[e] = v0[e]; vB[4 + e] = v1[e]; }
                oacc[nt] = mfma16_op<F16>(pA, vB, oacc[nt]);
            }
        }
        lsum += shfl_xor(lsum, 16);
        lsum += shfl_xor(lsum, 32);                            // denominator of query l15
        const float rl = 1.0f / lsum;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rr = shfl(rl, g * 4 + r);
            const int q = qw0 + g * 4 + r;
            if (q < T) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) put_op(p.out, row0 + q, p.C, h * 64 + l15 + nt * 16, oacc[nt][r] * rr, p.split);
            }
        }
    }
}

// ---- ISTFT head: spectrum -> hi/lo-split bf16 DFT operand ------------------------------------------------
// spec fp32 [rows][lds]: log-magnitude bins [0, nb) | phase bins [nb, 2nb), nb = n_fft/2 + 1.
// s3 bf16 [rows][K3]: [re_hi | im_hi | re_lo | im_lo | re_hi | im_hi | 0...]  (pairs with basis [B_hi | B_hi | B_lo])
struct IstftPrepArgs {
    const float* spec;
    long lds;
    bf16_t* s3;
    long K3;
    long rows;
    int nb;
    int f16;               // precision = fp16 (round 6): s3 = [re | im | 0...] as IEEE halves, ONE term per bin (K3 = 2 nb padded) -- against the basis
                           // [B | ...] in halves scaled by kDftScale (codec.cpp); the bf16 engines keep the three-term split below
};
constexpr float kDftScale = 512.0f;   // fp16 DFT basis = window * ck * cos / N * 2^9 (largest entry 0.53; entries below 2^-14 / 2^9 = 1.2e-7 -- 1e-4 of the largest --
                                      // are the only ones in the subnormal range); ola_kernel multiplies by 2^-9 (exact)
NTTS_KERNEL(256) void istft_prep_kernel(IstftPrepArgs p) {
    const long r = blockIdx.x;
    const float* x = p.spec + r * p.lds;
    bf16_t* o = p.s3 + r * p.K3;
    for (int k = threadIdx.x; k < p.nb; k += 256) {
        float mag = fexp(x[k]);
        if (mag > 100.f) mag = 100.f;              // hf:...modeling_xcodec2.py:771
        const float ph = x[p.nb + k];
        const float re = mag * cosf(ph), im = mag * sinf(ph);
        if (p.f16) { o[k] = f2h(re); o[p.nb + k] = f2h(im); continue; }
        const bf16_t rh = f2bf(re), ih = f2bf(im);
        const bf16_t rl = f2bf(re - bf2f(rh)), il = f2bf(im - bf2f(ih));
        o[k] = rh; o[p.nb + k] = ih;
        o[2 * p.nb + k] = rl; o[3 * p.nb + k] = il;
        o[4 * p.nb + k] = rh; o[5 * p.nb + k] = ih;
    }
    for (long k = (p.f16 ? 2L : 6L) * p.nb + threadIdx.x; k < p.K3; k += 256) o[k] = 0;
}

// ---- overlap-add of the windowed frames, trim, divide by the window envelope -------------------------------
struct OlaArgs {
    const float* frames;   // [rows][n_fft]  (window already folded into the DFT basis)
    const float* win2;     // [n_fft] window^2
    float* wav;            // [B][wav_stride]
    long wav_stride;
    CodecRows R;
    int hop, n_fft;
    float scale;           // 1, or 1 / kDftScale when the frames come from the fp16 DFT basis
};
// grid (B, ceil(hop*Tmax / 256))
NTTS_KERNEL(256) void ola_kernel(OlaArgs p) {
    const int b = blockIdx.x;
    const int T = p.R.lens[b];
    const long s = (long)blockIdx.y * 256 + threadIdx.x;
    if (s >= (long)p.hop * T) return;
    const int pad = (p.n_fft - p.hop) / 2;
    const long pos = s + pad;                       // index in the untrimmed overlap-add buffer
    const long row0 = codec_row0(p.R, b);
    int f_hi = (int)(pos / p.hop);
    if (f_hi > T - 1) f_hi = T - 1;
    float acc = 0.f, env = 0.f;
    for (int f = f_hi; f >= 0; --f) {
        const long n = pos - (long)f * p.hop;
        if (n >= p.n_fft) break;
        acc += p.frames[(row0 + f) * p.n_fft + n];
        env += p.win2[n];
    }
    if (env < 1e-11f) env = 1e-11f;                 // hf:...modeling_xcodec2.py:793
    p.wav[(long)b * p.wav_stride + s] = acc * p.scale / env;
}

}  // namespace ntts
