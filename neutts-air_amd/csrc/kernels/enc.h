// enc.h -- kernels of the reference-encoding path (NeuCodec encoder), fp32, activations channels-last [T][C].
//
// Replaces  codec.encode_code(wav16k[1,1,L]) -> codes[1,1,T]  (ref:neutts/neutts.py:266-271), i.e. op by op
//   SeamlessM4TFeatureExtractor (kaldi fbank)         hf:models/seamless_m4t/feature_extraction_seamless_m4t.py:115-140,254-290
//   Wav2Vec2BertModel, conformer layers 1..16         hf:models/wav2vec2_bert/modeling_wav2vec2_bert.py:126-131,147-154,196-226,263-337,423-461
//   Xcodec2SemanticAdapter / Xcodec2Encoder           hf:models/xcodec2/modeling_xcodec2.py:899-908, :400-413,:463-545,:561-636
//   fc_encoder + Xcodec2Quantizer (FSQ)               hf:models/xcodec2/modeling_xcodec2.py:1008-1016, :703-743,:811-818
// One-off per speaker and OFF the synthesis hot path: it computes in fp32 (the reference does; the output is INTEGER codes whose
// rounding boundaries a bf16 pipeline would cross for ~10 % of the frames), every Linear / Conv1d on the fp32 matrix core
// (v_mfma_f32_16x16x4_f32) through one implicit-GEMM kernel, the rest in small element / row kernels.
#pragma once
#include <ntts/dev.h>

namespace ntts {

constexpr int kFbFrame = 400, kFbShift = 160, kFbFft = 512, kFbBins = 257, kFbMel = 80;   // 25 ms / 10 ms at 16 kHz
constexpr int kRelQB = 4, kRelMaxT = 3072, kRelMaxPos = 128;                                // rel_attn_kernel limits (61 s of audio)

// ---------------------------------------------------------------------------------------------------------------------
// Y[m][n] = act( sum_k A(m, k) * W[n][k] + bias[n] ) * alpha + resid[m][n]
// A is the im2col view of a channels-last input:  A(m, tap * Cin + c) = X[m * stride + tap * dil - pad][c]  (0 outside [0, Tin));
// a Linear is taps = 1.  W is [N][K] row-major with k = tap * Cin + c (encoder.cpp repacks Conv1d weights once).
// 64 x 64 tile per 4-wave workgroup, K in steps of 16 through LDS (k-major, row stride 80 floats: the four k-groups of a
// fragment read land on disjoint bank halves), 2 x 2 MFMA 16x16x4 tiles per wave.
// ---------------------------------------------------------------------------------------------------------------------
struct SgemmArgs {
    const float* X; long ldx; int Tin, Cin, taps, dil, stride, pad;
    const float* W; long ldw;
    const float* bias;            // [N] or null
    const float* resid; long ldr; // [M][ldr] or null (may alias Y: each element is read then written by the same lane)
    float alpha;
    int act;                      // 0 none, 1 relu, 2 silu
    float* Y; long ldy;
    int M, N, K;
};

constexpr int kSgLd = 80;

NTTS_KERNEL(256) void sgemm_kernel(SgemmArgs p) {
    NTTS_SHARED float As[16 * kSgLd];
    NTTS_SHARED float Bs[16 * kSgLd];
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int lr = tid & 63, kq = (tid >> 6) * 4;      // this thread stages row lr, k offsets kq .. kq+3 of both tiles
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
    const int am = m0 + lr, bn = n0 + lr;
    const bool am_ok = am < p.M, bn_ok = bn < p.N;
    const long t0 = (long)am * p.stride - p.pad;
    const bool fast_a = ((p.Cin & 3) == 0) && ((p.ldx & 3) == 0);
    const bool fast_b = ((p.K & 3) == 0) && ((p.ldw & 3) == 0);
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < p.K; k0 += 16) {
        const int k = k0 + kq;
        float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (am_ok && k < p.K) {
            if (fast_a) {
                const int tap = k / p.Cin, c = k - tap * p.Cin;
                const long t = t0 + (long)tap * p.dil;
                if (t >= 0 && t < p.Tin) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(p.X + t * p.ldx + c);
                    av[0] = v[0]; av[1] = v[1]; av[2] = v[2]; av[3] = v[3];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kk = k + e;
                    if (kk < p.K) {
                        const int tap = kk / p.Cin, c = kk - tap * p.Cin;
                        const long t = t0 + (long)tap * p.dil;
                        if (t >= 0 && t < p.Tin) av[e] = p.X[t * p.ldx + c];
                    }
                }
            }
        }
        if (bn_ok && k < p.K) {
            if (fast_b) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p.W + (long)bn * p.ldw + k);
                bv[0] = v[0]; bv[1] = v[1]; bv[2] = v[2]; bv[3] = v[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k + e < p.K) bv[e] = p.W[(long)bn * p.ldw + k + e];
            }
        }
        sync();   // the previous step's fragment reads are done
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            As[(kq + e) * kSgLd + lr] = av[e];
            Bs[(kq + e) * kSgLd + lr] = bv[e];
        }
        sync();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kk = (ks * 4 + (lane >> 4)) * kSgLd + (lane & 15);
            const float a0 = As[kk + wm], a1 = As[kk + wm + 16];
            const float b0 = Bs[kk + wn], b1 = Bs[kk + wn + 16];
            acc[0][0] = mfma16_f32(a0, b0, acc[0][0]);
            acc[0][1] = mfma16_f32(a0, b1, acc[0][1]);
            acc[1][0] = mfma16_f32(a1, b0, acc[1][0]);
            acc[1][1] = mfma16_f32(a1, b1, acc[1][1]);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + j * 16 + (lane & 15);
            if (n >= p.N) continue;
            const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + (lane >> 4) * 4 + r;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + b;
                if (p.act == 1) v = v > 0.f ? v : 0.f;
                else if (p.act == 2) v = v / (1.f + expf(-v));
                v *= p.alpha;
                if (p.resid) v += p.resid[(long)m * p.ldr + n];
                p.Y[(long)m * p.ldy + n] = v;
            }
        }
}

inline void sgemm_launch(const SgemmArgs& p, hipStream_t s) {
    NTTS_LAUNCH((sgemm_kernel), dim3((p.N + 63) / 64, (p.M + 63) / 64), dim3(256), s, p);
}

// ---------------------------------------------------------------------------------------------------------------------
// nn.LayerNorm over the last dimension (biased variance, eps inside the sqrt), optional SiLU after it (the conformer's
// depthwise_layer_norm -> swish).  One wave64 per row.
// ---------------------------------------------------------------------------------------------------------------------
struct LayerNormArgs { const float* X; long ldx; float* Y; long ldy; const float* w; const float* b; float eps; int M, C, act; };

NTTS_KERNEL(256) void enc_layernorm_kernel(LayerNormArgs p) {
    const int lane = lane_id();
    int r = blockIdx.x * 4 + wave_id();
    const bool rok = r < p.M;
    if (!rok) r = p.M - 1;                         // keep every lane in the shuffles
    const float* x = p.X + (long)r * p.ldx;
    float s = 0.f;
    for (int c = lane; c < p.C; c += 64) s += x[c];
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) s += shfl_xor(s, sh);
    const float mean = s / (float)p.C;
    float q = 0.f;
    for (int c = lane; c < p.C; c += 64) { const float d = x[c] - mean; q += d * d; }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) q += shfl_xor(q, sh);
    const float rstd = 1.0f / sqrtf(q / (float)p.C + p.eps);
    if (!rok) return;
    float* y = p.Y + (long)r * p.ldy;
    for (int c = lane; c < p.C; c += 64) {
        float v = (x[c] - mean) * rstd * p.w[c] + p.b[c];
        if (p.act == 2) v = v / (1.f + expf(-v));
        y[c] = v;
    }
}

// nn.GLU(dim = channels): Y[t][c] = X[t][c] * sigmoid(X[t][C + c])
NTTS_KERNEL(256) void enc_glu_kernel(const float* X, long ldx, float* Y, long ldy, int T, int C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)T * C) return;
    const int c = (int)(idx % C);
    const long t = idx / C;
    const float a = X[t * ldx + c], g = X[t * ldx + C + c];
    Y[t * ldy + c] = a * (1.f / (1.f + expf(-g)));
}

// causal depthwise Conv1d (groups = C, all padding on the left): Y[t][c] = sum_j W[c][j] * X[t - (k-1) + j][c]
NTTS_KERNEL(256) void enc_dwconv_kernel(const float* X, long ldx, const float* W, float* Y, long ldy, int T, int C, int ksize) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)T * C) return;
    const int c = (int)(idx % C);
    const long t = idx / C;
    float acc = 0.f;
    for (int j = 0; j < ksize; ++j) {
        const long ts = t - (ksize - 1) + j;
        if (ts >= 0) acc = fmaf(W[(long)c * ksize + j], X[ts * ldx + c], acc);
    }
    Y[t * ldy + c] = acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// Conformer self-attention with "relative_key" positions: softmax_j( (q_i . k_j + q_i . E[clamp(j - i, -L, R) + L]) / sqrt(hd) ) V
// One workgroup = kRelQB queries of one head; scores of the whole sequence live in LDS (T <= kRelMaxT).
// ---------------------------------------------------------------------------------------------------------------------
struct RelAttnArgs { const float* qkv; long ld; const float* dist_emb; float* out; long ldo; int T, nh, hd, left, right; float scale; };

NTTS_KERNEL(256) void enc_rel_attn_kernel(RelAttnArgs p) {
    NTTS_SHARED float q[kRelQB][64];
    NTTS_SHARED float qe[kRelQB][kRelMaxPos];
    NTTS_SHARED float sc[kRelQB][kRelMaxT];
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int h = blockIdx.y, i0 = blockIdx.x * kRelQB;
    const int H = p.nh * p.hd, npos = p.left + p.right + 1;
    {
        const int qi = tid >> 6, d = tid & 63, i = i0 + qi;
        q[qi][d] = (d < p.hd && i < p.T) ? p.qkv[(long)i * p.ld + h * p.hd + d] * p.scale : 0.f;
    }
    sync();
    for (int idx = tid; idx < kRelQB * npos; idx += 256) {
        const int qi = idx / npos, pp = idx - qi * npos;
        float a = 0.f;
        for (int d = 0; d < p.hd; ++d) a = fmaf(q[qi][d], p.dist_emb[(long)pp * p.hd + d], a);
        qe[qi][pp] = a;
    }
    sync();
    for (int j = tid; j < p.T; j += 256) {
        const float* kr = p.qkv + (long)j * p.ld + H + h * p.hd;
        float s[kRelQB];
#pragma unroll
        for (int qi = 0; qi < kRelQB; ++qi) s[qi] = 0.f;
        for (int d = 0; d < p.hd; ++d) {
            const float kv = kr[d];
#pragma unroll
            for (int qi = 0; qi < kRelQB; ++qi) s[qi] = fmaf(q[qi][d], kv, s[qi]);
        }
#pragma unroll
        for (int qi = 0; qi < kRelQB; ++qi) {
            int dd = j - (i0 + qi);
            dd = dd < -p.left ? -p.left : (dd > p.right ? p.right : dd);
            sc[qi][j] = s[qi] + qe[qi][dd + p.left];
        }
    }
    sync();
    {   // softmax of query w by wave w
        float mx = -INFINITY;
        for (int j = lane; j < p.T; j += 64) mx = fmaxf(mx, sc[w][j]);
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) mx = fmaxf(mx, shfl_xor(mx, sh));
        float sum = 0.f;
        for (int j = lane; j < p.T; j += 64) { const float e = expf(sc[w][j] - mx); sc[w][j] = e; sum += e; }
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) sum += shfl_xor(sum, sh);
        for (int j = lane; j < p.T; j += 64) sc[w][j] = sc[w][j] / sum;
    }
    sync();
    {
        const int qi = tid >> 6, d = tid & 63, i = i0 + qi;
        if (d < p.hd && i < p.T) {
            const float* vc = p.qkv + 2 * H + h * p.hd + d;
            float a = 0.f;
            for (int j = 0; j < p.T; ++j) a = fmaf(sc[qi][j], vc[(long)j * p.ld], a);
            p.out[(long)i * p.ldo + h * p.hd + d] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Anti-aliased SnakeBeta: 2x up-sample (replicate pad 5, 12-tap Kaiser-sinc transposed conv x 2, crop 15 / 15), the
// activation u + sin^2(u e^alpha) / (e^beta + 1e-9), then replicate pad (5, 6) + the same 12 taps at stride 2 -- fused, one
// output sample per thread:  out[t] = sum_j f[j] * snake( u[clamp(2t + j - 5, 0, 2T - 1)] ),
//                            u[n]   = 2 * sum_i x[clamp(i - 5, 0, T - 1)] * f[n + 15 - 2 i],  i = ceil((n + 4) / 2) .. + 5
// ---------------------------------------------------------------------------------------------------------------------
struct SnakeArgs { const float* X; long ldx; float* Y; long ldy; const float* ea; const float* inv_b; int T, C; float f[12]; };

NTTS_KERNEL(256) void enc_snake_aa_kernel(SnakeArgs p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)p.T * p.C) return;
    const int c = (int)(idx % p.C);
    const int t = (int)(idx / p.C);
    const float a = p.ea[c], ib = p.inv_b[c];
    const int n_hi = 2 * p.T - 1;
    float out = 0.f;
    if (2 * t - 5 >= 0 && 2 * t + 6 <= n_hi) {
        // interior: no clamp is active; up-sampled sample n = 2t + j - 5 reads x[t + (j >> 1) + ii - 5], tap (j & 1) + 10 - 2 ii
        float xr[11];
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            int ts = t - 5 + i;
            ts = ts < 0 ? 0 : (ts > p.T - 1 ? p.T - 1 : ts);     // (the input row itself may still be replicate-padded)
            xr[i] = p.X[(long)ts * p.ldx + c];
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            float u = 0.f;
#pragma unroll
            for (int ii = 0; ii < 6; ++ii) u = fmaf(xr[(j >> 1) + ii], p.f[(j & 1) + 10 - 2 * ii], u);
            u *= 2.f;
            const float sn = sinf(u * a);
            out = fmaf(p.f[j], u + ib * (sn * sn), out);
        }
    } else {
        for (int j = 0; j < 12; ++j) {
            int n = 2 * t + j - 5;
            n = n < 0 ? 0 : (n > n_hi ? n_hi : n);
            const int i_lo = (n + 5) >> 1;                       // ceil((n + 4) / 2)
            float u = 0.f;
            for (int ii = 0; ii < 6; ++ii) {
                const int i = i_lo + ii;
                int ts = i - 5;
                ts = ts < 0 ? 0 : (ts > p.T - 1 ? p.T - 1 : ts);
                u = fmaf(p.X[(long)ts * p.ldx + c], p.f[n + 15 - 2 * i], u);
            }
            u *= 2.f;
            const float sn = sinf(u * a);
            out = fmaf(p.f[j], u + ib * (sn * sn), out);
        }
    }
    p.Y[(long)t * p.ldy + c] = out;
}

// ---------------------------------------------------------------------------------------------------------------------
// Kaldi-style log-mel filter bank of one 25 ms frame per workgroup: x * 2^15 -> minus frame mean -> pre-emphasis 0.97 ->
// povey window -> 512-point DFT power -> 80 triangular mel filters (floor 1.19e-7) -> ln.
// The input is the hop-padded clip with 160 zeros on both sides (the reference pads before calling the extractor).
// ---------------------------------------------------------------------------------------------------------------------
struct FbankArgs { const float* wav; long n_wav; const float* window; const float* tw; const float* melf; float* logmel; int nfr; };

NTTS_KERNEL(256) void enc_fbank_kernel(FbankArgs p) {
    NTTS_SHARED float fr[kFbFft];
    NTTS_SHARED float tws[kFbFft * 2];
    NTTS_SHARED float pw[kFbBins + 3];
    NTTS_SHARED float red[4];
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const long base = (long)blockIdx.x * kFbShift - kFbShift;      // first sample of this frame in the un-padded clip
    float x[2];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int n = tid + e * 256;
        const long g = base + n;
        x[e] = (n < kFbFrame && g >= 0 && g < p.n_wav) ? p.wav[g] * 32768.0f : 0.f;
        s += x[e];
    }
    for (int i = tid; i < kFbFft * 2; i += 256) tws[i] = p.tw[i];
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) s += shfl_xor(s, sh);
    if (lane == 0) red[w] = s;
    sync();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)kFbFrame;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int n = tid + e * 256;
        fr[n] = n < kFbFrame ? x[e] - mean : 0.f;
    }
    sync();
    float y[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int n = tid + e * 256;
        y[e] = 0.f;
        if (n < kFbFrame) y[e] = (n == 0 ? fr[0] * (1.0f - 0.97f) : fr[n] - 0.97f * fr[n - 1]) * p.window[n];
    }
    sync();
#pragma unroll
    for (int e = 0; e < 2; ++e) fr[tid + e * 256] = y[e];
    sync();
    for (int k = tid; k < kFbBins; k += 256) {
        float re = 0.f, im = 0.f;
        for (int n = 0; n < kFbFrame; ++n) {
            const int i = (k * n) & (kFbFft - 1);
            re = fmaf(fr[n], tws[2 * i], re);
            im = fmaf(fr[n], tws[2 * i + 1], im);
        }
        pw[k] = re * re + im * im;
    }
    sync();
    if (tid < kFbMel) {
        float m = 0.f;
        for (int k = 0; k < kFbBins; ++k) m = fmaf(p.melf[k * kFbMel + tid], pw[k], m);
        p.logmel[(long)blockIdx.x * kFbMel + tid] = logf(fmaxf(m, 1.192092955078125e-07f));
    }
}

// per-mel-bin (x - mean) / sqrt(var_ddof1 + 1e-7) over the clip's frames, frame pairs stacked: feats[f / 2][(f & 1) * 80 + m]
NTTS_KERNEL(256) void enc_melnorm_kernel(const float* logmel, float* feats, int nfr, int nfr_even) {
    NTTS_SHARED float red[4];
    const int m = blockIdx.x, tid = threadIdx.x, lane = lane_id(), w = wave_id();
    float s = 0.f;
    for (int f = tid; f < nfr; f += 256) s += logmel[(long)f * kFbMel + m];
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) s += shfl_xor(s, sh);
    if (lane == 0) red[w] = s;
    sync();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)nfr;
    sync();
    float q = 0.f;
    for (int f = tid; f < nfr; f += 256) { const float d = logmel[(long)f * kFbMel + m] - mean; q += d * d; }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) q += shfl_xor(q, sh);
    if (lane == 0) red[w] = q;
    sync();
    const float var = (red[0] + red[1] + red[2] + red[3]) / (float)(nfr - 1);
    const float rs = 1.0f / sqrtf(var + 1e-7f);
    for (int f = tid; f < nfr_even; f += 256)
        feats[(long)(f >> 1) * (2 * kFbMel) + (f & 1) * kFbMel + m] = (logmel[(long)f * kFbMel + m] - mean) * rs;
}

// ---------------------------------------------------------------------------------------------------------------------
// FSQ: bound twice (the quantizer bounds, then the FSQ module bounds again), round half-to-even, digits -> one index
// ---------------------------------------------------------------------------------------------------------------------
struct FsqArgs { const float* z; long ldz; float* lat; int* codes; int T, n; int levels[8]; float half_range[8], offset[8], shift[8]; };

NTTS_KERNEL(256) void enc_fsq_kernel(FsqArgs p) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= p.T) return;
    int code = 0, basis = 1;
    for (int d = 0; d < p.n; ++d) {
        float v = p.z[(long)t * p.ldz + d];
        v = tanhf(v + p.shift[d]) * p.half_range[d] - p.offset[d];
        v = tanhf(v + p.shift[d]) * p.half_range[d] - p.offset[d];
        p.lat[(long)t * p.n + d] = v;
        code += ((int)rintf(v) + p.levels[d] / 2) * basis;
        basis *= p.levels[d];
    }
    p.codes[t] = code;
}

}  // namespace ntts
