// codec.cpp -- NeuCodec decoder engine behind include/neutts_hip.h (ntts_codec_*).
//
// Replaces  self.codec.decode_code(codes[1,1,T]) -> wav[1,1,480*T]  (ref:neutts/neutts.py:288-291) for a batch
// of utterances: one non-autoregressive pass, all Linear/Conv1d/DFT work on the MFMA GEMM (gemm.h), the rest
// in kernels/codec.h.  Weights keep the parameter names of transformers' Xcodec2Model.
#include <ntts/dev.h>

#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/neutts_hip.h"
#include "kernels/codec.h"
#include "kernels/gemm.h"

using namespace ntts;

static std::string g_codec_create_err;

struct ResW { float *g1, *b1, *cb1, *g2, *b2, *cb2; bf16_t *w1, *w2; };
struct CLayerW { float *ln1, *ln2; bf16_t *wqkv, *wo, *fc1, *fc2; };

struct ntts_codec {
    ntts_codec_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;      // where the passes run: the engine's own stream or one the caller lent (ntts_codec_set_stream)
    hipStream_t own_stream = nullptr;
    std::string err;
    int H = 0, I = 0, nq = 0, n_fft = 0, nb = 0, NS = 0, lds_spec = 0;
    long K3 = 0, max_rows = 0;
    bool finalized = false;
    int S = 1;                   // 3 with precision = high: every bf16 GEMM operand is a split row [hi | lo | hi] of 3 x its columns (codec.h put_op)
    int fmt = kOpF16;            // operand format the non-GEMM kernels write (codec.h put_op): kOpF16 (default) / kOpSplit (high) / kOpBf16
    std::map<std::string, std::vector<float>> host;            // staged fp32 tensors until finalize
    std::map<std::string, std::vector<int64_t>> shapes;
    std::vector<void*> allocs;
    // device weights
    float *wf = nullptr, *bf = nullptr, *embed_b = nullptr, *fn_w = nullptr, *fn_b = nullptr, *head_b = nullptr, *win2 = nullptr;
    bf16_t *embed_w = nullptr, *head_w = nullptr, *basis3 = nullptr;
    ResW res[4]{};
    std::vector<CLayerW> layers;
    // workspaces
    float *h = nullptr, *t1 = nullptr, *spec = nullptr, *frames = nullptr, *wav = nullptr;
    bf16_t *xa = nullptr, *xb = nullptr, *qkv = nullptr, *act = nullptr, *vt = nullptr, *s3 = nullptr;
    int* meta = nullptr;
    size_t meta_cap = 0;
    // page-locked staging ring of the meta block (as backbone.cpp upload_meta): the copy from a pageable vector needed a stream-wide
    // synchronisation, which on a LENT stream (an engine gang's lane with a whole decode phase queued) blocked the launching thread
    static constexpr int kMetaStages = 2;
    int* meta_host[kMetaStages] = {nullptr, nullptr};
    hipEvent_t meta_ev[kMetaStages] = {nullptr, nullptr};
    bool meta_used[kMetaStages] = {false, false};
    int meta_next = 0;
    size_t wav_cap = 0;
    hipEvent_t ev[2]{};
    hipEvent_t ev_in = nullptr;   // orders a pass behind the stream that produced its device-side codes
    hipEvent_t ev_done = nullptr; // behind the asynchronous hand-over of the most recent pass (what ntts_codec_sync waits for on a lent stream)
    bool have_done = false;
    bool have_time = false;
    bool gn_reg = true;          // GroupNorm with the utterance slice in registers when it fits (NTTS_CODEC_GN_REG=0: the two-pass kernel)
    // per-stage taps of the residual stream (ntts_codec_set_debug / ntts_codec_read_stage: the error-budget tests): fp32 [4][max_rows][H]
    float* tap = nullptr;
    std::vector<int> tap_off, tap_len;   // row offset / frames of every utterance of the most recent decode call
    bool attn_resident = true;   // utterances of up to 256 frames: attn_full_resident_kernel (NTTS_CODEC_ATTN_RESIDENT=0: the two-sweep paged kernel)
};

static int cfail(ntts_codec* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_codec_create_err = buf;
    return code;
}
#define CHIP(c, call)                                                                                  \
    do {                                                                                               \
        hipError_t _s = (call);                                                                        \
        if (_s != hipSuccess) return cfail(c, _s == 2 ? NTTS_ENOMEM : NTTS_EHIP, "%s failed: %s", #call, hipGetErrorString(_s)); \
    } while (0)

extern "C" const char* ntts_codec_last_error(const ntts_codec* c) { return c ? c->err.c_str() : g_codec_create_err.c_str(); }

template <typename T>
static int dalloc(ntts_codec* c, T** p, size_t n) {
    void* v = nullptr;
    CHIP(c, hipMalloc(&v, n * sizeof(T)));
    c->allocs.push_back(v);
    *p = (T*)v;
    return NTTS_OK;
}

extern "C" void ntts_codec_destroy(ntts_codec* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (void* p : c->allocs) hipFree(p);
    if (c->tap) hipFree(c->tap);
    for (auto& e : c->ev)
        if (e) hipEventDestroy(e);
    if (c->ev_in) hipEventDestroy(c->ev_in);
    if (c->ev_done) hipEventDestroy(c->ev_done);
    for (int i = 0; i < ntts_codec::kMetaStages; ++i) {
        if (c->meta_host[i]) hipHostFree(c->meta_host[i]);
        if (c->meta_ev[i]) hipEventDestroy(c->meta_ev[i]);
    }
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" int ntts_codec_create(const ntts_codec_config* cf, int device, ntts_codec** out) {
    if (!cf || !out) return cfail(nullptr, NTTS_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device || device < 0)
        return cfail(nullptr, NTTS_ENODEV, "no HIP device %d (found %d): this library has no CPU fallback", device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return cfail(nullptr, NTTS_ENODEV, "device %d is '%s', kernels are built for gfx950 only", device, prop.gcnArchName);
    if (cf->head_dim != 64 || cf->num_heads * 64 != cf->hidden_size) return cfail(nullptr, NTTS_EINVAL, "need head_dim 64 and heads*64 == hidden");
    if (cf->hidden_size % 64 || cf->intermediate_size % 64 || cf->hidden_size > 2048 || (cf->hidden_size / 32) > 256 || 256 % (cf->hidden_size / 32))
        return cfail(nullptr, NTTS_EINVAL, "unsupported hidden/intermediate size");
    if (cf->n_levels < 1 || cf->n_levels > 8 || cf->hop_length < 8 || (cf->hop_length % 8)) return cfail(nullptr, NTTS_EINVAL, "bad levels / hop");
    if (cf->max_frames < 1 || cf->max_rows < cf->max_frames + 2 * kPadRows) return cfail(nullptr, NTTS_EINVAL, "bad max_frames / max_rows");
    ntts_codec* c = new ntts_codec();
    c->cfg = *cf;
    c->device = device;
    c->H = cf->hidden_size; c->I = cf->intermediate_size; c->nq = cf->n_levels;
    c->n_fft = cf->hop_length * 4; c->nb = c->n_fft / 2 + 1; c->NS = c->n_fft + 2;
    c->lds_spec = (c->NS + 3) / 4 * 4;
    c->K3 = ((cf->precision == 0 ? 2L : 6L) * c->nb + 63) / 64 * 64;      // fp16: one term per (re, im) bin; bf16 engines: the hi / lo split, three terms
    c->max_rows = cf->max_rows;
    if (cf->precision < 0 || cf->precision > 2) { delete c; return cfail(nullptr, NTTS_EINVAL, "unknown codec precision %d (0 = fp16 operands, 1 = high: split bf16 operands, 2 = bf16 operands)", cf->precision); }
    c->S = cf->precision == 1 ? 3 : 1;
    c->fmt = cf->precision == 0 ? kOpF16 : cf->precision == 1 ? kOpSplit : kOpBf16;
    { const char* ev = getenv("NTTS_CODEC_ATTN_RESIDENT"); if (ev && ev[0] == '0') c->attn_resident = false; }
    { const char* ev = getenv("NTTS_CODEC_GN_REG"); if (ev && ev[0] == '0') c->gn_reg = false; }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return cfail(nullptr, NTTS_EHIP, "stream creation failed");
    }
    c->stream = c->own_stream;
    hipEventCreate(&c->ev[0]); hipEventCreate(&c->ev[1]); hipEventCreate(&c->ev_in);
    hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
    const size_t R = c->max_rows, H = c->H;
    const int npages_max = (cf->max_frames + kPage - 1) / kPage;
    const size_t max_utts = R / (1 + 2 * kPadRows) + 1;
    int rc = NTTS_OK;
#define A(call) if (rc == NTTS_OK) rc = (call)
    const size_t S = c->S;        // (split operands: 3 x the columns)
    A(dalloc(c, &c->h, R * H)); A(dalloc(c, &c->t1, R * H)); A(dalloc(c, &c->xa, R * H * S)); A(dalloc(c, &c->xb, R * H * S));
    A(dalloc(c, &c->qkv, R * 3 * H)); A(dalloc(c, &c->act, R * c->I * S));
    A(dalloc(c, &c->vt, (R + (size_t)max_utts * kPage) * H));
    A(dalloc(c, &c->spec, R * c->lds_spec)); A(dalloc(c, &c->s3, R * c->K3)); A(dalloc(c, &c->frames, R * c->n_fft));
    c->wav_cap = R * cf->hop_length;
    A(dalloc(c, &c->wav, c->wav_cap));
    c->meta_cap = R + 3 * max_utts + 64;
    A(dalloc(c, &c->meta, c->meta_cap));
    for (int i = 0; i < ntts_codec::kMetaStages && rc == NTTS_OK; ++i)
        if (hipHostMalloc((void**)&c->meta_host[i], c->meta_cap * sizeof(int), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&c->meta_ev[i], hipEventDisableTiming) != hipSuccess) rc = cfail(nullptr, NTTS_ENOMEM, "page-locked meta staging");
    (void)npages_max;
    if (rc == NTTS_OK) {
        hipMemset(c->h, 0, R * H * 4); hipMemset(c->t1, 0, R * H * 4);
        hipMemset(c->xa, 0, R * H * S * 2); hipMemset(c->xb, 0, R * H * S * 2);
        hipMemset(c->qkv, 0, R * 3 * H * 2);
        hipDeviceSynchronize();
    } else {
        g_codec_create_err = c->err;
        ntts_codec_destroy(c);
        return rc;
    }
    *out = c;
    return NTTS_OK;
}

extern "C" int ntts_codec_load_tensor(ntts_codec* c, const char* name, const void* data, int dtype, const int64_t* shape,
                                      int ndim, int is_device) {
    if (!c || !name || !data || !shape || ndim < 1 || ndim > 3) return cfail(c, NTTS_EINVAL, "bad argument");
    if (c->finalized) return cfail(c, NTTS_ESTATE, "weights already finalised");
    if (dtype != NTTS_DT_F32 && dtype != NTTS_DT_BF16) return cfail(c, NTTS_EINVAL, "tensor '%s': dtype must be f32 or bf16", name);
    CHIP(c, hipSetDevice(c->device));
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    const size_t esz = dtype == NTTS_DT_F32 ? 4 : 2;
    std::vector<unsigned char> raw(n * esz);
    if (is_device) CHIP(c, hipMemcpy(raw.data(), data, n * esz, hipMemcpyDeviceToHost));
    else memcpy(raw.data(), data, n * esz);
    std::vector<float> v(n);
    if (dtype == NTTS_DT_F32) memcpy(v.data(), raw.data(), n * 4);
    else
        for (size_t i = 0; i < n; ++i) {
            const uint32_t u = (uint32_t)((const uint16_t*)raw.data())[i] << 16;
            memcpy(&v[i], &u, 4);
        }
    c->host[name] = std::move(v);
    c->shapes[name] = std::vector<int64_t>(shape, shape + ndim);
    return NTTS_OK;
}

static bf16_t h_f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// float -> IEEE half, round-to-nearest-even (|f| <= 65504 checked by the caller)
static bf16_t h_f2h(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const bf16_t sgn = (bf16_t)((u >> 16) & 0x8000u);
    const float a = fabsf(f);
    if (!(a == a)) return sgn | 0x7e00;
    if (a < 6.103515625e-05f) return sgn | (bf16_t)nearbyintf(ldexpf(a, 24));      // subnormal halves: multiples of 2^-24
    int ex;
    const float fr = frexpf(a, &ex);
    float mant = nearbyintf(ldexpf(fr, 11));
    int e = ex - 1 + 15;
    if (mant == 2048.0f) { mant = 1024.0f; e += 1; }
    if (e >= 31) return sgn | 0x7bff;
    return sgn | (bf16_t)((e << 10) | ((int)mant - 1024));
}
static float h_bf2f(bf16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct Finalizer {
    ntts_codec* c;
    int rc = NTTS_OK;
    const std::vector<float>* get(const std::string& name, std::initializer_list<int64_t> shp) {
        auto it = c->host.find(name);
        if (it == c->host.end()) { rc = cfail(c, NTTS_ESTATE, "tensor '%s' not loaded", name.c_str()); return nullptr; }
        const auto& s = c->shapes[name];
        if (s.size() != shp.size() || !std::equal(s.begin(), s.end(), shp.begin())) {
            rc = cfail(c, NTTS_EINVAL, "tensor '%s': unexpected shape", name.c_str());
            return nullptr;
        }
        return &it->second;
    }
    float* up_f32(const std::vector<float>& v) {
        float* d = nullptr;
        if (dalloc(c, &d, v.size()) != NTTS_OK || hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = NTTS_EHIP; return nullptr; }
        return d;
    }
    // a GEMM weight in the engine's 16-bit operand format: bf16, or IEEE half (precision = fp16: the values must fit its range)
    bf16_t* up_bf16(const std::vector<float>& v) {
        std::vector<bf16_t> b(v.size());
        if (c->fmt == kOpF16) {
            for (size_t i = 0; i < v.size(); ++i) {
                if (!(fabsf(v[i]) <= 65504.0f)) { rc = cfail(c, NTTS_EINVAL, "a GEMM weight of magnitude %g does not fit fp16: create the engine with precision = 1 (high) or 2 (bf16)", (double)v[i]); return nullptr; }
                b[i] = h_f2h(v[i]);
            }
        } else
        for (size_t i = 0; i < v.size(); ++i) b[i] = h_f2bf(v[i]);
        bf16_t* d = nullptr;
        if (dalloc(c, &d, b.size()) != NTTS_OK || hipMemcpy(d, b.data(), b.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { rc = NTTS_EHIP; return nullptr; }
        return d;
    }
    // a GEMM weight [rows][K] whose X operand is made of segments of `seg` columns (the activation buffer's row width: K = seg for a
    // Linear, K = taps x seg for a conv over overlapping rows).  precision = high: every segment becomes [wh | wh | wl] (3 seg columns),
    // the counterpart of the split activation row [xh | xl | xh] (codec.h put_op): the K-loop forms xh wh + xl wh + xh wl
    bf16_t* op(const std::vector<float>& v, int64_t rows, int64_t K, int64_t seg) {
        if (c->S == 1) return up_bf16(v);
        std::vector<bf16_t> b((size_t)rows * 3 * K);
        for (int64_t r = 0; r < rows; ++r)
            for (int64_t s0 = 0; s0 < K; s0 += seg)
                for (int64_t i = 0; i < seg; ++i) {
                    const float w = v[(size_t)r * K + s0 + i];
                    const bf16_t hi = h_f2bf(w), lo = h_f2bf(w - h_bf2f(hi));
                    bf16_t* d = &b[(size_t)r * 3 * K + 3 * s0];
                    d[i] = hi; d[seg + i] = hi; d[2 * seg + i] = lo;
                }
        bf16_t* d = nullptr;
        if (dalloc(c, &d, b.size()) != NTTS_OK || hipMemcpy(d, b.data(), b.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { rc = NTTS_EHIP; return nullptr; }
        return d;
    }
    float* vec(const std::string& name, int64_t n) {
        auto* v = get(name, {n});
        return v ? up_f32(*v) : nullptr;
    }
    bf16_t* mat(const std::string& name, int64_t r, int64_t k) {
        auto* v = get(name, {r, k});
        return v ? op(*v, r, k, k) : nullptr;
    }
    // Conv1d weight [Cout][Cin][k] -> GEMM weight [Cout][k][Cin]  (row of the overlapping-rows GEMM = k frames x Cin)
    bf16_t* conv(const std::string& name, int64_t co, int64_t ci, int64_t k) {
        auto* v = get(name, {co, ci, k});
        if (!v) return nullptr;
        std::vector<float> r((size_t)co * k * ci);
        for (int64_t o = 0; o < co; ++o)
            for (int64_t i = 0; i < ci; ++i)
                for (int64_t t = 0; t < k; ++t) r[((size_t)o * k + t) * ci + i] = (*v)[((size_t)o * ci + i) * k + t];
        return op(r, co, k * ci, ci);
    }
};

extern "C" int ntts_codec_finalize(ntts_codec* c) {
    if (!c) return NTTS_EINVAL;
    if (c->finalized) return NTTS_OK;
    CHIP(c, hipSetDevice(c->device));
    Finalizer f{c};
    const int64_t H = c->H, I = c->I, Q = c->cfg.quantization_dim, nq = c->nq, NS = c->NS;
    // ---- fold quantizer.project_out (nq -> Q) and decoder.fc (Q -> H): no nonlinearity in between
    {
        auto* pw = f.get("quantizer.project_out.weight", {Q, nq});
        auto* pb = f.get("quantizer.project_out.bias", {Q});
        auto* fw = f.get("decoder.fc.weight", {H, Q});
        auto* fb = f.get("decoder.fc.bias", {H});
        if (!pw || !pb || !fw || !fb) return f.rc;
        std::vector<float> wf((size_t)H * nq), bf(H);
        for (int64_t o = 0; o < H; ++o) {
            double b = (*fb)[o];
            std::vector<double> acc(nq, 0.0);
            for (int64_t q = 0; q < Q; ++q) {
                const double w = (*fw)[(size_t)o * Q + q];
                b += w * (*pb)[q];
                for (int64_t i = 0; i < nq; ++i) acc[i] += w * (*pw)[(size_t)q * nq + i];
            }
            bf[o] = (float)b;
            for (int64_t i = 0; i < nq; ++i) wf[(size_t)o * nq + i] = (float)acc[i];
        }
        c->wf = f.up_f32(wf); c->bf = f.up_f32(bf);
    }
    c->embed_w = f.conv("decoder.embed.weight", H, H, 7);
    c->embed_b = f.vec("decoder.embed.bias", H);
    const char* nets[2] = {"prior_net", "post_net"};
    for (int n = 0; n < 2; ++n)
        for (int b = 0; b < 2; ++b) {
            const std::string p = std::string("decoder.") + nets[n] + "." + std::to_string(b) + ".";
            ResW& r = c->res[n * 2 + b];
            r.g1 = f.vec(p + "norm1.weight", H); r.b1 = f.vec(p + "norm1.bias", H);
            r.w1 = f.conv(p + "conv1.weight", H, H, 3); r.cb1 = f.vec(p + "conv1.bias", H);
            r.g2 = f.vec(p + "norm2.weight", H); r.b2 = f.vec(p + "norm2.bias", H);
            r.w2 = f.conv(p + "conv2.weight", H, H, 3); r.cb2 = f.vec(p + "conv2.bias", H);
        }
    c->layers.resize(c->cfg.num_layers);
    for (int i = 0; i < c->cfg.num_layers && f.rc == NTTS_OK; ++i) {
        const std::string p = "decoder.layers." + std::to_string(i) + ".";
        CLayerW& L = c->layers[i];
        L.ln1 = f.vec(p + "input_layernorm.weight", H);
        L.ln2 = f.vec(p + "post_attention_layernorm.weight", H);
        auto* q = f.get(p + "self_attn.q_proj.weight", {H, H});
        auto* k = f.get(p + "self_attn.k_proj.weight", {H, H});
        auto* v = f.get(p + "self_attn.v_proj.weight", {H, H});
        if (!q || !k || !v) return f.rc;
        std::vector<float> qkv;
        qkv.reserve((size_t)3 * H * H);
        qkv.insert(qkv.end(), q->begin(), q->end()); qkv.insert(qkv.end(), k->begin(), k->end()); qkv.insert(qkv.end(), v->begin(), v->end());
        L.wqkv = f.op(qkv, 3 * H, H, H);
        L.wo = f.mat(p + "self_attn.o_proj.weight", H, H);
        L.fc1 = f.mat(p + "mlp.fc1.weight", I, H);
        L.fc2 = f.mat(p + "mlp.fc2.weight", H, I);
    }
    c->fn_w = f.vec("decoder.norm.weight", H); c->fn_b = f.vec("decoder.norm.bias", H);
    c->head_w = f.mat("decoder.head.linear.weight", NS, H);
    c->head_b = f.vec("decoder.head.linear.bias", NS);
    if (f.rc != NTTS_OK) return f.rc;
    // ---- windowed inverse real DFT as a GEMM operand, split hi/lo so bf16 MFMA reaches ~fp32 accuracy:
    //      frame[n] = hann[n]/N * ( Re0 + (-1)^n Re_{N/2} + 2 sum_k (Re_k cos(2 pi k n / N) - Im_k sin(2 pi k n / N)) )
    //      (torch.fft.irfft norm="backward" + window, hf:...modeling_xcodec2.py:776-777); row layout [B_hi | B_hi | B_lo].
    {
        const int N = c->n_fft, nb = c->nb;
        const long K3 = c->K3;
        std::vector<bf16_t> B((size_t)N * K3, 0);
        std::vector<float> w2(N);
        for (int n = 0; n < N; ++n) {
            const double win = 0.5 - 0.5 * cos(2.0 * M_PI * n / N);   // torch.hann_window (periodic)
            w2[n] = (float)(win * win);
            for (int k = 0; k < nb; ++k) {
                const double ck = (k == 0 || k == N / 2) ? 1.0 : 2.0;
                const double ang = 2.0 * M_PI * (double)((long)k * n % N) / N;
                const float cr = (float)(win * ck * cos(ang) / N);
                const float ci = (k == 0 || k == N / 2) ? 0.f : (float)(-win * ck * sin(ang) / N);
                if (c->fmt == kOpF16) {       // ONE fp16 term per bin, basis scaled by 2^9 into fp16's normal range (codec.h kDftScale; ola_kernel takes it back)
                    bf16_t* row = &B[(size_t)n * K3];
                    row[k] = h_f2h(cr * kDftScale); row[nb + k] = h_f2h(ci * kDftScale);
                    continue;
                }
                const bf16_t crh = h_f2bf(cr), cih = h_f2bf(ci);
                const bf16_t crl = h_f2bf(cr - h_bf2f(crh)), cil = h_f2bf(ci - h_bf2f(cih));
                bf16_t* row = &B[(size_t)n * K3];
                row[k] = crh; row[nb + k] = cih;
                row[2 * nb + k] = crh; row[3 * nb + k] = cih;
                row[4 * nb + k] = crl; row[5 * nb + k] = cil;
            }
        }
        if (dalloc(c, &c->basis3, B.size()) != NTTS_OK) return NTTS_ENOMEM;
        CHIP(c, hipMemcpy(c->basis3, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        c->win2 = f.up_f32(w2);
    }
    if (f.rc != NTTS_OK) return f.rc;
    c->host.clear();
    c->finalized = true;
    return NTTS_OK;
}

static GemmArgs cg(const bf16_t* X, long ldx, const bf16_t* W, long K, const float* bias, void* out, long ldo, long M, int N,
                   const float* resid = nullptr, long ldr = 0) {
    GemmArgs a{};
    a.X = X; a.ldx = ldx; a.W = W; a.ldw = K; a.bias_f32 = bias; a.out = out; a.ldo = ldo; a.M = (int)M; a.N = N; a.K = (int)K;
    a.resid = resid; a.ldr = ldr;
    return a;
}

// big-M GEMM of the codec on the engine's operand format
#define CGEMM(EPI, ga, st) do { if (c->fmt == kOpF16) NTTS_GEMM_BIG_F16(EPI, ga, st); else NTTS_GEMM_BIG(EPI, ga, st); } while (0)

static void resnet_block(ntts_codec* c, const ResW& w, const CodecRows& R, long rows) {
    const int H = c->H;
    hipStream_t st = c->stream;
    GroupNormArgs g{};
    const long S = c->S;          // split operand rows: 3 x the columns, on the X side (ldx, K) and in the packed weights alike
    g.x = c->h; g.y = c->xa; g.gamma = w.g1; g.beta = w.b1; g.R = R; g.C = H; g.eps = 1e-6f; g.split = c->fmt;
    groupnorm_silu_launch(g, c->gn_reg ? R.Tp - 2 * kPadRows : (1 << 30), st);
    { GemmArgs ga_ = cg(c->xa, S * H, w.w1, 3L * S * H, w.cb1, c->t1 + H, H, rows - 2, H); CGEMM(EPI_F32, ga_, st); }
    g.x = c->t1; g.y = c->xb; g.gamma = w.g2; g.beta = w.b2;
    groupnorm_silu_launch(g, c->gn_reg ? R.Tp - 2 * kPadRows : (1 << 30), st);
    { GemmArgs ga_ = cg(c->xb, S * H, w.w2, 3L * S * H, w.cb2, c->h + H, H, rows - 2, H, c->h + H, H); CGEMM(EPI_F32, ga_, st); }
}

// codes: HOST packed codes (codes_dev == null), or DEVICE codes, utterance i at codes_dev + i * codes_stride (already in range:
// they come from ntts_backbone_export_codes).  out_kind: 0 = host destination, blocking (the classic entry point);
// 1 = host destination (pinned), asynchronous; 2 = device destination, asynchronous.  producer: a HIP stream whose work so far
// must complete before the codes are read (the backbone's stream), or null.
static int codec_decode_impl(ntts_codec* c, int32_t n, const int32_t* codes, const int32_t* codes_dev, int32_t codes_stride,
                             const int32_t* lens, float* wav_out, int64_t wav_stride, int out_kind, hipStream_t producer) {
    if (!c || n < 1 || (!codes && !codes_dev) || !lens || !wav_out) return cfail(c, NTTS_EINVAL, "null/empty argument");
    if (!c->finalized) return cfail(c, NTTS_ESTATE, "weights not finalised");
    CHIP(c, hipSetDevice(c->device));
    const int H = c->H, hop = c->cfg.hop_length;
    int Tmax = 0;
    long total = 0;
    long ncodes = 1;
    for (int i = 0; i < c->nq; ++i) ncodes *= c->cfg.levels[i];
    for (int i = 0; i < n; ++i) {
        if (lens[i] < 1 || lens[i] > c->cfg.max_frames) return cfail(c, NTTS_EINVAL, "utterance %d: %d frames (1..%d)", i, lens[i], c->cfg.max_frames);
        if (lens[i] > Tmax) Tmax = lens[i];
        total += lens[i];
    }
    if (codes)
        for (long i = 0; i < total; ++i)
            if (codes[i] < 0 || codes[i] >= ncodes) return cfail(c, NTTS_EINVAL, "code %d out of range [0, %ld)", codes[i], ncodes);
    if (codes_dev && codes_stride < Tmax) return cfail(c, NTTS_EINVAL, "codes_stride %d < longest utterance %d", codes_stride, Tmax);
    if (wav_stride < (int64_t)hop * Tmax) return cfail(c, NTTS_EINVAL, "wav_stride %ld < %d samples", (long)wav_stride, hop * Tmax);
    const int Tp = Tmax + 2 * kPadRows;
    const long rows = total + 2L * kPadRows * n;          // packed: every utterance brings its own frames + pad rows (codec.h)
    if (rows > c->max_rows) return cfail(c, NTTS_EINVAL, "%ld rows exceed max_rows %ld: decode fewer utterances per call", rows, c->max_rows);
    const int npages = (Tmax + kPage - 1) / kPage, qtiles = (Tmax + 63) / 64;
    // the waveform staging buffer holds n rows of the LONGEST utterance's samples (workspace rows x hop samples in all)
    if ((long)n * Tmax > c->max_rows) return cfail(c, NTTS_EINVAL, "%d utterances x %d frames exceed max_rows %ld: decode fewer utterances per call", n, Tmax, c->max_rows);
    // the V^T pages of the paged attention path are laid out per utterance x the LONGEST utterance's pages (not packed)
    if (!(c->attn_resident && npages <= kAttnResPages) && (long)n * npages * kPage > c->max_rows + (c->max_rows / (1 + 2 * kPadRows) + 1) * kPage)
        return cfail(c, NTTS_EINVAL, "%d utterances x %d pages exceed the V^T workspace: decode fewer utterances per call", n, npages);
    // ---- meta: [lens n][code_off n][row_off n + 1][codes total], built in a page-locked ring slot: the copy is asynchronous, no stream-wide wait
    const size_t m_size = 3 * (size_t)n + 1 + (codes ? (size_t)total : 0);
    if (m_size > c->meta_cap) return cfail(c, NTTS_EINVAL, "meta block too large");
    hipStream_t st = c->stream;
    {
        const int k = c->meta_next;
        c->meta_next = (k + 1) % ntts_codec::kMetaStages;
        if (c->meta_used[k]) CHIP(c, hipEventSynchronize(c->meta_ev[k]));     // (the copy out of this slot two passes ago)
        int* m = c->meta_host[k];
        size_t at = 0;
        memcpy(m, lens, (size_t)n * sizeof(int)); at = n;
        long off = 0;
        for (int i = 0; i < n; ++i) { m[at++] = codes ? (int)off : i * codes_stride; off += lens[i]; }
        long roff = 0;
        for (int i = 0; i < n; ++i) { m[at++] = (int)roff; roff += lens[i] + 2 * kPadRows; }
        m[at++] = (int)roff;
        if (codes) { memcpy(m + at, codes, (size_t)total * sizeof(int)); at += total; }
        CHIP(c, hipMemcpyAsync(c->meta, m, at * sizeof(int), hipMemcpyHostToDevice, st));
        c->meta_used[k] = true;
        CHIP(c, hipEventRecord(c->meta_ev[k], st));
    }
    if (producer) {                        // the codes are being written on another stream: order this pass behind it
        CHIP(c, hipEventRecord(c->ev_in, producer));
        CHIP(c, hipStreamWaitEvent(st, c->ev_in, 0));
    }
    CodecRows R{c->meta, c->meta + 2 * n, n, Tp, rows};
    CHIP(c, hipEventRecord(c->ev[0], st));

    CodecEmbedArgs ea{};
    const long S = c->S;
    ea.codes = codes ? c->meta + 3 * n + 1 : codes_dev; ea.code_off = c->meta + n; ea.wf = c->wf; ea.bf = c->bf; ea.out = c->xa; ea.R = R; ea.H = H; ea.nq = c->nq; ea.split = c->fmt;
    for (int i = 0; i < 8; ++i) ea.levels[i] = i < c->nq ? c->cfg.levels[i] : 1;
    NTTS_LAUNCH((codec_embed_kernel), dim3((unsigned)((rows + kEmbedRows - 1) / kEmbedRows)), dim3(256), st, ea);
    // stem Conv1d(k=7, padding 3): window rows r..r+6 -> centre row r+3
    { GemmArgs ga_ = cg(c->xa, S * H, c->embed_w, 7L * S * H, c->embed_b, c->h + 3L * H, H, rows - 6, H); CGEMM(EPI_F32, ga_, st); }
    auto tap = [&](int stage) -> hipError_t {   // debug only: the residual stream after a stage (hf:models/xcodec2/modeling_xcodec2.py:841-859)
        if (!c->tap) return hipSuccess;
        return hipMemcpyAsync(c->tap + (size_t)stage * c->max_rows * H, c->h, (size_t)rows * H * sizeof(float), hipMemcpyDeviceToDevice, st);
    };
    if (c->tap) {
        c->tap_off.assign(n, 0); c->tap_len.assign(lens, lens + n);
        long ro = 0;
        for (int i = 0; i < n; ++i) { c->tap_off[i] = (int)ro; ro += lens[i] + 2 * kPadRows; }
    }
    CHIP(c, tap(0));
    resnet_block(c, c->res[0], R, rows);
    resnet_block(c, c->res[1], R, rows);
    CHIP(c, tap(1));
    for (int i = 0; i < c->cfg.num_layers; ++i) {
        const CLayerW& L = c->layers[i];
        RowNormArgs rn{};
        rn.x = c->h; rn.y = c->xa; rn.w = L.ln1; rn.rows = rows; rn.C = H; rn.eps = c->cfg.rms_eps; rn.split = c->fmt;
        rownorm_launch(rn, st);
        { GemmArgs ga_ = cg(c->xa, S * H, L.wqkv, S * H, nullptr, c->qkv, 3L * H, rows, 3 * H); CGEMM(EPI_BF16, ga_, st); }
        AttnFullArgs at{};
        at.qkv = c->qkv; at.vt = c->vt; at.out = c->xb; at.R = R; at.C = H; at.nh = c->cfg.num_heads; at.npages = npages; at.qtiles = qtiles; at.split = c->fmt;
        if (c->attn_resident && npages <= kAttnResPages) {   // up to 256 frames: K / V^T resident in LDS, one sweep, no V^T pass
            const dim3 ag(n, c->cfg.num_heads);
            if (c->fmt == kOpF16) {
                if (npages <= 8) NTTS_LAUNCH((attn_full_resident_kernel<8, true>), ag, dim3(512), st, at);
                else if (npages <= 12) NTTS_LAUNCH((attn_full_resident_kernel<12, true>), ag, dim3(512), st, at);
                else NTTS_LAUNCH((attn_full_resident_kernel<16, true>), ag, dim3(512), st, at);
            } else {
                if (npages <= 8) NTTS_LAUNCH((attn_full_resident_kernel<8, false>), ag, dim3(512), st, at);
                else if (npages <= 12) NTTS_LAUNCH((attn_full_resident_kernel<12, false>), ag, dim3(512), st, at);
                else NTTS_LAUNCH((attn_full_resident_kernel<16, false>), ag, dim3(512), st, at);
            }
        } else {
            VTransposeArgs vt{};
            vt.qkv = c->qkv; vt.vt = c->vt; vt.R = R; vt.C = H; vt.nh = c->cfg.num_heads; vt.npages = npages;
            NTTS_LAUNCH((v_transpose_kernel), dim3(n * npages, c->cfg.num_heads), dim3(256), st, vt);
            if (c->fmt == kOpF16) NTTS_LAUNCH((attn_full_kernel<true>), dim3(n * qtiles, c->cfg.num_heads), dim3(256), st, at);
            else NTTS_LAUNCH((attn_full_kernel<false>), dim3(n * qtiles, c->cfg.num_heads), dim3(256), st, at);
        }
        { GemmArgs ga_ = cg(c->xb, S * H, L.wo, S * H, nullptr, c->h, H, rows, H, c->h, H); CGEMM(EPI_F32, ga_, st); }
        rn.w = L.ln2;
        rownorm_launch(rn, st);
        { GemmArgs ga_ = cg(c->xa, S * H, L.fc1, S * H, nullptr, c->act, S * c->I, rows, c->I);
          if (S > 1) NTTS_GEMM_BIG(EPI_SILU_SPLIT3, ga_, st); else CGEMM(EPI_BF16_SILU, ga_, st); }
        { GemmArgs ga_ = cg(c->act, S * c->I, L.fc2, S * c->I, nullptr, c->h, H, rows, H, c->h, H); CGEMM(EPI_F32, ga_, st); }
    }
    CHIP(c, tap(2));
    resnet_block(c, c->res[2], R, rows);
    resnet_block(c, c->res[3], R, rows);
    CHIP(c, tap(3));
    RowNormArgs fn{};
    fn.x = c->h; fn.y = c->xa; fn.w = c->fn_w; fn.bias = c->fn_b; fn.rows = rows; fn.C = H; fn.eps = 1e-6f; fn.split = c->fmt;
    rownorm_launch(fn, st);
    { GemmArgs ga_ = cg(c->xa, S * H, c->head_w, S * H, c->head_b, c->spec, c->lds_spec, rows, c->NS); CGEMM(EPI_F32, ga_, st); }
    IstftPrepArgs ip{};
    ip.spec = c->spec; ip.lds = c->lds_spec; ip.s3 = c->s3; ip.K3 = c->K3; ip.rows = rows; ip.nb = c->nb; ip.f16 = c->fmt == kOpF16;
    NTTS_LAUNCH((istft_prep_kernel), dim3((unsigned)rows), dim3(256), st, ip);
    { GemmArgs ga_ = cg(c->s3, c->K3, c->basis3, c->K3, nullptr, c->frames, c->n_fft, rows, c->n_fft); CGEMM(EPI_F32, ga_, st); }
    OlaArgs oa{};
    oa.frames = c->frames; oa.win2 = c->win2; oa.wav = c->wav; oa.wav_stride = (long)hop * Tmax; oa.R = R; oa.hop = hop; oa.n_fft = c->n_fft;
    oa.scale = c->fmt == kOpF16 ? 1.0f / kDftScale : 1.0f;
    NTTS_LAUNCH((ola_kernel), dim3(n, (unsigned)(((long)hop * Tmax + 255) / 256)), dim3(256), st, oa);
    CHIP(c, hipEventRecord(c->ev[1], st));
    c->have_time = true;
    const hipMemcpyKind kind = out_kind == 2 ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (out_kind == 0) {
        CHIP(c, hipStreamSynchronize(st));
        CHIP(c, hipGetLastError());
        // one strided device-to-host copy for the whole batch (rows shorter than Tmax carry don't-care tails)
        if (wav_stride == (int64_t)hop * Tmax)   // dense destination: one linear copy (DMA engine at PCIe speed into pinned memory)
            CHIP(c, hipMemcpy(wav_out, c->wav, (size_t)n * hop * Tmax * sizeof(float), kind));
        else
            CHIP(c, hipMemcpy2D(wav_out, (size_t)wav_stride * sizeof(float), c->wav, (size_t)hop * Tmax * sizeof(float),
                                (size_t)hop * Tmax * sizeof(float), n, kind));
        return NTTS_OK;
    }
    // asynchronous hand-over, stream-ordered behind the pass (ntts_codec_sync waits for it)
    if (wav_stride == (int64_t)hop * Tmax)
        CHIP(c, hipMemcpyAsync(wav_out, c->wav, (size_t)n * hop * Tmax * sizeof(float), kind, st));
    else
        CHIP(c, hipMemcpy2DAsync(wav_out, (size_t)wav_stride * sizeof(float), c->wav, (size_t)hop * Tmax * sizeof(float),
                                 (size_t)hop * Tmax * sizeof(float), n, kind, st));
    if (c->ev_done && hipEventRecord(c->ev_done, st) == hipSuccess) c->have_done = true;
    else { (void)hipGetLastError(); c->have_done = false; }
    return NTTS_OK;
}

extern "C" int ntts_codec_decode(ntts_codec* c, int32_t n, const int32_t* codes, const int32_t* lens, float* wav_out,
                                 int64_t wav_stride) {
    if (!codes) return cfail(c, NTTS_EINVAL, "null/empty argument");
    return codec_decode_impl(c, n, codes, nullptr, 0, lens, wav_out, wav_stride, 0, nullptr);
}

extern "C" int ntts_codec_decode_dev(ntts_codec* c, int32_t n, const int32_t* codes_dev, int32_t codes_stride, const int32_t* lens,
                                     float* wav_out, int64_t wav_stride, int32_t wav_on_device, void* producer_stream) {
    if (!codes_dev) return cfail(c, NTTS_EINVAL, "null/empty argument");
    return codec_decode_impl(c, n, nullptr, codes_dev, codes_stride, lens, wav_out, wav_stride, wav_on_device ? 2 : 1, (hipStream_t)producer_stream);
}

extern "C" int ntts_codec_set_cu_mask(ntts_codec* c, const uint32_t* mask, int32_t n_words) {
    if (!c || n_words < 0 || (n_words > 0 && !mask)) return cfail(c, NTTS_EINVAL, "bad CU mask");
    CHIP(c, hipSetDevice(c->device));
    CHIP(c, hipStreamSynchronize(c->stream));
    hipStream_t ns = nullptr;
    if (n_words == 0) CHIP(c, hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
    else CHIP(c, hipExtStreamCreateWithCUMask(&ns, (uint32_t)n_words, mask));
    const bool lent = c->stream != c->own_stream;       // (a stream lent by the caller stays in use; the mask applies to the engine's own)
    hipStreamDestroy(c->own_stream);
    c->own_stream = ns;
    if (!lent) c->stream = ns;
    return NTTS_OK;
}

// The codec passes on a stream of the caller's; nullptr returns to the engine's own.  Blocking: drains the stream in use.
extern "C" int ntts_codec_set_stream(ntts_codec* c, void* stream) {
    if (!c) return NTTS_EINVAL;
    CHIP(c, hipSetDevice(c->device));
    CHIP(c, hipStreamSynchronize(c->stream));
    c->stream = stream ? (hipStream_t)stream : c->own_stream;
    return NTTS_OK;
}

extern "C" int ntts_codec_sync(ntts_codec* c) {
    if (!c) return NTTS_EINVAL;
    CHIP(c, hipSetDevice(c->device));
    // On a stream the caller lent, wait for the codec's own most recent pass only: the caller may have enqueued a whole decode phase
    // behind it on the same stream (bench.py's lanes), which is none of this call's business.
    if (c->stream != c->own_stream && c->have_done) CHIP(c, hipEventSynchronize(c->ev_done));
    else CHIP(c, hipStreamSynchronize(c->stream));
    CHIP(c, hipGetLastError());
    return NTTS_OK;
}

// Test tap: keep the fp32 residual stream after the four stages of hf:models/xcodec2/modeling_xcodec2.py:838-862 -- 0 embed (fc + k = 7 conv),
// 1 prior_net (2 ResNet blocks), 2 the transformer layers, 3 post_net (2 ResNet blocks) -- of the most recent decode call.
extern "C" int ntts_codec_set_debug(ntts_codec* c, int32_t keep_stages) {
    if (!c) return NTTS_EINVAL;
    CHIP(c, hipSetDevice(c->device));
    CHIP(c, hipStreamSynchronize(c->stream));
    if (keep_stages && !c->tap) {
        CHIP(c, hipMalloc((void**)&c->tap, 4 * (size_t)c->max_rows * c->H * sizeof(float)));
        CHIP(c, hipMemset(c->tap, 0, 4 * (size_t)c->max_rows * c->H * sizeof(float)));
    } else if (!keep_stages && c->tap) {
        CHIP(c, hipFree(c->tap));
        c->tap = nullptr;
    }
    c->tap_off.clear(); c->tap_len.clear();
    return NTTS_OK;
}
extern "C" int ntts_codec_read_stage(ntts_codec* c, int32_t stage, int32_t utt, float* out, int64_t cap, int32_t* rows, int32_t* cols) {
    if (!c || !out || !rows || !cols || stage < 0 || stage > 3) return cfail(c, NTTS_EINVAL, "bad argument");
    if (!c->tap || utt < 0 || utt >= (int)c->tap_len.size()) return cfail(c, NTTS_ESTATE, "no stage outputs kept for utterance %d: ntts_codec_set_debug(c, 1), then decode", utt);
    const long need = (long)c->tap_len[utt] * c->H;
    if (cap < need) return cfail(c, NTTS_EINVAL, "stage output needs %ld floats", need);
    CHIP(c, hipSetDevice(c->device));
    CHIP(c, hipStreamSynchronize(c->stream));
    CHIP(c, hipMemcpy(out, c->tap + ((size_t)stage * c->max_rows + c->tap_off[utt] + kPadRows) * c->H, (size_t)need * sizeof(float), hipMemcpyDeviceToHost));
    *rows = c->tap_len[utt];
    *cols = c->H;
    return NTTS_OK;
}

extern "C" int ntts_codec_stream(ntts_codec* c, void** stream) {
    if (!c || !stream) return NTTS_EINVAL;
    *stream = (void*)c->stream;
    return NTTS_OK;
}
extern "C" int ntts_codec_limits(ntts_codec* c, int32_t* max_frames, int64_t* max_rows) {
    if (!c || !max_frames || !max_rows) return NTTS_EINVAL;
    *max_frames = c->cfg.max_frames;
    *max_rows = c->max_rows;
    return NTTS_OK;
}

extern "C" int ntts_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return NTTS_EINVAL;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? NTTS_OK : NTTS_ENOMEM;
}
extern "C" int ntts_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? NTTS_OK : NTTS_EHIP; }

// A plain non-blocking HIP stream on `device` for callers without a runtime of their own (neutts._hip.EngineGang: the lanes
// several engines' decode chains and the shared matrix-pass stream run on).  Streams created back to back land in different
// hardware queues (the runtime assigns the least-loaded of its four).
extern "C" int ntts_stream_create(int32_t device, void** stream) {
    if (!stream) return NTTS_EINVAL;
    hipStream_t st = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        return NTTS_EHIP;
    }
    *stream = (void*)st;
    return NTTS_OK;
}
extern "C" int ntts_stream_destroy(int32_t device, void* stream) {
    if (!stream) return NTTS_OK;
    if (hipSetDevice(device) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipStreamDestroy((hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        return NTTS_EHIP;
    }
    return NTTS_OK;
}

extern "C" int ntts_codec_last_timing(ntts_codec* c, float* ms) {
    if (!c || !ms) return NTTS_EINVAL;
    *ms = 0;
    if (c->have_time) CHIP(c, hipEventElapsedTime(ms, c->ev[0], c->ev[1]));
    return NTTS_OK;
}
