// encoder.cpp -- NeuCodec encoder engine behind include/neutts_hip.h (ntts_encoder_*): reference enrolment.
//
// Replaces  self.codec.encode_code(audio_or_path=wav[1,1,L]) -> codes[1,1,T]  (ref:neutts/neutts.py:266-271).  One clip per
// call, fp32 end to end, every Linear / Conv1d through kernels/enc.h's implicit-GEMM kernel on the fp32 matrix core.
// Weights keep the parameter names of transformers' Xcodec2Model (semantic_encoder.*, semantic_adapter.*, acoustic_encoder.*,
// fc_encoder.*, quantizer.project_in.*); activations are channels-last [T][C].
#include <ntts/dev.h>

#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/neutts_hip.h"
#include "kernels/enc.h"

using namespace ntts;

static std::string g_enc_create_err;

struct LinW { float* w = nullptr; float* b = nullptr; int n = 0, k = 0; };
struct SnakeW { float* ea = nullptr; float* inv_b = nullptr; };
struct ConfLayerW {
    float *ln_ffn1_w, *ln_ffn1_b, *ln_attn_w, *ln_attn_b, *ln_conv_w, *ln_conv_b, *ln_dw_w, *ln_dw_b, *ln_ffn2_w, *ln_ffn2_b, *ln_out_w, *ln_out_b;
    LinW ffn1_in, ffn1_out, qkv, attn_out, pw1, pw2, ffn2_in, ffn2_out;
    float* dist_emb;
    float* dw;
};
struct ResUnitW { SnakeW s1, s2; LinW c1, c2; };
struct AcBlockW { ResUnitW ru[3]; SnakeW s; LinW conv; int stride; };

struct ntts_encoder {
    ntts_encoder_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool finalized = false;
    int H = 0, I = 0, hd = 0, cat = 0, hop = 1, nq = 0;
    long max_padded = 0;
    int max_T = 0;
    std::map<std::string, std::vector<float>> host;
    std::map<std::string, std::vector<int64_t>> shapes;
    std::vector<void*> allocs;
    // constants of the fbank front-end and the anti-alias filter
    float *window = nullptr, *tw = nullptr, *melf = nullptr;
    float aa_filter[12]{};
    // weights
    float *fp_ln_w = nullptr, *fp_ln_b = nullptr;
    LinW fp_proj;
    std::vector<ConfLayerW> layers;
    LinW ad[4];
    LinW ac_conv1, ac_conv2, fc, proj_in;
    std::vector<AcBlockW> blocks;
    SnakeW ac_snake;
    // workspaces
    float *wav = nullptr, *logmel = nullptr, *feats = nullptr;
    float *x = nullptr, *xn = nullptr, *big = nullptr, *attn = nullptr, *t1 = nullptr, *t2 = nullptr;   // semantic [T][*]
    float *a0 = nullptr, *a1 = nullptr, *a2 = nullptr;                                             // acoustic ping-pong
    float *catb = nullptr, *fcb = nullptr, *z = nullptr, *lat = nullptr;
    int* codes = nullptr;
    int last_T = 0;
    hipEvent_t ev[2]{};
    bool have_time = false;
};

static int efail(ntts_encoder* e, int code, const char* fmt, ...) {
    char buf[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf; else g_enc_create_err = buf;
    return code;
}
#define EHIP(e, call)                                                                                  \
    do {                                                                                               \
        hipError_t _s = (call);                                                                        \
        if (_s != hipSuccess) return efail(e, _s == 2 ? NTTS_ENOMEM : NTTS_EHIP, "%s failed: %s", #call, hipGetErrorString(_s)); \
    } while (0)

extern "C" const char* ntts_encoder_last_error(const ntts_encoder* e) { return e ? e->err.c_str() : g_enc_create_err.c_str(); }

template <typename T>
static int ealloc(ntts_encoder* e, T** p, size_t n) {
    void* v = nullptr;
    EHIP(e, hipMalloc(&v, (n ? n : 1) * sizeof(T)));
    e->allocs.push_back(v);
    *p = (T*)v;
    return NTTS_OK;
}

extern "C" void ntts_encoder_destroy(ntts_encoder* e) {
    if (!e) return;
    hipSetDevice(e->device);
    hipDeviceSynchronize();
    for (void* p : e->allocs) hipFree(p);
    for (auto& ev : e->ev)
        if (ev) hipEventDestroy(ev);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

// kaiser_sinc_filter1d(cutoff 0.25, half width 0.3, 12 taps)  hf:models/xcodec2/modeling_xcodec2.py:416-460
static double bessel_i0(double x) {
    double s = 1.0, t = 1.0;
    for (int k = 1; k < 64; ++k) { t *= (x / (2.0 * k)) * (x / (2.0 * k)); s += t; if (t < 1e-18 * s) break; }
    return s;
}
static void kaiser_sinc12(float* f) {
    const int ks = 12, half = 6;
    const double cutoff = 0.25, hw = 0.3;
    const double att = 2.285 * (half - 1) * M_PI * 4.0 * hw + 7.95;
    const double beta = att > 50.0 ? 0.1102 * (att - 8.7) : (att >= 21.0 ? 0.5842 * pow(att - 21.0, 0.4) + 0.07886 * (att - 21.0) : 0.0);
    double v[12], sum = 0.0;
    for (int n = 0; n < ks; ++n) {
        const double r = 2.0 * n / (ks - 1) - 1.0;                             // torch.kaiser_window(periodic=False)
        const double win = bessel_i0(beta * sqrt(fmax(0.0, 1.0 - r * r))) / bessel_i0(beta);
        const double t = (n - half) + 0.5;
        const double a = 2.0 * cutoff * t;
        const double sinc = a == 0.0 ? 1.0 : sin(M_PI * a) / (M_PI * a);
        v[n] = 2.0 * cutoff * (double)(float)win * sinc;
        sum += v[n];
    }
    for (int n = 0; n < ks; ++n) f[n] = (float)(v[n] / sum);
}

extern "C" int ntts_encoder_create(const ntts_encoder_config* cf, int device, ntts_encoder** out) {
    if (!cf || !out) return efail(nullptr, NTTS_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device || device < 0)
        return efail(nullptr, NTTS_ENODEV, "no HIP device %d (found %d): this library has no CPU fallback", device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return efail(nullptr, NTTS_ENODEV, "device %d is '%s', kernels are built for gfx950 only", device, prop.gcnArchName);
    if (cf->sem_heads < 1 || cf->sem_hidden % cf->sem_heads || cf->sem_hidden / cf->sem_heads > 64 || (cf->sem_hidden & 3) || (cf->sem_ffn & 3))
        return efail(nullptr, NTTS_EINVAL, "semantic encoder: need hidden %% heads == 0, head size <= 64, sizes multiple of 4");
    if (cf->sem_left < 0 || cf->sem_right < 0 || cf->sem_left + cf->sem_right + 1 > kRelMaxPos)
        return efail(nullptr, NTTS_EINVAL, "relative_key span %d exceeds %d", cf->sem_left + cf->sem_right + 1, kRelMaxPos);
    if (cf->sem_layers < 1 || cf->sem_conv_kernel < 1 || cf->sem_conv_kernel > 128) return efail(nullptr, NTTS_EINVAL, "bad conformer layers / kernel");
    if (cf->n_ratios < 1 || cf->n_ratios > 8 || cf->n_levels < 1 || cf->n_levels > 8 || cf->ac_hidden < 1 || (cf->codec_hidden & 3))
        return efail(nullptr, NTTS_EINVAL, "bad acoustic encoder / FSQ geometry");
    long hop = 1;
    for (int i = 0; i < cf->n_ratios; ++i) {
        if (cf->ratios[i] < 1 || cf->ratios[i] > 16) return efail(nullptr, NTTS_EINVAL, "bad down-sampling ratio");
        hop *= cf->ratios[i];
    }
    if (hop != 2 * kFbShift) return efail(nullptr, NTTS_EINVAL, "hop %ld: the semantic branch runs at 2 x 10 ms frames, the ratios must multiply to %d", hop, 2 * kFbShift);
    if (cf->max_samples < hop) return efail(nullptr, NTTS_EINVAL, "max_samples below one hop");
    ntts_encoder* e = new ntts_encoder();
    e->cfg = *cf;
    e->device = device;
    e->H = cf->sem_hidden; e->I = cf->sem_ffn; e->hd = cf->sem_hidden / cf->sem_heads;
    e->cat = cf->sem_hidden + cf->codec_hidden; e->hop = (int)hop; e->nq = cf->n_levels;
    e->max_padded = ((long)cf->max_samples + 1 + hop - 1) / hop * hop;
    e->max_T = (int)(e->max_padded / hop);
    if (e->max_T > kRelMaxT) { delete e; return efail(nullptr, NTTS_EINVAL, "max_samples: %d frames exceed the attention kernel's %d", e->max_T, kRelMaxT); }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
        delete e;
        return efail(nullptr, NTTS_EHIP, "stream creation failed");
    }
    hipEventCreate(&e->ev[0]); hipEventCreate(&e->ev[1]);
    const size_t T = e->max_T, H = e->H, L = e->max_padded;
    size_t big = std::max<size_t>(std::max<size_t>(e->I, 3 * H), std::max<size_t>(2 * H, 2 * kFbMel));
    size_t ac = 0;   // largest acoustic activation: [L / prod(ratios so far)][width]
    {
        size_t len = L, width = cf->ac_hidden;
        ac = len * width;
        for (int i = 0; i < cf->n_ratios; ++i) { len /= cf->ratios[i]; width *= 2; ac = std::max(ac, len * width); }
    }
    int rc = NTTS_OK;
#define A(call) if (rc == NTTS_OK) rc = (call)
    A(ealloc(e, &e->wav, L)); A(ealloc(e, &e->logmel, 2 * T * kFbMel)); A(ealloc(e, &e->feats, T * 2 * kFbMel));
    A(ealloc(e, &e->x, T * H)); A(ealloc(e, &e->xn, T * H)); A(ealloc(e, &e->big, T * big)); A(ealloc(e, &e->attn, T * H));
    A(ealloc(e, &e->t1, T * H)); A(ealloc(e, &e->t2, T * H));
    A(ealloc(e, &e->a0, ac)); A(ealloc(e, &e->a1, ac)); A(ealloc(e, &e->a2, ac));
    A(ealloc(e, &e->catb, T * e->cat)); A(ealloc(e, &e->fcb, T * e->cat)); A(ealloc(e, &e->z, T * 8)); A(ealloc(e, &e->lat, T * 8));
    A(ealloc(e, &e->codes, T));
    A(ealloc(e, &e->window, kFbFrame)); A(ealloc(e, &e->tw, kFbFft * 2)); A(ealloc(e, &e->melf, kFbBins * kFbMel));
    if (rc != NTTS_OK) {
        g_enc_create_err = e->err;
        ntts_encoder_destroy(e);
        return rc;
    }
    {   // povey window, DFT twiddles, kaldi mel triangles (drawn in mel space)  -- hf:audio_utils.py window_function / mel_filter_bank
        std::vector<float> win(kFbFrame), tw(kFbFft * 2), mf((size_t)kFbBins * kFbMel);
        for (int n = 0; n < kFbFrame; ++n) win[n] = (float)pow(0.5 - 0.5 * cos(2.0 * M_PI * n / (kFbFrame - 1)), 0.85);
        for (int i = 0; i < kFbFft; ++i) { tw[2 * i] = (float)cos(2.0 * M_PI * i / kFbFft); tw[2 * i + 1] = (float)sin(2.0 * M_PI * i / kFbFft); }
        auto mel = [](double f) { return 1127.0 * log(1.0 + f / 700.0); };
        const double lo = mel(20.0), hi = mel(8000.0);
        std::vector<double> edges(kFbMel + 2);
        for (int i = 0; i < kFbMel + 2; ++i) edges[i] = lo + (hi - lo) * i / (kFbMel + 1);
        for (int k = 0; k < kFbBins; ++k) {
            const double fm = mel(16000.0 / ((kFbBins - 1) * 2) * k);
            for (int m = 0; m < kFbMel; ++m) {
                const double down = (fm - edges[m]) / (edges[m + 1] - edges[m]);          // rising edge
                const double up = (edges[m + 2] - fm) / (edges[m + 2] - edges[m + 1]);    // falling edge
                mf[(size_t)k * kFbMel + m] = (float)fmax(0.0, fmin(down, up));
            }
        }
        hipMemcpy(e->window, win.data(), win.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(e->tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(e->melf, mf.data(), mf.size() * 4, hipMemcpyHostToDevice);
        kaiser_sinc12(e->aa_filter);
    }
    *out = e;
    return NTTS_OK;
}

extern "C" int ntts_encoder_load_tensor(ntts_encoder* e, const char* name, const void* data, int dtype, const int64_t* shape,
                                        int ndim, int is_device) {
    if (!e || !name || !data || !shape || ndim < 1 || ndim > 3) return efail(e, NTTS_EINVAL, "bad argument");
    if (e->finalized) return efail(e, NTTS_ESTATE, "weights already finalised");
    if (dtype != NTTS_DT_F32 && dtype != NTTS_DT_BF16) return efail(e, NTTS_EINVAL, "tensor '%s': dtype must be f32 or bf16", name);
    EHIP(e, hipSetDevice(e->device));
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    const size_t esz = dtype == NTTS_DT_F32 ? 4 : 2;
    std::vector<unsigned char> raw(n * esz);
    if (is_device) EHIP(e, hipMemcpy(raw.data(), data, n * esz, hipMemcpyDeviceToHost));
    else memcpy(raw.data(), data, n * esz);
    std::vector<float> v(n);
    if (dtype == NTTS_DT_F32) memcpy(v.data(), raw.data(), n * 4);
    else
        for (size_t i = 0; i < n; ++i) {
            const uint32_t u = (uint32_t)((const uint16_t*)raw.data())[i] << 16;
            memcpy(&v[i], &u, 4);
        }
    e->host[name] = std::move(v);
    e->shapes[name] = std::vector<int64_t>(shape, shape + ndim);
    return NTTS_OK;
}

namespace {
struct EncFinalizer {
    ntts_encoder* e;
    int rc = NTTS_OK;
    std::string missing, bad;
    const std::vector<float>* get(const std::string& name, std::initializer_list<int64_t> shp) {
        auto it = e->host.find(name);
        if (it == e->host.end()) { missing += (missing.empty() ? "" : ", ") + name; return nullptr; }
        const auto& s = e->shapes[name];
        if (s.size() != shp.size() || !std::equal(s.begin(), s.end(), shp.begin())) { bad += (bad.empty() ? "" : ", ") + name; return nullptr; }
        return &it->second;
    }
    float* up(const std::vector<float>& v) {
        float* d = nullptr;
        if (ealloc(e, &d, v.size()) != NTTS_OK || hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = NTTS_EHIP; return nullptr; }
        return d;
    }
    float* vec(const std::string& name, int64_t n) {
        auto* v = get(name, {n});
        return v ? up(*v) : nullptr;
    }
    LinW lin(const std::string& name, int64_t n, int64_t k, bool bias = true) {
        LinW r; r.n = (int)n; r.k = (int)k;
        auto* v = get(name + ".weight", {n, k});
        if (v) r.w = up(*v);
        if (bias) r.b = vec(name + ".bias", n);
        return r;
    }
    // Conv1d weight [Cout][Cin][k] -> implicit-GEMM weight [Cout][k][Cin]
    LinW conv(const std::string& name, int64_t co, int64_t ci, int64_t k, bool bias = true) {
        LinW r; r.n = (int)co; r.k = (int)(ci * k);
        auto* v = get(name + ".weight", {co, ci, k});
        if (v) {
            std::vector<float> t((size_t)co * k * ci);
            for (int64_t o = 0; o < co; ++o)
                for (int64_t i = 0; i < ci; ++i)
                    for (int64_t j = 0; j < k; ++j) t[((size_t)o * k + j) * ci + i] = (*v)[((size_t)o * ci + i) * k + j];
            r.w = up(t);
        }
        if (bias) r.b = vec(name + ".bias", co);
        return r;
    }
    // SnakeBeta parameters -> e^alpha and 1 / (e^beta + 1e-9)  hf:models/xcodec2/modeling_xcodec2.py:405-411
    SnakeW snake(const std::string& name, int64_t c) {
        SnakeW r;
        auto* a = get(name + ".act.alpha", {c});
        auto* b = get(name + ".act.beta", {c});
        if (a && b) {
            std::vector<float> ea(c), ib(c);
            for (int64_t i = 0; i < c; ++i) { ea[i] = expf((*a)[i]); ib[i] = 1.0f / (expf((*b)[i]) + 0.000000001f); }
            r.ea = up(ea); r.inv_b = up(ib);
        }
        return r;
    }
};
}  // namespace

extern "C" int ntts_encoder_finalize(ntts_encoder* e) {
    if (!e) return NTTS_EINVAL;
    if (e->finalized) return NTTS_OK;
    EHIP(e, hipSetDevice(e->device));
    const auto& c = e->cfg;
    const int64_t H = e->H, I = e->I;
    EncFinalizer f{e};
    const std::string sp = "semantic_encoder.";
    e->fp_ln_w = f.vec(sp + "feature_projection.layer_norm.weight", 2 * kFbMel);
    e->fp_ln_b = f.vec(sp + "feature_projection.layer_norm.bias", 2 * kFbMel);
    e->fp_proj = f.lin(sp + "feature_projection.projection", H, 2 * kFbMel);
    e->layers.resize(c.sem_layers);
    for (int i = 0; i < c.sem_layers; ++i) {
        const std::string p = sp + "encoder.layers." + std::to_string(i) + ".";
        ConfLayerW& L = e->layers[i];
        L.ln_ffn1_w = f.vec(p + "ffn1_layer_norm.weight", H); L.ln_ffn1_b = f.vec(p + "ffn1_layer_norm.bias", H);
        L.ffn1_in = f.lin(p + "ffn1.intermediate_dense", I, H); L.ffn1_out = f.lin(p + "ffn1.output_dense", H, I);
        L.ln_attn_w = f.vec(p + "self_attn_layer_norm.weight", H); L.ln_attn_b = f.vec(p + "self_attn_layer_norm.bias", H);
        {   // q, k, v projections as one [3H][H] GEMM
            auto* wq = f.get(p + "self_attn.linear_q.weight", {H, H}); auto* bq = f.get(p + "self_attn.linear_q.bias", {H});
            auto* wk = f.get(p + "self_attn.linear_k.weight", {H, H}); auto* bk = f.get(p + "self_attn.linear_k.bias", {H});
            auto* wv = f.get(p + "self_attn.linear_v.weight", {H, H}); auto* bv = f.get(p + "self_attn.linear_v.bias", {H});
            L.qkv.n = (int)(3 * H); L.qkv.k = (int)H;
            if (wq && wk && wv && bq && bk && bv) {
                std::vector<float> w, b;
                for (auto* t : {wq, wk, wv}) w.insert(w.end(), t->begin(), t->end());
                for (auto* t : {bq, bk, bv}) b.insert(b.end(), t->begin(), t->end());
                L.qkv.w = f.up(w); L.qkv.b = f.up(b);
            }
        }
        L.attn_out = f.lin(p + "self_attn.linear_out", H, H);
        {
            auto* d = f.get(p + "self_attn.distance_embedding.weight", {c.sem_left + c.sem_right + 1, e->hd});
            L.dist_emb = d ? f.up(*d) : nullptr;
        }
        L.ln_conv_w = f.vec(p + "conv_module.layer_norm.weight", H); L.ln_conv_b = f.vec(p + "conv_module.layer_norm.bias", H);
        L.pw1 = f.conv(p + "conv_module.pointwise_conv1", 2 * H, H, 1, false);
        {
            auto* d = f.get(p + "conv_module.depthwise_conv.weight", {H, 1, c.sem_conv_kernel});
            L.dw = d ? f.up(*d) : nullptr;
        }
        L.ln_dw_w = f.vec(p + "conv_module.depthwise_layer_norm.weight", H); L.ln_dw_b = f.vec(p + "conv_module.depthwise_layer_norm.bias", H);
        L.pw2 = f.conv(p + "conv_module.pointwise_conv2", H, H, 1, false);
        L.ln_ffn2_w = f.vec(p + "ffn2_layer_norm.weight", H); L.ln_ffn2_b = f.vec(p + "ffn2_layer_norm.bias", H);
        L.ffn2_in = f.lin(p + "ffn2.intermediate_dense", I, H); L.ffn2_out = f.lin(p + "ffn2.output_dense", H, I);
        L.ln_out_w = f.vec(p + "final_layer_norm.weight", H); L.ln_out_b = f.vec(p + "final_layer_norm.bias", H);
    }
    e->ad[0] = f.conv("semantic_adapter.conv1", H, H, 3, false);
    e->ad[1] = f.conv("semantic_adapter.conv2", H, H, 3, true);
    e->ad[2] = f.conv("semantic_adapter.conv3", H, H, 3, true);
    e->ad[3] = f.conv("semantic_adapter.conv4", H, H, 3, false);
    const std::string ap = "acoustic_encoder.";
    e->ac_conv1 = f.conv(ap + "conv1", c.ac_hidden, 1, 7);
    e->blocks.resize(c.n_ratios);
    for (int bi = 0; bi < c.n_ratios; ++bi) {
        const int64_t dim = (int64_t)c.ac_hidden << (bi + 1), half = dim / 2;
        const std::string b = ap + "block." + std::to_string(bi) + ".";
        AcBlockW& B = e->blocks[bi];
        B.stride = c.ratios[bi];
        for (int u = 0; u < 3; ++u) {
            const std::string r = b + "res_unit" + std::to_string(u + 1) + ".";
            B.ru[u].s1 = f.snake(r + "snake1", half);
            B.ru[u].c1 = f.conv(r + "conv1", half, half, 7);
            B.ru[u].s2 = f.snake(r + "snake2", half);
            B.ru[u].c2 = f.conv(r + "conv2", half, half, 1);
        }
        B.s = f.snake(b + "snake1", half);
        B.conv = f.conv(b + "conv1", dim, half, 2 * c.ratios[bi]);
    }
    const int64_t dmodel = (int64_t)c.ac_hidden << c.n_ratios;
    e->ac_snake = f.snake(ap + "snake1", dmodel);
    e->ac_conv2 = f.conv(ap + "conv2", c.codec_hidden, dmodel, 3);
    e->fc = f.lin("fc_encoder", e->cat, e->cat);
    e->proj_in = f.lin("quantizer.project_in", c.n_levels, e->cat);
    if (!f.missing.empty() || !f.bad.empty())
        return efail(e, NTTS_ESTATE, "encoder weights incomplete -- missing: [%s]  unexpected shape: [%s]", f.missing.c_str(), f.bad.c_str());
    if (f.rc != NTTS_OK) return efail(e, f.rc, "uploading encoder weights failed");
    e->host.clear();
    e->shapes.clear();
    e->finalized = true;
    return NTTS_OK;
}

// ------------------------------------------------------------------------------------------------------------ launch helpers
static void gemm(ntts_encoder* e, const float* X, long ldx, int Tin, int Cin, int taps, int dil, int stride, int pad, const LinW& w,
                 float* Y, long ldy, int M, int act = 0, float alpha = 1.f, const float* resid = nullptr, long ldr = 0) {
    SgemmArgs a{};
    a.X = X; a.ldx = ldx; a.Tin = Tin; a.Cin = Cin; a.taps = taps; a.dil = dil; a.stride = stride; a.pad = pad;
    a.W = w.w; a.ldw = w.k; a.bias = w.b; a.resid = resid; a.ldr = ldr; a.alpha = alpha; a.act = act;
    a.Y = Y; a.ldy = ldy; a.M = M; a.N = w.n; a.K = w.k;
    sgemm_launch(a, e->stream);
}
static void linear(ntts_encoder* e, const float* X, long ldx, const LinW& w, float* Y, long ldy, int M, int act = 0, float alpha = 1.f,
                   const float* resid = nullptr, long ldr = 0) {
    gemm(e, X, ldx, M, w.k, 1, 1, 1, 0, w, Y, ldy, M, act, alpha, resid, ldr);
}
static void layernorm(ntts_encoder* e, const float* X, long ldx, float* Y, long ldy, const float* w, const float* b, int M, int C, int act = 0) {
    LayerNormArgs a{X, ldx, Y, ldy, w, b, e->cfg.sem_ln_eps, M, C, act};
    NTTS_LAUNCH((enc_layernorm_kernel), dim3((M + 3) / 4), dim3(256), e->stream, a);
}
static void snake(ntts_encoder* e, const float* X, float* Y, const SnakeW& s, int T, int C) {
    SnakeArgs a{};
    a.X = X; a.ldx = C; a.Y = Y; a.ldy = C; a.ea = s.ea; a.inv_b = s.inv_b; a.T = T; a.C = C;
    memcpy(a.f, e->aa_filter, sizeof a.f);
    const long n = (long)T * C;
    NTTS_LAUNCH((enc_snake_aa_kernel), dim3((unsigned)((n + 255) / 256)), dim3(256), e->stream, a);
}

extern "C" int ntts_encoder_encode(ntts_encoder* e, const float* wav, int64_t n_samples, int32_t* codes_out, int32_t cap, int32_t* n_codes) {
    if (!e || !wav || !codes_out || !n_codes || n_samples < 1) return efail(e, NTTS_EINVAL, "bad argument");
    if (!e->finalized) return efail(e, NTTS_ESTATE, "weights not finalised");
    if (n_samples > e->cfg.max_samples) return efail(e, NTTS_EINVAL, "clip of %lld samples exceeds max_samples %d", (long long)n_samples, e->cfg.max_samples);
    EHIP(e, hipSetDevice(e->device));
    const auto& c = e->cfg;
    hipStream_t st = e->stream;
    // zero-pad: one sample, then up to the next hop multiple (hf:models/xcodec2/feature_extraction_xcodec2.py:149-158)
    const long Lp = ((long)n_samples + 1 + e->hop - 1) / e->hop * e->hop;
    const int T = (int)(Lp / e->hop);
    if (T > cap) return efail(e, NTTS_EINVAL, "codes_out holds %d codes, the clip produces %d", cap, T);
    const int H = e->H, I = e->I;
    EHIP(e, hipMemsetAsync(e->wav, 0, Lp * 4, st));
    EHIP(e, hipMemcpyAsync(e->wav, wav, (size_t)n_samples * 4, hipMemcpyHostToDevice, st));
    EHIP(e, hipEventRecord(e->ev[0], st));
    // ---- semantic branch: fbank -> per-bin normalisation, frame pairs -> LayerNorm -> Linear -> conformer layers -> adapter
    const int nfr = 1 + (int)((Lp + 2 * kFbShift - kFbFrame) / kFbShift);      // == 2 T
    {
        FbankArgs fa{e->wav, Lp, e->window, e->tw, e->melf, e->logmel, nfr};
        NTTS_LAUNCH((enc_fbank_kernel), dim3(nfr), dim3(256), st, fa);
        NTTS_LAUNCH((enc_melnorm_kernel), dim3(kFbMel), dim3(256), st, (const float*)e->logmel, e->feats, nfr, 2 * T);
    }
    layernorm(e, e->feats, 2 * kFbMel, e->big, 2 * kFbMel, e->fp_ln_w, e->fp_ln_b, T, 2 * kFbMel);
    linear(e, e->big, 2 * kFbMel, e->fp_proj, e->x, H, T);
    for (int li = 0; li < c.sem_layers; ++li) {
        const ConfLayerW& L = e->layers[li];
        // 1. half-step feed-forward: x = ffn1(ln(x)) * 0.5 + x
        layernorm(e, e->x, H, e->xn, H, L.ln_ffn1_w, L.ln_ffn1_b, T, H);
        linear(e, e->xn, H, L.ffn1_in, e->big, I, T, 2);
        linear(e, e->big, I, L.ffn1_out, e->x, H, T, 0, 0.5f, e->x, H);
        // 2. self-attention (relative_key)
        layernorm(e, e->x, H, e->xn, H, L.ln_attn_w, L.ln_attn_b, T, H);
        linear(e, e->xn, H, L.qkv, e->big, 3 * H, T);
        {
            RelAttnArgs a{e->big, 3L * H, L.dist_emb, e->attn, H, T, c.sem_heads, e->hd, c.sem_left, c.sem_right, 1.0f / sqrtf((float)e->hd)};
            NTTS_LAUNCH((enc_rel_attn_kernel), dim3((T + kRelQB - 1) / kRelQB, c.sem_heads), dim3(256), st, a);
        }
        linear(e, e->attn, H, L.attn_out, e->x, H, T, 0, 1.f, e->x, H);
        // 3. convolution module: ln -> pointwise (H -> 2H) -> GLU -> causal depthwise -> ln + swish -> pointwise -> + x
        layernorm(e, e->x, H, e->xn, H, L.ln_conv_w, L.ln_conv_b, T, H);
        linear(e, e->xn, H, L.pw1, e->big, 2 * H, T);
        {
            const long n = (long)T * H;
            NTTS_LAUNCH((enc_glu_kernel), dim3((unsigned)((n + 255) / 256)), dim3(256), st, (const float*)e->big, 2L * H, e->t1, (long)H, T, H);
            NTTS_LAUNCH((enc_dwconv_kernel), dim3((unsigned)((n + 255) / 256)), dim3(256), st, (const float*)e->t1, (long)H, (const float*)L.dw, e->t2, (long)H, T, H, c.sem_conv_kernel);
        }
        layernorm(e, e->t2, H, e->t1, H, L.ln_dw_w, L.ln_dw_b, T, H, 2);
        linear(e, e->t1, H, L.pw2, e->x, H, T, 0, 1.f, e->x, H);
        // 4. half-step feed-forward, then the layer's final LayerNorm
        layernorm(e, e->x, H, e->xn, H, L.ln_ffn2_w, L.ln_ffn2_b, T, H);
        linear(e, e->xn, H, L.ffn2_in, e->big, I, T, 2);
        linear(e, e->big, I, L.ffn2_out, e->x, H, T, 0, 0.5f, e->x, H);
        layernorm(e, e->x, H, e->xn, H, L.ln_out_w, L.ln_out_b, T, H);
        std::swap(e->x, e->xn);
    }
    // semantic adapter (four k = 3 convolutions, ReLU, one skip): the result is the left half of the concat buffer
    gemm(e, e->x, H, T, H, 3, 1, 1, 1, e->ad[0], e->t1, H, T, 1);
    gemm(e, e->t1, H, T, H, 3, 1, 1, 1, e->ad[1], e->t2, H, T, 1);
    gemm(e, e->t2, H, T, H, 3, 1, 1, 1, e->ad[2], e->xn, H, T, 0, 1.f, e->t1, H);
    gemm(e, e->xn, H, T, H, 3, 1, 1, 1, e->ad[3], e->catb, e->cat, T);
    // ---- acoustic branch at the sample rate: conv(1 -> C, k 7), then per ratio three dilated residual units + a strided conv
    float *h = e->a0, *s = e->a1, *y = e->a2;
    long len = Lp;
    int ch = c.ac_hidden;
    gemm(e, e->wav, 1, (int)len, 1, 7, 1, 1, 3, e->ac_conv1, h, ch, (int)len);
    for (int bi = 0; bi < c.n_ratios; ++bi) {
        const AcBlockW& B = e->blocks[bi];
        static const int dils[3] = {1, 3, 9};
        for (int u = 0; u < 3; ++u) {
            snake(e, h, s, B.ru[u].s1, (int)len, ch);
            gemm(e, s, ch, (int)len, ch, 7, dils[u], 1, 3 * dils[u], B.ru[u].c1, y, ch, (int)len);
            snake(e, y, s, B.ru[u].s2, (int)len, ch);
            gemm(e, s, ch, (int)len, ch, 1, 1, 1, 0, B.ru[u].c2, h, ch, (int)len, 0, 1.f, h, ch);
        }
        snake(e, h, s, B.s, (int)len, ch);
        const int stride = B.stride;
        const long out_len = len / stride;
        gemm(e, s, ch, (int)len, ch, 2 * stride, 1, stride, (stride + 1) / 2, B.conv, y, 2 * ch, (int)out_len);
        std::swap(h, y);
        len = out_len;
        ch *= 2;
    }
    snake(e, h, s, e->ac_snake, (int)len, ch);
    gemm(e, s, ch, (int)len, ch, 3, 1, 1, 1, e->ac_conv2, e->catb + H, e->cat, T);
    // ---- concat -> Linear -> project to the FSQ dimensions -> bound, round, index
    linear(e, e->catb, e->cat, e->fc, e->fcb, e->cat, T);
    linear(e, e->fcb, e->cat, e->proj_in, e->z, 8, T);
    {
        FsqArgs q{};
        q.z = e->z; q.ldz = 8; q.lat = e->lat; q.codes = e->codes; q.T = T; q.n = c.n_levels;
        for (int d = 0; d < c.n_levels; ++d) {
            q.levels[d] = c.levels[d];
            q.half_range[d] = (float)(c.levels[d] - 1) * (1.0f + 1e-3f) / 2.0f;
            q.offset[d] = (c.levels[d] % 2 == 0) ? 0.5f : 0.0f;
            q.shift[d] = atanhf(q.offset[d] / q.half_range[d]);
        }
        NTTS_LAUNCH((enc_fsq_kernel), dim3((T + 255) / 256), dim3(256), st, q);
    }
    EHIP(e, hipEventRecord(e->ev[1], st));
    EHIP(e, hipMemcpyAsync(codes_out, e->codes, (size_t)T * 4, hipMemcpyDeviceToHost, st));
    EHIP(e, hipStreamSynchronize(st));
    EHIP(e, hipGetLastError());
    e->last_T = T;
    e->have_time = true;
    *n_codes = T;
    return NTTS_OK;
}

extern "C" int ntts_encoder_read_stage(ntts_encoder* e, int32_t stage, float* out, int64_t cap, int32_t* rows, int32_t* cols) {
    if (!e || !out || !rows || !cols) return efail(e, NTTS_EINVAL, "bad argument");
    if (e->last_T < 1) return efail(e, NTTS_ESTATE, "no encode call yet");
    EHIP(e, hipSetDevice(e->device));
    const float* src = nullptr;
    int C = 0;
    switch (stage) {
        case 0: src = e->feats; C = 2 * kFbMel; break;
        case 1: src = e->catb; C = e->cat; break;
        case 2: src = e->fcb; C = e->cat; break;
        case 3: src = e->lat; C = e->cfg.n_levels; break;
        default: return efail(e, NTTS_EINVAL, "unknown stage %d", stage);
    }
    if ((int64_t)e->last_T * C > cap) return efail(e, NTTS_EINVAL, "stage %d needs %lld floats", stage, (long long)e->last_T * C);
    EHIP(e, hipMemcpy(out, src, (size_t)e->last_T * C * 4, hipMemcpyDeviceToHost));
    *rows = e->last_T;
    *cols = C;
    return NTTS_OK;
}

extern "C" int ntts_encoder_last_timing(ntts_encoder* e, float* ms) {
    if (!e || !ms) return NTTS_EINVAL;
    if (!e->have_time) return efail(e, NTTS_ESTATE, "no encode call yet");
    EHIP(e, hipSetDevice(e->device));
    EHIP(e, hipEventElapsedTime(ms, e->ev[0], e->ev[1]));
    return NTTS_OK;
}
