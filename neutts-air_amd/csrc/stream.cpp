// stream.cpp -- device-side streaming behind include/neutts_hip.h (ABI 6): ntts_streams_*.
//
// The reference's infer_stream (ref:neutts/neutts.py:373-465; GGUF backend only there) keeps a token cache per utterance, decodes a
// window of <= 81 codes whenever 30 undecoded tokens have arrived, keeps 27 frames of it and cross-fades them with the previous
// chunk.  Rounds 2-3 did the cache, the window assembly and the cross-fade in Python / numpy: at BASELINE configs[4]'s own size (512
// concurrent streams) the loop was host-paced -- 512 windows built, 512 whole windows shipped D2H, 512 chunks blended per burst.
// Here the codes never leave the device (ntts_backbone_append_codes -> stream_gather_kernel -> ntts_codec_decode_dev) and only each
// stream's NEW samples do (stream_blend_kernel -> one D2H copy into page-locked memory); the host decides WHICH windows exist -- a few
// integers per stream and burst -- and yields.  Built on the two engines' public entry points only.
#include <ntts/dev.h>

#include <stdarg.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/neutts_hip.h"
#include "kernels/stream.h"

using namespace ntts;

struct ntts_streams {
    ntts_backbone* e = nullptr;
    ntts_codec* c = nullptr;
    ntts_stream_params prm{};
    int device = 0, n = 0, hop = 0;
    std::vector<int> slots;
    hipStream_t est = nullptr, cst = nullptr;      // the backbone's / the codec's stream (borrowed)
    int max_frames = 0;
    long max_rows = 0;
    // device
    int* ibuf = nullptr;                            // cache [n][cache_stride] | clen [n] | fin [n] | seen [n]
    int *cache = nullptr, *clen = nullptr, *fin = nullptr, *seen = nullptr;
    int cache_stride = 0;
    int* win = nullptr;                             // [2 n][W]
    int W = 0;
    float *wav_win = nullptr, *prev = nullptr, *out_dev = nullptr;
    long wav_stride = 0, prev_stride = 0, out_stride = 0;
    StreamJob* jobs_dev = nullptr;
    int* parity_dev = nullptr;
    // page-locked host
    int* snap_host = nullptr;                       // clen [n] | fin [n]
    float* out_host = nullptr;
    StreamJob* jobs_host = nullptr;
    int* parity_host = nullptr;
    hipEvent_t ev_snap = nullptr;
    bool snap_open = false, have_snap = false;
    // host view of every stream
    std::vector<int> n_dec, prev_len, frames, done;
    std::string err;
};

static std::string g_streams_err;
static int sfail(ntts_streams* s, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (s) s->err = buf; else g_streams_err = buf;
    return code;
}
#define SHIP(s, call)                                                                                   \
    do {                                                                                                \
        hipError_t _s = (call);                                                                         \
        if (_s != hipSuccess) return sfail(s, NTTS_EHIP, "%s failed: %s", #call, hipGetErrorString(_s)); \
    } while (0)

extern "C" const char* ntts_streams_last_error(const ntts_streams* s) { return s ? s->err.c_str() : g_streams_err.c_str(); }

extern "C" void ntts_streams_destroy(ntts_streams* s) {
    if (!s) return;
    hipSetDevice(s->device);
    if (s->cst) hipStreamSynchronize(s->cst);
    if (s->est) hipStreamSynchronize(s->est);
    for (void* b : {(void*)s->ibuf, (void*)s->win, (void*)s->wav_win, (void*)s->prev, (void*)s->out_dev, (void*)s->jobs_dev, (void*)s->parity_dev})
        if (b) hipFree(b);
    for (void* b : {(void*)s->snap_host, (void*)s->out_host, (void*)s->jobs_host, (void*)s->parity_host})
        if (b) hipHostFree(b);
    if (s->ev_snap) hipEventDestroy(s->ev_snap);
    delete s;
}

extern "C" int ntts_streams_create(ntts_backbone* e, ntts_codec* c, const ntts_stream_params* prm, int32_t device, int32_t n, const int32_t* slots,
                                   const int32_t* ref_codes, const int32_t* ref_lens, int32_t max_new_tokens, ntts_streams** out) {
    if (!e || !c || !prm || !out || n < 1 || !slots || !ref_codes || !ref_lens || max_new_tokens < 1) return sfail(nullptr, NTTS_EINVAL, "null/empty argument");
    if (prm->chunk < 1 || prm->lookforward < 0 || prm->lookback < 0 || prm->overlap < 0 || prm->hop_length < 1 || prm->n_codes < 1)
        return sfail(nullptr, NTTS_EINVAL, "bad stream parameters");
    // one previous frame must be all that overlaps a new one (neutts.py _StreamBlender): frame = chunk + 2 overlap hops at a stride of chunk hops
    if (2 * prm->overlap >= prm->chunk) return sfail(nullptr, NTTS_EINVAL, "overlap %d too large for chunk %d", prm->overlap, prm->chunk);
    // a regular window ends `lookforward` frames past the chunk, and the blend reads (chunk + 2 overlap) hops of it from the chunk's start
    // on: with lookforward < 2 overlap it would read past the decoded window (the reference's slice is simply shorter there --
    // those parameters stay on the host loop, neutts.py _stream_on_device)
    if (prm->lookforward < 2 * prm->overlap)
        return sfail(nullptr, NTTS_EINVAL, "lookforward %d < 2 x overlap %d: not supported on the device-side stream path", prm->lookforward, prm->overlap);
    ntts_streams* s = new ntts_streams();
    s->e = e; s->c = c; s->prm = *prm; s->device = device; s->n = n; s->hop = prm->hop_length;
    s->slots.assign(slots, slots + n);
#define SCR(call)                                                                             \
    do {                                                                                      \
        hipError_t _s = (call);                                                               \
        if (_s != hipSuccess) {                                                               \
            int rc = sfail(nullptr, _s == 2 ? NTTS_ENOMEM : NTTS_EHIP, "%s failed: %s", #call, hipGetErrorString(_s)); \
            ntts_streams_destroy(s);                                                          \
            return rc;                                                                        \
        }                                                                                     \
    } while (0)
    SCR(hipSetDevice(device));
    void* st = nullptr;
    if (ntts_backbone_stream(e, &st) != NTTS_OK) { ntts_streams_destroy(s); return sfail(nullptr, NTTS_EINVAL, "backbone stream"); }
    s->est = (hipStream_t)st;
    if (ntts_codec_stream(c, &st) != NTTS_OK || ntts_codec_limits(c, &s->max_frames, &s->max_rows) != NTTS_OK) {
        ntts_streams_destroy(s);
        return sfail(nullptr, NTTS_EINVAL, "codec stream / limits");
    }
    s->cst = (hipStream_t)st;
    const int need = prm->chunk + prm->lookforward;
    s->W = prm->lookback + prm->overlap + need + prm->overlap;            // longest window: ref :407-415 (start = n_dec - 51, end = n_dec + 31)
    if (s->W > s->max_frames) { ntts_streams_destroy(s); return sfail(nullptr, NTTS_EINVAL, "windows of %d frames exceed the codec engine's max_frames %d", s->W, s->max_frames); }
    int max_ref = 0;
    long total_ref = 0;
    for (int i = 0; i < n; ++i) {
        if (ref_lens[i] < 0) { ntts_streams_destroy(s); return sfail(nullptr, NTTS_EINVAL, "negative reference length"); }
        if (ref_lens[i] > max_ref) max_ref = ref_lens[i];
        total_ref += ref_lens[i];
    }
    s->cache_stride = max_ref + max_new_tokens + 1;
    const size_t n_int = (size_t)n * s->cache_stride + 3 * (size_t)n;
    SCR(hipMalloc((void**)&s->ibuf, n_int * sizeof(int)));
    SCR(hipMemset(s->ibuf, 0, n_int * sizeof(int)));
    s->cache = s->ibuf; s->clen = s->ibuf + (size_t)n * s->cache_stride; s->fin = s->clen + n; s->seen = s->fin + n;
    {   // the reference codes open every stream's cache (ref:neutts/neutts.py:385-387)
        std::vector<int> rows((size_t)n * s->cache_stride, 0), lens(ref_lens, ref_lens + n);
        long off = 0;
        for (int i = 0; i < n; ++i) {
            for (int t = 0; t < ref_lens[i]; ++t) {
                const int cde = ref_codes[off + t];
                if (cde < 0 || cde >= prm->n_codes) { ntts_streams_destroy(s); return sfail(nullptr, NTTS_EINVAL, "reference code %d out of range", cde); }
                rows[(size_t)i * s->cache_stride + t] = cde;
            }
            off += ref_lens[i];
        }
        SCR(hipMemcpy(s->cache, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice));
        SCR(hipMemcpy(s->clen, lens.data(), n * sizeof(int), hipMemcpyHostToDevice));
    }
    const int J = 2 * n;                                                  // per round: one regular window + one final window per stream
    s->wav_stride = (long)s->W * s->hop;
    s->prev_stride = (long)(need + prm->overlap) * s->hop;                // longest frame: a final window's remaining (< need) + overlap frames
    s->out_stride = s->prev_stride;
    SCR(hipMalloc((void**)&s->win, (size_t)J * s->W * sizeof(int)));
    SCR(hipMalloc((void**)&s->wav_win, (size_t)J * s->wav_stride * sizeof(float)));
    SCR(hipMalloc((void**)&s->prev, 2 * (size_t)n * s->prev_stride * sizeof(float)));
    SCR(hipMemset(s->prev, 0, 2 * (size_t)n * s->prev_stride * sizeof(float)));
    SCR(hipMalloc((void**)&s->out_dev, (size_t)J * s->out_stride * sizeof(float)));
    SCR(hipMalloc((void**)&s->jobs_dev, (size_t)J * sizeof(StreamJob)));
    SCR(hipMalloc((void**)&s->parity_dev, (size_t)J * sizeof(int)));
    SCR(hipHostMalloc((void**)&s->snap_host, 2 * (size_t)n * sizeof(int), hipHostMallocDefault));
    SCR(hipHostMalloc((void**)&s->out_host, (size_t)J * s->out_stride * sizeof(float), hipHostMallocDefault));
    SCR(hipHostMalloc((void**)&s->jobs_host, (size_t)J * sizeof(StreamJob), hipHostMallocDefault));
    SCR(hipHostMalloc((void**)&s->parity_host, (size_t)J * sizeof(int), hipHostMallocDefault));
    SCR(hipEventCreateWithFlags(&s->ev_snap, hipEventDisableTiming));
    s->n_dec.assign(ref_lens, ref_lens + n);                              // tokens already turned into audio: the reference codes (ref :387)
    s->prev_len.assign(n, 0);
    s->frames.assign(n, 0);
    s->done.assign(n, 0);
    SCR(hipDeviceSynchronize());
#undef SCR
    *out = s;
    return NTTS_OK;
}

extern "C" int ntts_streams_pump_begin(ntts_streams* s) {
    if (!s) return NTTS_EINVAL;
    if (s->snap_open) return sfail(s, NTTS_ESTATE, "a pump is already open (ntts_streams_pump_end first)");
    SHIP(s, hipSetDevice(s->device));
    const int rc = ntts_backbone_append_codes(s->e, s->n, s->slots.data(), s->prm.speech_base, s->prm.n_codes, s->prm.modulo, s->cache, s->cache_stride,
                                              s->clen, s->seen, s->fin);
    if (rc != NTTS_OK) return sfail(s, rc, "append_codes: %s", ntts_last_error(s->e));
    SHIP(s, hipMemcpyAsync(s->snap_host, s->clen, 2 * (size_t)s->n * sizeof(int), hipMemcpyDeviceToHost, s->est));   // clen | fin are adjacent
    SHIP(s, hipEventRecord(s->ev_snap, s->est));
    s->snap_open = true;
    return NTTS_OK;
}

extern "C" int ntts_streams_pump_wait(ntts_streams* s, int32_t* n_running) {
    if (!s || !n_running) return NTTS_EINVAL;
    if (!s->snap_open && !s->have_snap) return sfail(s, NTTS_ESTATE, "no pump is open (ntts_streams_pump_begin first)");
    SHIP(s, hipSetDevice(s->device));
    if (s->snap_open) {
        SHIP(s, hipEventSynchronize(s->ev_snap));          // waits for the steps enqueued BEFORE pump_begin only
        s->snap_open = false;
        s->have_snap = true;
    }
    int running = 0;
    for (int u = 0; u < s->n; ++u) running += (!s->done[u] && !s->snap_host[s->n + u]) ? 1 : 0;
    *n_running = running;
    return NTTS_OK;
}

extern "C" int ntts_streams_pump_end(ntts_streams* s, int32_t cap, int32_t* n_chunks, int32_t* chunk_stream, int32_t* chunk_samples,
                                     int32_t* chunk_last, const float** chunks, int64_t* row_stride, int32_t* n_running, int32_t* more) {
    if (!s || !n_chunks || !chunk_stream || !chunk_samples || !chunk_last || !chunks || !row_stride || cap < 2 * s->n) return sfail(s, NTTS_EINVAL, "bad argument (cap >= 2 n)");
    if (!s->snap_open && !s->have_snap) return sfail(s, NTTS_ESTATE, "no pump is open (ntts_streams_pump_begin first)");
    SHIP(s, hipSetDevice(s->device));
    if (s->snap_open) {
        SHIP(s, hipEventSynchronize(s->ev_snap));          // waits for the steps enqueued BEFORE pump_begin only
        s->snap_open = false;
        s->have_snap = true;
    }
    const ntts_stream_params& P = s->prm;
    const int n = s->n, hop = s->hop, need = P.chunk + P.lookforward, stride = P.chunk * hop;
    const int* clen = s->snap_host;
    const int* fin = s->snap_host + n;
    // ---- which windows exist now (ref:neutts/neutts.py:401-415 / :443-459): per stream at most one regular window per round, and its
    //      final window once it has finished and no regular window is left
    int J = 0, pending = 0, running = 0;
    std::vector<int> last_of(n, 0);
    auto add = [&](int u, int t0, int t1, int s0, int n1, bool last) {
        StreamJob& j = s->jobs_host[J];
        const bool hp = s->frames[u] > 0;
        j.u = u; j.t0 = t0; j.t1 = t1; j.s0 = s0; j.n1 = n1; j.n0 = s->prev_len[u]; j.flags = (hp ? 1 : 0) | (last ? 2 : 0);
        if (!hp) j.out_len = last ? n1 : stride;
        else j.out_len = last ? ((j.n0 > stride + n1 ? j.n0 : stride + n1) - stride) : stride;
        s->parity_host[J] = s->frames[u] & 1;
        chunk_stream[J] = u; chunk_samples[J] = j.out_len; chunk_last[J] = last ? 1 : 0;
        s->prev_len[u] = n1;
        s->frames[u]++;
        ++J;
    };
    for (int u = 0; u < n; ++u) {
        if (s->done[u]) continue;
        if (!fin[u]) ++running;
        if (clen[u] - s->n_dec[u] >= need) {
            const int nd = s->n_dec[u];
            const int t0 = nd - P.lookback - P.overlap > 0 ? nd - P.lookback - P.overlap : 0;
            const int t1 = nd + need;                      // the reference slices [start : n_dec + 31) at the moment the 30th undecoded token arrives: ends at n_dec + 30
            add(u, t0, t1, (nd - t0) * hop, (P.chunk + 2 * P.overlap) * hop, false);
            s->n_dec[u] += P.chunk;
        }
        if (clen[u] - s->n_dec[u] >= need) { ++pending; continue; }      // another regular window: next round
        if (fin[u]) {
            const int remaining = clen[u] - s->n_dec[u];
            if (remaining > 0) {
                const int t0 = clen[u] - (P.lookback + P.overlap + remaining) > 0 ? clen[u] - (P.lookback + P.overlap + remaining) : 0;
                const int s0 = (clen[u] - t0 - remaining - P.overlap) * hop;
                if (s0 < 0) return sfail(s, NTTS_EINVAL, "stream %d: final window starts before its first frame", u);
                add(u, t0, clen[u], s0, (clen[u] - t0) * hop - s0, true);
                s->n_dec[u] = clen[u];
            }
            s->done[u] = 1;
        }
    }
    *n_chunks = J;
    *chunks = s->out_host;
    *row_stride = s->out_stride;
    if (n_running) *n_running = running;
    if (more) *more = pending > 0 ? 1 : 0;
    if (pending == 0) s->have_snap = false;
    if (J == 0) return NTTS_OK;
    // ---- device: window codes -> codec pass(es) -> cross-fade -> the new samples to page-locked memory; all on the codec's stream, behind
    //      the append (ev_snap was recorded after it on the backbone's stream), beside whatever the backbone runs next
    {   // (ntts_codec_set_cu_mask re-creates the codec's stream: ask for it every time)
        void* st = nullptr;
        if (ntts_codec_stream(s->c, &st) != NTTS_OK) return sfail(s, NTTS_EINVAL, "codec stream");
        s->cst = (hipStream_t)st;
    }
    SHIP(s, hipStreamWaitEvent(s->cst, s->ev_snap, 0));
    SHIP(s, hipMemcpyAsync(s->jobs_dev, s->jobs_host, (size_t)J * sizeof(StreamJob), hipMemcpyHostToDevice, s->cst));
    SHIP(s, hipMemcpyAsync(s->parity_dev, s->parity_host, (size_t)J * sizeof(int), hipMemcpyHostToDevice, s->cst));
    StreamGatherArgs ga{};
    ga.jobs = s->jobs_dev; ga.cache = s->cache; ga.cache_stride = s->cache_stride; ga.win = s->win; ga.win_stride = s->W;
    NTTS_LAUNCH((stream_gather_kernel), dim3(J), dim3(128), s->cst, ga);
    std::vector<int> lens(J);
    for (int k = 0; k < J; ++k) lens[k] = s->jobs_host[k].t1 - s->jobs_host[k].t0;
    for (int k0 = 0; k0 < J;) {                             // as many windows per codec pass as its workspace holds
        long rows = 0;
        int k1 = k0, tmax = 0;
        while (k1 < J) {
            const int tm = lens[k1] > tmax ? lens[k1] : tmax;
            if (rows + lens[k1] + 6 > s->max_rows || (long)(k1 - k0 + 1) * tm > s->max_rows) break;
            rows += lens[k1] + 6; tmax = tm; ++k1;
        }
        if (k1 == k0) return sfail(s, NTTS_EINVAL, "the codec engine's max_rows %ld cannot hold one window", s->max_rows);
        const int rc = ntts_codec_decode_dev(s->c, k1 - k0, s->win + (size_t)k0 * s->W, s->W, lens.data() + k0, s->wav_win + (size_t)k0 * s->wav_stride,
                                             s->wav_stride, 1, nullptr);
        if (rc != NTTS_OK) return sfail(s, rc, "codec pass: %s", ntts_codec_last_error(s->c));
        k0 = k1;
    }
    StreamBlendArgs ba{};
    ba.jobs = s->jobs_dev; ba.wav = s->wav_win; ba.wav_stride = s->wav_stride; ba.prev = s->prev; ba.prev_stride = s->prev_stride;
    ba.parity = s->parity_dev; ba.n_streams = n; ba.stride = stride; ba.out = s->out_dev; ba.out_stride = s->out_stride;
    // a stream's regular and final window of one round depend on each other through `prev`: the final windows are listed behind all
    // regular ones of their stream (add() order), so two launches -- regular jobs, then final jobs -- keep the order without atomics
    for (int phase = 0; phase < 2; ++phase) {
        int a = -1, b = -1;
        for (int k = 0; k < J; ++k)
            if (((s->jobs_host[k].flags >> 1) & 1) == phase) { if (a < 0) a = k; b = k; }
        if (a < 0) continue;
        // (jobs of the two kinds interleave in the list: launch over the whole range and let the other kind return at once)
        StreamBlendArgs pa = ba;
        NTTS_LAUNCH((stream_blend_phase_kernel), dim3(b - a + 1, 8), dim3(256), s->cst, pa, a, phase);
    }
    SHIP(s, hipMemcpyAsync(s->out_host, s->out_dev, (size_t)J * s->out_stride * sizeof(float), hipMemcpyDeviceToHost, s->cst));
    SHIP(s, hipStreamSynchronize(s->cst));
    SHIP(s, hipGetLastError());
    return NTTS_OK;
}

extern "C" int ntts_streams_done(ntts_streams* s, int32_t* n_done) {
    if (!s || !n_done) return NTTS_EINVAL;
    int d = 0;
    for (int v : s->done) d += v;
    *n_done = d;
    return NTTS_OK;
}
