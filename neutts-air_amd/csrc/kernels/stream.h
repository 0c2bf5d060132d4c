// stream.h -- the streaming post-process of ref:neutts/neutts.py:401-465 on the device: window assembly and the triangular cross-fade.
//
// Replaces, per stream and per 25-token chunk:
//   token_cache[start:end]                                  ref:neutts/neutts.py:407-415, :449-455   (window of <= 81 codes)
//   recon[sample_start:sample_end]                          ref:neutts/neutts.py:416-429, :456-459   (27 frames of the decoded window)
//   _linear_overlap_add(audio_cache, stride)[n_dec_samp:]   ref:neutts/neutts.py:46-70, :433-440, :461-465
// The reference re-blends its whole audio cache for every chunk; frames are 27 hops long at a stride of 25, so only the PREVIOUS frame
// can touch the samples a new frame releases (neutts.py _StreamBlender): the kernel keeps one previous frame per stream.
// Arithmetic = numpy's, operation for operation, so that the result is bit-identical to the host path (tests compare them):
//   t   = np.linspace(0, 1, n + 2, dtype=float32)[1:-1]   -> float32( (double)(x + 1) * (1.0 / (double)(n + 1)) )
//   tri = np.abs(0.5 - (t - 0.5))                          -> float32 ops
//   mixed += tri * f ; weight_sum += tri                   -> separately rounded products and sums (no fma contraction), frame 0 first
//   mixed / weight_sum                                     -> IEEE float32 division
#pragma once
#include <ntts/dev.h>

namespace ntts {

// one decodable window: stream `u` takes codes cache[u][t0 .. t1) -> window row `row`; of the decoded window the samples [s0, s0 + n1)
// are the new frame; `flags` bit 0: the stream has a previous frame (of n0 samples), bit 1: last chunk of the stream (emit everything)
struct StreamJob { int u, t0, t1, s0, n1, n0, flags, out_len; };

struct StreamGatherArgs {
    const StreamJob* jobs;
    const int* cache;        // [n][cache_stride]
    int cache_stride;
    int* win;                // [jobs][win_stride]
    int win_stride;
};
NTTS_KERNEL(128) void stream_gather_kernel(StreamGatherArgs p) {
    const StreamJob j = p.jobs[blockIdx.x];
    for (int t = threadIdx.x; t < j.t1 - j.t0; t += 128) p.win[(long)blockIdx.x * p.win_stride + t] = p.cache[(long)j.u * p.cache_stride + j.t0 + t];
}

NTTS_D float stream_tri(int x, int n) {
    const float t = (float)((double)(x + 1) * (1.0 / (double)(n + 1)));
    return __builtin_fabsf(0.5f - (t - 0.5f));
}
// products and sums rounded one by one, like numpy's element-wise passes (the compiler must not contract them into fmas)
NTTS_D float mul_rn(float a, float b) { volatile float r = a * b; return r; }
NTTS_D float add_rn(float a, float b) { volatile float r = a + b; return r; }

struct StreamBlendArgs {
    const StreamJob* jobs;
    const float* wav;        // [jobs][wav_stride] decoded windows
    long wav_stride;
    float* prev;             // [2][n][prev_stride] previous frame per stream, ping-pong (parity = frames emitted so far & 1)
    long prev_stride;
    const int* parity;       // [jobs] which half holds the stream's previous frame; the new frame goes to the other half
    int n_streams;
    int stride;              // samples per chunk (streaming_stride_samples)
    float* out;              // [jobs][out_stride]
    long out_stride;
};
// job0 / phase: the launch covers jobs job0 .. job0 + gridDim.x - 1 and handles those whose "last chunk" flag equals `phase` (a stream's
// final window must be blended AFTER its regular window of the same round: two launches, regular jobs first)
NTTS_KERNEL(256) void stream_blend_phase_kernel(StreamBlendArgs p, int job0, int phase) {
    const int jb = job0 + blockIdx.x;
    const StreamJob j = p.jobs[jb];
    if (((j.flags >> 1) & 1) != phase) return;    // (block-uniform)
    const float* f = p.wav + (long)jb * p.wav_stride + j.s0;
    const int par = p.parity[jb];
    const float* pv = p.prev + ((long)par * p.n_streams + j.u) * p.prev_stride;
    float* nv = p.prev + ((long)(par ^ 1) * p.n_streams + j.u) * p.prev_stride;
    float* o = p.out + (long)jb * p.out_stride;
    const bool has_prev = j.flags & 1;
    const int st = p.stride;
    for (int x = blockIdx.y * 256 + threadIdx.x; x < j.out_len; x += gridDim.y * 256) {
        // output sample x is position q of the blended signal: q = x without a previous frame, q = stride + x with one
        float mixed, wsum;
        if (!has_prev) {
            const float w1 = stream_tri(x, j.n1);
            mixed = add_rn(0.f, mul_rn(w1, f[x]));
            wsum = add_rn(0.f, w1);
        } else {
            const int q = st + x;
            mixed = 0.f; wsum = 0.f;
            if (q < j.n0) { const float w0 = stream_tri(q, j.n0); mixed = add_rn(mixed, mul_rn(w0, pv[q])); wsum = add_rn(wsum, w0); }
            if (x < j.n1) { const float w1 = stream_tri(x, j.n1); mixed = add_rn(mixed, mul_rn(w1, f[x])); wsum = add_rn(wsum, w1); }
        }
        o[x] = mixed / wsum;
    }
    for (int x = blockIdx.y * 256 + threadIdx.x; x < j.n1; x += gridDim.y * 256) nv[x] = f[x];   // this frame is the next chunk's previous one
}

}  // namespace ntts
