// SIMT emulator twin of neutts-air_amd/csrc/ntts/dev.h  --  TEST INFRASTRUCTURE ONLY.
//
// Compiles the *unchanged* kernel + engine sources for the host CPU so that their indexing,
// fragment layouts, synchronisation and host logic can be checked against the oracle in the
// `-m "not gpu"` suite, before any GPU minute is spent.  Each HIP thread is a fiber; wave64
// collectives (MFMA, shuffles, LDS-DMA) rendezvous the 64 lanes and follow the lane layouts
// documented for gfx950 (cdna_hip_programming.md section 3).  It is NOT a product path: the
// library it builds (libneutts_emu.so) lives under tests/ and nothing in neutts-air_amd/
// loads it; the product library fails loudly without a gfx950 device.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <functional>

#define NTTS_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define NTTS_HD inline
#define NTTS_D inline
#define NTTS_KERNEL(threads) static
#define NTTS_SHARED static thread_local __attribute__((aligned(16)))   // one "LDS" per OS thread that runs workgroups (emu.cpp)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {
void launch(const std::function<void()>& body, dim3 grid, dim3 block);
void barrier();
// one rendezvous of the calling lane's wave; returns the double-buffer index (0/1) to use
struct WaveSlots { alignas(16) unsigned char b[2][64][64]; };
WaveSlots& wave_slots();
int wave_parity();          // buffer to DEPOSIT into for the next collective
void wave_sync();           // all lanes of the wave deposited; flips parity
int wave_lanes();
}  // namespace emu

namespace ntts {

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kWave = 64;

inline float bf2f(bf16_t v) {
    uint32_t u = (uint32_t)v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline bf16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
inline float rbf(float f) { return bf2f(f2bf(f)); }

inline int lane_id() { return threadIdx.x & 63; }
inline int wave_id() { return threadIdx.x >> 6; }

template <typename T>
inline T emu_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 64, "slot");
    int p = emu::wave_parity();
    auto& s = emu::wave_slots();
    memcpy(s.b[p][lane_id()], &v, sizeof(T));
    emu::wave_sync();
    T r;
    memcpy(&r, s.b[p][src_lane & 63], sizeof(T));
    return r;
}

// v_mfma_f32_16x16x32_bf16 with the documented gfx950 layout (see the HIP dev.h).
// fp32 accumulate, k ascending: the real matrix core's internal order differs, which is exactly
// the freedom the parity tolerances allow.
inline f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    struct Dep { bf16x8 a, b; };
    int p = emu::wave_parity();
    auto& s = emu::wave_slots();
    Dep d{a, b};
    memcpy(s.b[p][lane_id()], &d, sizeof(d));
    emu::wave_sync();
    int l = lane_id(), col = l & 15, rg = l >> 4;
    f32x4 out = c;
    for (int r = 0; r < 4; ++r) {
        int row = rg * 4 + r;
        float acc = c[r];
        for (int kg = 0; kg < 4; ++kg) {
            Dep da, db;
            memcpy(&da, s.b[p][row + 16 * kg], sizeof(Dep));   // lane holding A[row][kg*8..]
            memcpy(&db, s.b[p][col + 16 * kg], sizeof(Dep));   // lane holding B[kg*8..][col]
            for (int j = 0; j < 8; ++j)
                acc += bf2f((bf16_t)da.a[j]) * bf2f((bf16_t)db.b[j]);
        }
        out[r] = acc;
    }
    return out;
}

// ---- fp16 (IEEE half) operands: conversions by bit arithmetic (no _Float16 runtime support needed on the host)
inline float h2f(bf16_t v) {
    const int s = v >> 15, e = (v >> 10) & 31, m = v & 1023;
    float r;
    if (e == 31) r = m ? NAN : INFINITY;
    else if (e == 0) r = ldexpf((float)m, -24);
    else r = ldexpf(1.0f + m / 1024.0f, e - 15);
    return s ? -r : r;
}
inline bf16_t f2h(float f) {                 // saturating round-to-nearest-even, as the HIP dev.h (clamp, then v_cvt_f16_f32)
    f = fminf(fmaxf(f, -65504.0f), 65504.0f);
    const bf16_t s = signbit(f) ? 0x8000 : 0;
    const float a = fabsf(f);
    if (a < ldexpf(1.0f, -14)) return s | (bf16_t)nearbyintf(ldexpf(a, 24));    // subnormals: multiples of 2^-24 (1024 = the smallest normal)
    int ex;
    const float fr = frexpf(a, &ex);         // a = fr * 2^ex, fr in [0.5, 1)
    float mant = nearbyintf(ldexpf(fr, 11)); // 1.m scaled to [1024, 2048]
    int e = ex - 1 + 15;
    if (mant == 2048.0f) { mant = 1024.0f; e += 1; }
    if (e >= 31) return s | 0x7bff;
    return s | (bf16_t)((e << 10) | ((int)mant - 1024));
}
inline f32x4 mfma16_f16(bf16x8 a, bf16x8 b, f32x4 c) {
    struct Dep { bf16x8 a, b; };
    int p = emu::wave_parity();
    auto& s = emu::wave_slots();
    Dep d{a, b};
    memcpy(s.b[p][lane_id()], &d, sizeof(d));
    emu::wave_sync();
    int l = lane_id(), col = l & 15, rg = l >> 4;
    f32x4 out = c;
    for (int r = 0; r < 4; ++r) {
        int row = rg * 4 + r;
        float acc = c[r];
        for (int kg = 0; kg < 4; ++kg) {
            Dep da, db;
            memcpy(&da, s.b[p][row + 16 * kg], sizeof(Dep));
            memcpy(&db, s.b[p][col + 16 * kg], sizeof(Dep));
            for (int j = 0; j < 8; ++j)
                acc += h2f((bf16_t)da.a[j]) * h2f((bf16_t)db.b[j]);
        }
        out[r] = acc;
    }
    return out;
}
template <bool F16> inline bf16_t f2op(float f) { if constexpr (F16) return f2h(f); else return f2bf(f); }
template <bool F16> inline f32x4 mfma16_op(bf16x8 a, bf16x8 b, f32x4 c) { if constexpr (F16) return mfma16_f16(a, b, c); else return mfma16(a, b, c); }

// v_mfma_f32_16x16x4_f32: lane l holds A[row l&15][k l>>4], B[k l>>4][col l&15]; k ascending, fp32 accumulate
inline f32x4 mfma16_f32(float a, float b, f32x4 c) {
    struct Dep { float a, b; };
    int p = emu::wave_parity();
    auto& s = emu::wave_slots();
    Dep d{a, b};
    memcpy(s.b[p][lane_id()], &d, sizeof(d));
    emu::wave_sync();
    int l = lane_id(), col = l & 15, rg = l >> 4;
    f32x4 out = c;
    for (int r = 0; r < 4; ++r) {
        int row = rg * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            Dep da, db;
            memcpy(&da, s.b[p][row + 16 * k], sizeof(Dep));
            memcpy(&db, s.b[p][col + 16 * k], sizeof(Dep));
            acc = fmaf(da.a, db.b, acc);
        }
        out[r] = acc;
    }
    return out;
}

// ---- fp8 e4m3fn (OCP): 1 sign, 4 exponent (bias 7), 3 mantissa bits; 0x7f = NaN, max finite 448, subnormals 2^-9 .. 7 * 2^-9
typedef __attribute__((ext_vector_type(2))) long i64x2;
constexpr float kFp8Max = 448.0f;
inline float fp82f(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r;
    if (e == 15 && m == 7) r = NAN;
    else if (e == 0) r = ldexpf((float)m, -9);
    else r = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -r : r;
}
inline unsigned char f2fp8(float f) {       // round-to-nearest-even, input clamped to +-448 by the caller
    if (f != f) return 0x7f;
    const unsigned char s = signbit(f) ? 0x80 : 0;
    float a = fabsf(f);
    if (a > kFp8Max) a = kFp8Max;
    if (a < ldexpf(1.0f, -6)) {             // subnormal range: multiples of 2^-9
        const float q = nearbyintf(ldexpf(a, 9));      // default rounding mode = to nearest even
        return s | (unsigned char)q;        // q == 8 is exactly the smallest normal (e = 1, m = 0) = 0x08
    }
    int ex;
    const float fr = frexpf(a, &ex);        // a = fr * 2^ex, fr in [0.5, 1)
    float mant = nearbyintf(ldexpf(fr, 4)); // 1.mmm scaled to [8, 16]
    int e = ex - 1 + 7;
    if (mant == 16.0f) { mant = 8.0f; e += 1; }
    if (e > 15 || (e == 15 && mant > 14.0f)) return s | 0x7e;   // 448
    return s | (unsigned char)((e << 3) | ((int)mant - 8));
}
inline unsigned short f2fp8x2(float a, float b) {
    a = fminf(fmaxf(a, -kFp8Max), kFp8Max);
    b = fminf(fmaxf(b, -kFp8Max), kFp8Max);
    return (unsigned short)(f2fp8(a) | (f2fp8(b) << 8));
}
inline unsigned char f2fp8c(float a) { return (unsigned char)(f2fp8x2(a, 0.f) & 0xff); }
// v_mfma_f32_16x16x32_fp8_fp8: same lane layout as mfma16 with 8 e4m3 bytes per lane (k ascending, fp32 accumulate)
inline f32x4 mfma16_fp8(long a, long b, f32x4 c) {
    struct Dep { long a, b; };
    int p = emu::wave_parity();
    auto& s = emu::wave_slots();
    Dep d{a, b};
    memcpy(s.b[p][lane_id()], &d, sizeof(d));
    emu::wave_sync();
    int l = lane_id(), col = l & 15, rg = l >> 4;
    f32x4 out = c;
    for (int r = 0; r < 4; ++r) {
        int row = rg * 4 + r;
        float acc = c[r];
        for (int kg = 0; kg < 4; ++kg) {
            Dep da, db;
            memcpy(&da, s.b[p][row + 16 * kg], sizeof(Dep));
            memcpy(&db, s.b[p][col + 16 * kg], sizeof(Dep));
            for (int j = 0; j < 8; ++j)
                acc += fp82f((unsigned char)(da.a >> (8 * j))) * fp82f((unsigned char)(db.b >> (8 * j)));
        }
        out[r] = acc;
    }
    return out;
}

inline float shfl_xor(float v, int m) { return emu_exchange(v, lane_id() ^ m); }
inline int shfl_xor(int v, int m) { return emu_exchange(v, lane_id() ^ m); }
inline float shfl(float v, int src) { return emu_exchange(v, src); }
inline int shfl(int v, int src) { return emu_exchange(v, src); }

inline unsigned long long ballot(bool pr) {
    int par = emu::wave_parity();
    auto& s = emu::wave_slots();
    int v = pr ? 1 : 0;
    memcpy(s.b[par][lane_id()], &v, 4);
    emu::wave_sync();
    unsigned long long m = 0;
    for (int l = 0; l < emu::wave_lanes(); ++l) { int x; memcpy(&x, s.b[par][l], 4); if (x) m |= 1ull << l; }
    return m;
}
inline int popc64(unsigned long long m) { return __builtin_popcountll(m); }

inline void sync() { emu::barrier(); }

// LDS-DMA: destination = lane 0's base + lane*16, whatever the other lanes passed (hardware
// takes the base from M0, i.e. a readfirstlane) -- catches per-lane-destination mistakes.
inline void glds16(const void* gsrc, void* lds_wave_base) {
    struct Dep { const void* g; void* l; };
    int p = emu::wave_parity();
    auto& s = emu::wave_slots();
    Dep d{gsrc, lds_wave_base};
    memcpy(s.b[p][lane_id()], &d, sizeof(d));
    emu::wave_sync();
    Dep d0;
    memcpy(&d0, s.b[p][0], sizeof(d0));
    memcpy((char*)d0.l + lane_id() * 16, gsrc, 16);
}
inline void glds16_nt(const void* gsrc, void* lds_wave_base) { glds16(gsrc, lds_wave_base); }
inline unsigned int atomic_add_global(unsigned int* p, unsigned int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }   // (workgroups run on several OS threads)
inline unsigned int atomic_add_lds(unsigned int* p, unsigned int v) { unsigned int o = *p; *p = o + v; return o; }
inline unsigned int atomic_max_global_u32(unsigned int* p, unsigned int v) {
    unsigned int o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
NTTS_D unsigned long long now_ticks() { return 0; }   // no clock on the emulator
inline void wait_vmem() {}
inline void sync_keep_dma() { emu::barrier(); }
inline void lds_barrier() { emu::barrier(); }
inline void sched_fence() {}
template <int N>
inline void wait_vmem_le() {}

inline float fexp(float x) { return expf(x); }
inline float fexp2(float x) { return exp2f(x); }
inline float fexp_neg(float x) {
    const float hi = 1.44269502162933349609375f, lo = 1.925963033500011e-8f;
    const float t = x * hi;
    float r = fmaf(x, hi, -t);
    r = fmaf(x, lo, r);
    const float y = exp2f(t);
    return fmaf(y, r * 0.693147180559945309f, y);
}
inline int opaque(int v) { return v; }
inline int opaque_u(int v) { return v; }
inline float fexp_neg8(float d) {
    const float hi = 1.44269502162933349609375f * 0.125f, lo = 1.925963033500011e-8f * 0.125f;
    const float t = d * hi;
    float r = fmaf(d, hi, -t);
    r = fmaf(d, lo, r);
    const float y = exp2f(t);
    return fmaf(y, r * 0.693147180559945309f, y);
}
inline float fdiv_r(float e, float d, float r) {
    const float q = e * r;
    return fmaf(fmaf(-q, d, e), r, q);
}
inline float frcp_refined(float d) {
    float r = 1.0f / d;
    return fmaf(fmaf(-d, r, 1.0f), r, r);
}
inline float frsqrt_exact(float x) { return 1.0f / sqrtf(x); }
inline float frcp_raw(float d) { return 1.0f / d; }
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
inline f32x2_t rbf2(f32x2_t v) { f32x2_t o; o[0] = rbf(v[0]); o[1] = rbf(v[1]); return o; }
inline unsigned int pack_bf2(float a, float b) { return (unsigned int)f2bf(a) | ((unsigned int)f2bf(b) << 16); }
inline bool any_lane(bool p) { return ballot(p) != 0; }

template <typename T>
inline T ld16(const void* p) {
    T v;
    memcpy(&v, p, sizeof(T));
    return v;
}
template <typename T>
inline T ld16_nt(const void* p) { return ld16<T>(p); }

}  // namespace ntts

#define NTTS_LAUNCH(kern, grid, block, stream, ...) \
    emu::launch([=]() { kern(__VA_ARGS__); }, grid, block)

// ---------------------------------------------------------------------------------------------
// minimal HIP runtime facade: device memory is host memory, streams are immediate
// ---------------------------------------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
typedef struct emuEvent { double t; }* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotSupported = 801 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamNonBlocking = 1 };
struct hipDeviceProp_t { char gcnArchName[64]; int multiProcessorCount; size_t totalGlobalMem; };

inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    strcpy(p->gcnArchName, "gfx950:emu"); p->multiProcessorCount = 256; p->totalGlobalMem = 1ull << 34;
    return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? 0 : 2; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
enum { hipHostMallocDefault = 0 };
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? 0 : 2; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return 0; }
inline hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind) {
    for (size_t r = 0; r < h; ++r) memmove((char*)d + r * dp, (const char*)s + r * sp, w);
    return 0;
}
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return 0; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipGetLastError() { return 0; }
double emu_now_ms();
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent{0}; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emuEvent{0}; return 0; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = emu_now_ms(); return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t) {
    return hipMemcpy2D(d, dp, s, sp, w, h, k);
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return 0; }
// graphs: not emulated -- the engine falls back to direct launches when capture is unsupported
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return hipErrorNotSupported; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return hipErrorNotSupported; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return 0; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return 0; }
