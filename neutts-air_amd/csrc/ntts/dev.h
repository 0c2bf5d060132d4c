// dev.h -- gfx950 device primitives used by every kernel (wave64, MFMA 16x16x32 bf16, LDS-DMA).
// The kernels only speak this vocabulary, so their logic can also be desk-checked on a CPU by the
// SIMT emulator under tests/simt_emu/ (test infrastructure, never part of the product build).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NTTS_HD __host__ __device__ __forceinline__
#define NTTS_D __device__ __forceinline__
#define NTTS_KERNEL(threads) static __global__ __launch_bounds__(threads)
#define NTTS_SHARED __shared__ __attribute__((aligned(16)))

namespace ntts {

typedef unsigned short bf16_t;  // storage type: bf16 bit pattern
typedef __attribute__((ext_vector_type(8))) short bf16x8;  // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;   // one MFMA C/D fragment
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kWave = 64;

NTTS_D float bf2f(bf16_t v) { return __builtin_bit_cast(float, (unsigned int)v << 16); }
// round-to-nearest-even, NaN preserved: lowers to v_cvt_pk_bf16_f32 on gfx950
NTTS_D bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
// value of a float after one bf16 rounding
NTTS_D float rbf(float f) { return bf2f(f2bf(f)); }

NTTS_D int lane_id() { return threadIdx.x & 63; }
NTTS_D int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// D = A(16x32) * B(32x16) + C on the matrix core.  Lane l holds A[row l&15][k (l>>4)*8 .. +7],
// B[k (l>>4)*8 .. +7][col l&15]; D/C: col l&15, rows (l>>4)*4 + r, r = 0..3.
NTTS_D f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- fp16 operands (the codec's default GEMM-operand format: 11 significant bits instead of bf16's 8 at the SAME matrix-core rate;
//      range 6.1e-5 .. 65504 for normal numbers -- post-norm activations and weights are O(1-10); conversions saturate instead of
//      producing inf).  Storage type stays the 16-bit pattern (bf16_t / bf16x8): staging, LDS rings and swizzles are format-blind.
NTTS_D bf16_t f2h(float f) {
    f = __builtin_fminf(__builtin_fmaxf(f, -65504.0f), 65504.0f);      // (NaN propagates through fmin/fmax as the other operand: a NaN input becomes -65504 -- the codec never makes one)
    return __builtin_bit_cast(unsigned short, (_Float16)f);            // v_cvt_f16_f32, round-to-nearest-even
}
NTTS_D float h2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// v_mfma_f32_16x16x32_f16: lane layout of mfma16, operands read as IEEE half
NTTS_D f32x4 mfma16_f16(bf16x8 a, bf16x8 b, f32x4 c) {
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// 16-bit operand of format F16 ? half : bf16
template <bool F16> NTTS_D bf16_t f2op(float f) { if constexpr (F16) return f2h(f); else return f2bf(f); }
template <bool F16> NTTS_D f32x4 mfma16_op(bf16x8 a, bf16x8 b, f32x4 c) { if constexpr (F16) return mfma16_f16(a, b, c); else return mfma16(a, b, c); }

// fp32 matrix core (v_mfma_f32_16x16x4_f32): D = A(16x4) * B(4x16) + C.  Lane l holds A[row l&15][k l>>4] and
// B[k l>>4][col l&15] (one float each); D/C as mfma16.  The reference-encoding path (kernels/enc.h) computes in fp32.
NTTS_D f32x4 mfma16_f32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---- fp8 (OCP e4m3fn on gfx950: max 448, no inf) -- weights and GEMM-input activations of the fp8 model variant
typedef __attribute__((ext_vector_type(2))) long i64x2;   // one 16-byte LDS chunk = two fp8 MFMA fragments (8 fp8 each)
constexpr float kFp8Max = 448.0f;
// D = A(16x32) * B(32x16) + C with e4m3 operands: lane l holds A[row l&15][k (l>>4)*8 .. +7] as 8 bytes (same layout as mfma16)
NTTS_D f32x4 mfma16_fp8(long a, long b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, 0, 0, 0); }
// two floats -> two e4m3 bytes (round-to-nearest-even, clamped to +-448 first: e4m3fn has no inf, an overflow would be NaN)
NTTS_D unsigned short f2fp8x2(float a, float b) {
    a = __builtin_fminf(__builtin_fmaxf(a, -kFp8Max), kFp8Max);
    b = __builtin_fminf(__builtin_fmaxf(b, -kFp8Max), kFp8Max);
    return (unsigned short)(__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffff);
}
NTTS_D unsigned char f2fp8c(float a) { return (unsigned char)(f2fp8x2(a, 0.f) & 0xff); }
NTTS_D float fp82f(unsigned char v) { return __builtin_amdgcn_cvt_f32_fp8((int)v, 0); }

NTTS_D float shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
NTTS_D int shfl_xor(int v, int m) { return __shfl_xor(v, m, 64); }
NTTS_D float shfl(float v, int src) { return __shfl(v, src, 64); }
NTTS_D int shfl(int v, int src) { return __shfl(v, src, 64); }

NTTS_D unsigned long long ballot(bool p) { return __ballot(p); }     // bit l = lane l's predicate (wave64)
NTTS_D int popc64(unsigned long long m) { return __popcll(m); }

NTTS_D void sync() { __syncthreads(); }

// LDS-DMA: every lane supplies its own 16-byte global source; the wave's 64 pieces land at
// lds_wave_base + lane*16 (the destination is wave-uniform base + lane-linear, never a scatter).
NTTS_D void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// the same with the non-temporal cache policy (aux = 2, "nt"): for bytes ONE workgroup reads ONCE (decode-step weight streams);
// MI355X_MICROARCH.md price list row nt-weights: issued -> landed -18 %, 5-10 % per decode layer.  Never for re-read operands.
NTTS_D void glds16_nt(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}
// 16-byte global load with the non-temporal policy into registers (streamed-once K/V pages, weight fragments)
template <typename T>
NTTS_D T ld16_nt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const T*>(p)); }
// all of this wave's outstanding vector-memory ops (incl. LDS-DMA) have landed
NTTS_D void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// at most N of this wave's vector-memory ops are still outstanding (they retire in issue order)
template <int N>
NTTS_D void wait_vmem_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Workgroup barrier that does NOT drain in-flight LDS-DMA: __syncthreads() carries a vmcnt(0) whenever a
// global_load_lds is outstanding, which collapses a multi-tile prefetch ring to depth 1.  This one only retires
// the wave's own LDS reads/writes (lgkmcnt) before the rendezvous; DMA completion is the caller's counted vmcnt.
NTTS_D void sync_keep_dma() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// The same instruction sequence where the point is that ordinary global loads (register prefetch rings) stay in flight:
// a workgroup barrier that orders LDS traffic only.  Use it when what the barrier publishes lives in LDS.
NTTS_D void lds_barrier() { sync_keep_dma(); }
// nothing is scheduled across this point by the compiler (issue order of memory requests matters: in-order returns)
NTTS_D void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// the value, opaque to the optimiser, with a side effect where it stands: code that uses it is neither hoisted out of its loop nor
// if-converted out of its (wave-uniform) branch
NTTS_D int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
NTTS_D int opaque_u(int v) { asm volatile("" : "+s"(v)); return v; }   // the same for a wave-uniform value (stays in a scalar register)

NTTS_D unsigned int atomic_add_global(unsigned int* p, unsigned int v) { return atomicAdd(p, v); }
NTTS_D unsigned int atomic_add_lds(unsigned int* p, unsigned int v) { return atomicAdd(p, v); }
NTTS_D unsigned int atomic_max_global_u32(unsigned int* p, unsigned int v) { return atomicMax(p, v); }
// constant-rate timestamp (s_memrealtime, 100 MHz): phase timelines of a kernel (diagnostics only)
NTTS_D unsigned long long now_ticks() { return wall_clock64(); }

NTTS_D float fexp(float x) { return expf(x); }
// 2^x as one v_exp_f32 (~1 ulp; flushes to 0 for very negative x): callers whose bar is a tolerance, not a rounding contract
NTTS_D float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
// exp(x) for FINITE x <= ~0 (softmax arguments after the max is subtracted; masked scores are -1e30, never -inf):
// x*log2(e) is split into its rounded product t and the exact remainder r (fma), exp2(t) is one v_exp_f32 and the
// remainder is a first-order correction.  ~1.5 ulp, 6 instructions, no range checks (libm's expf is ~12 with them).
NTTS_D float fexp_neg(float x) {
    const float hi = 1.44269502162933349609375f, lo = 1.925963033500011e-8f;   // log2(e) = hi + lo
    const float t = x * hi;
    float r = __builtin_fmaf(x, hi, -t);
    r = __builtin_fmaf(x, lo, r);
    const float y = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(y, r * 0.693147180559945309f, y);
}
// exp(d / 8) for finite d <= 0: fexp_neg(d * 0.125f) bit for bit with the exact 2^-3 scaling folded into the constants (every product
// and fma below is the one fexp_neg forms, scaled by a power of two on both sides of its rounding)
NTTS_D float fexp_neg8(float d) {
    const float hi = 1.44269502162933349609375f * 0.125f, lo = 1.925963033500011e-8f * 0.125f;
    const float t = d * hi;
    float r = __builtin_fmaf(d, hi, -t);
    r = __builtin_fmaf(d, lo, r);
    const float y = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(y, r * 0.693147180559945309f, y);
}
// e / d given r ~ 1/d (v_rcp_f32 refined once): one Newton step on the quotient, within ~0.5 ulp of IEEE division
NTTS_D float fdiv_r(float e, float d, float r) {
    const float q = e * r;
    return __builtin_fmaf(__builtin_fmaf(-q, d, e), r, q);
}
NTTS_D float frcp_refined(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
NTTS_D float frsqrt_exact(float x) { return 1.0f / sqrtf(x); }
NTTS_D float frcp_raw(float d) { return __builtin_amdgcn_rcpf(d); }   // v_rcp_f32 (frcp_refined's first step)
// two floats after one bf16 rounding each: ONE v_cvt_pk_bf16_f32 + two unpacks
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
NTTS_D f32x2_t rbf2(f32x2_t v) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    const unsigned int u = __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf2_t));
    f32x2_t o;
    o[0] = __builtin_bit_cast(float, u << 16);
    o[1] = __builtin_bit_cast(float, u & 0xffff0000u);
    return o;
}
// two floats rounded to bf16, packed: word 0 = bf16(a), word 1 = bf16(b) (one v_cvt_pk_bf16_f32)
NTTS_D unsigned int pack_bf2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{a, b}, bf2_t));
}
// true on every lane if the predicate holds on any lane of the wave
NTTS_D bool any_lane(bool p) { return ballot(p) != 0; }

template <typename T>
NTTS_D T ld16(const void* p) { return *reinterpret_cast<const T*>(p); }

}  // namespace ntts

// host-side launch: NTTS_LAUNCH((kernel<..>), grid, block, stream, args...)
#define NTTS_LAUNCH(kern, grid, block, stream, ...) hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__)
