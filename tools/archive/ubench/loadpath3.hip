// loadpath.hip -- per-CU load-path microbenchmark (diagnostic tool, not part of the product library).
// Question: what does ONE CU pull through its vector-memory path, by instruction kind and data residency?
//   kind 0: global_load_lds_dwordx4 (LDS-DMA), kind 1: global_load_dwordx4 -> VGPR (values xor-folded),
//   kind 2: global_load_dwordx4 -> VGPR -> ds_write_b128 (register staging into LDS)
// Geometry: one workgroup of NW waves per CU-slot (grid = nwg), every wave keeps D 1-KB wave-loads in flight.
// Residency: region_bytes per workgroup; "shared" makes all workgroups of an XCD (wg % 8) read the SAME region (L2 hits after
// the first pass), otherwise private regions (HBM / Infinity Cache depending on the total).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int KIND, int D>
__global__ __launch_bounds__(1024) void lp_kernel(const char* src, long region_bytes, int shared, int passes, unsigned* sink, long off = 0) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nw = blockDim.x >> 6, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long base = (shared ? (long)(blockIdx.x & 7) : (long)blockIdx.x) * region_bytes;
    const char* p = src + base + off;
    const long chunks = region_bytes >> 10;             // 1-KB wave-loads in the region
    u32x4 acc = {0, 0, 0, 0};
    char* my_lds = lds + (long)w * D * 1024;
    for (int it = 0; it < passes; ++it) {
        for (long c = w; c < chunks; c += (long)nw * D) {
            if constexpr (KIND == 0) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    long cc = c + (long)d * nw; if (cc >= chunks) cc = c;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + cc * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(my_lds + d * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                u32x4 v[D];
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    long cc = c + (long)d * nw; if (cc >= chunks) cc = c;
                    if constexpr (KIND == 3) {   // fragment-shaped: 1-KB chunk cc = half ks of 2-KB block cc/2: 16 rows x 64 B pieces
                        const long blk = cc >> 1; const int ks = (int)(cc & 1);
                        v[d] = *(const u32x4*)(p + blk * 2048 + (lane & 15) * 128 + ks * 64 + (lane >> 4) * 16);
                    } else v[d] = *(const u32x4*)(p + cc * 1024 + lane * 16);
                }
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if constexpr (KIND == 2) *(u32x4*)(my_lds + d * 1024 + lane * 16) = v[d];
                    else acc ^= v[d];
                }
            }
        }
    }
    if constexpr (KIND == 0 || KIND == 2) { __syncthreads(); acc = *(u32x4*)(lds + ((threadIdx.x * 16) & 4095)); }
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[0] = acc[2] ^ acc[3];
}

template <int KIND, int D>
static float run(const char* src, long region, int shared, int passes, int nwg, int nw, unsigned* sink, int reps, long rot = 0, long total = 0) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const size_t lds = (size_t)nw * D * 1024;
    lp_kernel<KIND, D><<<nwg, nw * 64, lds>>>(src, region, shared, passes, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) {
        const long off = rot ? ((long)(r + 1) * rot) % (total - rot) : 0;     // rot > 0: every launch reads fresh (HBM-cold) memory
        lp_kernel<KIND, D><<<nwg, nw * 64, lds>>>(src, region, shared, passes, sink, off);
    }
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const long total = 2L << 30;
    char* src; unsigned* sink;
    CHECK(hipMalloc(&src, total)); CHECK(hipMemset(src, 1, total)); CHECK(hipMalloc(&sink, 64));
    printf("== short cold streams: every launch reads fresh memory; time per launch INCLUDES the back-to-back launch gap\n");
    {   // empty-ish kernel: region 1 KB
        float ms = run<1, 2>(src, 1024, 0, 1, 256, 4, sink, 20, 0, 0);
        printf("  (1 KB per WG, 256 WG: %.2f us per launch = the launch floor)\n", ms * 1e3);
    }
    for (int nwg : {152, 256}) for (long kb : {64L, 128L}) for (int nw : {4, 8}) {
        const long region = kb << 10; const long rot = (long)nwg * region; const double bytes = (double)region * nwg;
#define ROW2(KIND, D) { float ms = run<KIND, D>(src, region, 0, 1, nwg, nw, sink, 20, rot, total); \
        printf("  %3d WG x %3ld KB (%5.1f MB) kind %d waves %2d depth %2d : %6.2f us/launch  %5.2f TB/s\n", nwg, kb, bytes / 1e6, KIND, nw, D, ms * 1e3, bytes / ms / 1e9); }
        ROW2(1, 8) ROW2(1, 16) ROW2(3, 8) ROW2(3, 16)
    }
    return 0;
}
