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
__global__ __launch_bounds__(1024) void lp_kernel(const char* src, long region_bytes, int shared, int passes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nw = blockDim.x >> 6, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long base = (shared ? (long)(blockIdx.x & 7) : (long)blockIdx.x) * region_bytes;
    const char* p = src + base;
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
                    v[d] = *(const u32x4*)(p + cc * 1024 + lane * 16);
                }
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if constexpr (KIND == 2) *(u32x4*)(my_lds + d * 1024 + lane * 16) = v[d];
                    else acc ^= v[d];
                }
            }
        }
    }
    if constexpr (KIND != 1) { __syncthreads(); acc = *(u32x4*)(lds + ((threadIdx.x * 16) & 4095)); }
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[0] = acc[2] ^ acc[3];
}

template <int KIND, int D>
static float run(const char* src, long region, int shared, int passes, int nwg, int nw, unsigned* sink, int reps) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const size_t lds = (size_t)nw * D * 1024;
    lp_kernel<KIND, D><<<nwg, nw * 64, lds>>>(src, region, shared, passes, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) lp_kernel<KIND, D><<<nwg, nw * 64, lds>>>(src, region, shared, passes, sink);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const long total = 2L << 30;
    char* src; unsigned* sink;
    CHECK(hipMalloc(&src, total)); CHECK(hipMemset(src, 1, total)); CHECK(hipMalloc(&sink, 64));
    struct Cfg { const char* name; long region; int shared; int passes; int nwg; };
    std::vector<Cfg> cfgs = {
        {"L2-shared 512KB x16 passes, 256 WG", 512 << 10, 1, 16, 256},
        {"private 4MB x1 pass (HBM/MALL 1GB), 256 WG", 4 << 20, 0, 1, 256},
        {"private 512KB x16 passes (128MB: MALL/L2), 256 WG", 512 << 10, 0, 16, 256},
        {"L2-shared 512KB x16 passes, 128 WG", 512 << 10, 1, 16, 128},
        {"private 4MB x1 pass, 128 WG", 4 << 20, 0, 1, 128},
    };
    for (auto& c : cfgs) {
        printf("== %s\n", c.name);
        for (int nw : {4, 8, 16}) {
            const double bytes = (double)c.region * c.passes * c.nwg;
#define ROW(KIND, D) { float ms = run<KIND, D>(src, c.region, c.shared, c.passes, c.nwg, nw, sink, 5); \
            printf("  kind %d  waves %2d  depth %d : %8.1f us  %7.2f TB/s  %6.1f GB/s per WG\n", KIND, nw, D, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / c.nwg); }
            ROW(0, 2) ROW(0, 4) ROW(0, 8)
            ROW(1, 2) ROW(1, 4) ROW(1, 8)
            ROW(2, 4) ROW(2, 8)
        }
    }
    return 0;
}
