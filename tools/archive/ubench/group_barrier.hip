// group_barrier.hip -- what would an in-launch barrier cost if only the workgroups that share an m-block's rows took part?
// ("the o_proj -> norm -> gate/up -> down -> norm chain of a layer as a single launch whose barriers are
//  XCD-group-local".)  256 persistent workgroups (one per CU); workgroup b runs on XCD b % 8 (observed); group p = the
//  8 / ngroups XCDs that own m-block p.  A barrier = __syncthreads, lane 0: release fence -> arrive on the group's counter ->
//  poll until everyone arrived -> acquire fence, __syncthreads.  Optionally every workgroup dirties `kb` KB of fp32 partials before
//  it arrives and reads a neighbour's after (the split-K slab hand-off the barrier would be there for).
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/group_barrier tools/ubench/group_barrier.hip ; run through gpurun
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void barrier_kernel(unsigned* cnt, int iters, int xps, int gsize, float* slab, int kb, float* sink) {
    const int xcd = blockIdx.x & 7, group = xcd / xps;
    unsigned* c = cnt + group * 64;                       // one counter per group, 256 B apart
    const int nf = kb * 256;                               // floats per workgroup per phase
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (nf) {                                          // dirty this workgroup's slab (16-byte stores), as a split-K epilogue would
            float4* dst = (float4*)(slab + (size_t)blockIdx.x * nf);
            for (int i = threadIdx.x; i < nf / 4; i += 256) dst[i] = make_float4(it, i, 1.f, 2.f);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(it + 1) * gsize;
            while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (nf) {                                          // read the slab of the next workgroup of the same group (other CU, maybe other XCD)
            const int nb = xps >= 2 ? (int)(blockIdx.x ^ 1) : (int)((blockIdx.x + 8) % gridDim.x);   // a workgroup of the same group: other XCD if the group has two
            const float4* src = (const float4*)(slab + (size_t)nb * nf);
            for (int i = threadIdx.x; i < nf / 4; i += 256) acc += src[i].x;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void empty_kernel(float* sink) { if (threadIdx.x == 1234567) sink[0] = 1.f; }

int main() {
    unsigned* cnt; float *slab, *sink;
    CK(hipMalloc(&cnt, 8 * 256)); CK(hipMalloc(&slab, 256 * 64 * 1024)); CK(hipMalloc(&sink, 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int kb : {0, 16, 64})
        for (int ngroups : {1, 4, 8}) {                    // 1 = grid-wide (256 workgroups), 4 = XCD pairs (64), 8 = one XCD (32)
            const int xps = 8 / ngroups, gsize = 256 / ngroups;
            CK(hipMemset(cnt, 0, 8 * 256));
            hipLaunchKernelGGL(barrier_kernel, dim3(256), dim3(256), 0, 0, cnt, 10, xps, gsize, slab, kb, sink);   // warm
            CK(hipDeviceSynchronize());
            CK(hipMemset(cnt, 0, 8 * 256));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(barrier_kernel, dim3(256), dim3(256), 0, 0, cnt, iters, xps, gsize, slab, kb, sink);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("{\"workgroups_per_barrier\": %d, \"groups\": %d, \"dirty_kb_per_workgroup\": %d, \"us_per_phase\": %.3f}\n", gsize, ngroups, kb, ms * 1e3 / iters);
            fflush(stdout);
        }
    // the kernel boundary it would replace: dependent launches of an empty 256-workgroup kernel
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"eager_dependent_empty_launch_us\": %.3f}\n", ms * 1e3 / 2000);
    return 0;
}
