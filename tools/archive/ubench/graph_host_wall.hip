// graph_host_wall.hip -- what does the LAUNCHING THREAD pay per hipGraphLaunch, and is it per node?
// (host_wall_decode_call = 249 ms for 249 replays of the 171-node decode graph.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/graph_host_wall tools/ubench/graph_host_wall.hip
//   tools/ubench/graph_host_wall            (through gpurun; one JSON line per case)
// Cases: graphs of N dependent launches (57 / 171 / 342 / 1368) of a kernel that spins for ~D us (0 / 5), replayed
//   (a) onto an IDLE stream (sync after every launch): the call's own cost,
//   (b) 64 times back to back without a sync: the call's cost while the hardware queue still holds earlier replays
//       (per-call times: first, median, max -- back-pressure shows as a jump once the queue is full),
//   (c) the same N launches issued eagerly (hipLaunchKernelGGL), per launch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(int* sink, long ticks) {   // ~ticks of the 100 MHz constant clock
    const long t0 = wall_clock64();
    if (ticks > 0) while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && sink[blockIdx.x & 1023] == 0x7fffffff) sink[0] = 1;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    int* sink = nullptr;
    CK(hipMalloc(&sink, 4096)); CK(hipMemset(sink, 0, 4096));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int grid = 256;
    for (int dur_us : {0, 5}) {
        const long ticks = dur_us * 100L;
        for (int n : {57, 171, 342, 1368}) {
            hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, st, sink, ticks);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            // (a) idle stream
            std::vector<double> idle;
            double gpu_idle = 0;
            for (int r = 0; r < 12; ++r) {
                const double t0 = now_us();
                CK(hipGraphLaunch(ge, st));
                const double t1 = now_us();
                CK(hipStreamSynchronize(st));
                const double t2 = now_us();
                idle.push_back(t1 - t0);
                gpu_idle += t2 - t0;
            }
            std::sort(idle.begin(), idle.end());
            // (b) back to back
            const int reps = 64;
            std::vector<double> bb(reps);
            const double tb0 = now_us();
            for (int r = 0; r < reps; ++r) { const double t0 = now_us(); CK(hipGraphLaunch(ge, st)); bb[r] = now_us() - t0; }
            const double tb1 = now_us();
            CK(hipStreamSynchronize(st));
            const double tb2 = now_us();
            std::vector<double> bs = bb; std::sort(bs.begin(), bs.end());
            // (c) eager
            CK(hipStreamSynchronize(st));
            const double te0 = now_us();
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, st, sink, ticks);
            const double te1 = now_us();
            CK(hipStreamSynchronize(st));
            const double te2 = now_us();
            printf("{\"nodes\": %d, \"kernel_us\": %d, \"idle_launch_host_us_median\": %.1f, \"idle_launch_host_us_per_node\": %.3f, \"idle_launch_to_done_us\": %.1f, "
                   "\"b2b_first_us\": %.1f, \"b2b_median_us\": %.1f, \"b2b_max_us\": %.1f, \"b2b_host_total_us\": %.1f, \"b2b_wall_total_us\": %.1f, \"b2b_gpu_us_per_graph\": %.1f, "
                   "\"eager_host_us_per_launch\": %.3f, \"eager_wall_us\": %.1f}\n",
                   n, dur_us, idle[idle.size() / 2], idle[idle.size() / 2] / n, gpu_idle / 12, bb[0], bs[reps / 2], bs[reps - 1], tb1 - tb0, tb2 - tb0, (tb2 - tb0) / reps,
                   (te1 - te0) / n, te2 - te0);
            fflush(stdout);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}
