// SIMT emulator runtime (fibers + wave rendezvous).  TEST INFRASTRUCTURE ONLY -- see ntts/dev.h here.
#include <ntts/dev.h>

#include <sys/mman.h>
#include <time.h>

#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

extern "C" void emu_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

double emu_now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

namespace emu {

constexpr size_t kStack = 256 * 1024;
constexpr int kMaxThreads = 1024;

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = true;
    dim3 tid;
};
struct Wave {
    int arrived = 0, gen = 0, lanes = 0;
    WaveSlots slots;
};

static Fiber g_fib[kMaxThreads];
static std::vector<Wave> g_waves(kMaxThreads / 64);
static void* g_sched_sp = nullptr;
static int g_cur = -1, g_nthreads = 0, g_alive = 0;
static int g_bar_arrived = 0, g_bar_gen = 0;
static const std::function<void()>* g_body = nullptr;
static long g_progress = 0;

static void yield() { emu_ctx_switch(&g_fib[g_cur].sp, g_sched_sp); }

static void fiber_main() {
    (*g_body)();
    g_fib[g_cur].done = true;
    --g_alive;
    ++g_progress;
    if (g_bar_arrived > 0 && g_bar_arrived == g_alive) {   // a workgroup barrier counts the threads still alive (as s_barrier does
        g_bar_arrived = 0;                                 // with terminated waves): the last exit may be what completes it
        ++g_bar_gen;
    }
    yield();
    abort();
}

static void prepare(Fiber& f) {
    if (!f.stack) {
        f.stack = (char*)mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (f.stack == MAP_FAILED) { perror("mmap"); abort(); }
    }
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;               // fake return address of fiber_main (keeps rsp % 16 == 8 at entry)
    *--sp = (void*)&fiber_main;    // popped by `ret`
    for (int i = 0; i < 6; ++i) *--sp = nullptr;  // r15 r14 r13 r12 rbx rbp
    f.sp = sp;
    f.done = false;
}

void barrier() {
    int gen = g_bar_gen;
    ++g_progress;
    if (++g_bar_arrived == g_alive) {
        g_bar_arrived = 0;
        ++g_bar_gen;
        return;
    }
    while (g_bar_gen == gen) yield();
}

WaveSlots& wave_slots() { return g_waves[g_cur >> 6].slots; }
int wave_parity() { return g_waves[g_cur >> 6].gen & 1; }
int wave_lanes() { return g_waves[g_cur >> 6].lanes; }
void wave_sync() {
    Wave& w = g_waves[g_cur >> 6];
    int gen = w.gen;
    ++g_progress;
    if (++w.arrived == w.lanes) {
        w.arrived = 0;
        ++w.gen;
        return;
    }
    while (w.gen == gen) yield();
}

void launch(const std::function<void()>& body, dim3 grid, dim3 block) {
    int nt = block.x * block.y * block.z;
    if (nt > kMaxThreads || nt <= 0) { fprintf(stderr, "emu: bad block size %d\n", nt); abort(); }
    g_body = &body;
    gridDim = grid;
    blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = dim3(bx, by, bz);
                g_nthreads = g_alive = nt;
                g_bar_arrived = 0;
                for (int t = 0; t < nt; ++t) {
                    prepare(g_fib[t]);
                    g_fib[t].tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                }
                for (int w = 0; w < (nt + 63) / 64; ++w) {
                    g_waves[w].arrived = 0;
                    g_waves[w].gen = 0;
                    g_waves[w].lanes = (w * 64 + 64 <= nt) ? 64 : nt - w * 64;
                }
                while (g_alive > 0) {
                    long before = g_progress;
                    for (int t = 0; t < nt; ++t) {
                        if (g_fib[t].done) continue;
                        g_cur = t;
                        threadIdx = g_fib[t].tid;
                        emu_ctx_switch(&g_sched_sp, g_fib[t].sp);
                    }
                    if (g_progress == before) {
                        fprintf(stderr, "emu: deadlock in block (%u,%u,%u): %d threads alive, none progressing "
                                        "(divergent barrier / collective, or a lane exited early)\n", bx, by, bz, g_alive);
                        abort();
                    }
                }
            }
    g_body = nullptr;
}

}  // namespace emu
