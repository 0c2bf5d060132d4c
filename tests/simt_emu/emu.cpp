// SIMT emulator runtime (fibers + wave rendezvous).  TEST INFRASTRUCTURE ONLY -- see ntts/dev.h here.
#include <ntts/dev.h>

#include <pthread.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

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

// The workgroups of a launch are independent (as on the GPU): they are dealt out to a small pool of OS threads, each with its own fibers, wave
// rendezvous state, built-in index variables and "LDS" (NTTS_SHARED arrays are thread_local here).  Inside a workgroup nothing changes: one OS
// thread switches between the fibers of its block.  NTTS_EMU_THREADS (default: the machine's cores, at most 8; 1 = everything on the caller).
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
struct Ctx {                       // one per OS thread that runs workgroups
    Fiber fib[kMaxThreads];
    std::vector<Wave> waves = std::vector<Wave>(kMaxThreads / 64);
    void* sched_sp = nullptr;
    int cur = -1, nthreads = 0, alive = 0;
    int bar_arrived = 0, bar_gen = 0;
    long progress = 0;
};
static thread_local Ctx* g = nullptr;
static const std::function<void()>* g_body = nullptr;      // the launch in progress (read-only while workers run)
static dim3 g_grid, g_block;
static std::atomic<long> g_next{0};
static long g_total = 0;

static void yield() { emu_ctx_switch(&g->fib[g->cur].sp, g->sched_sp); }

static void fiber_main() {
    (*g_body)();
    g->fib[g->cur].done = true;
    --g->alive;
    ++g->progress;
    if (g->bar_arrived > 0 && g->bar_arrived == g->alive) {   // a workgroup barrier counts the threads still alive (as s_barrier does
        g->bar_arrived = 0;                                   // with terminated waves): the last exit may be what completes it
        ++g->bar_gen;
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
    int gen = g->bar_gen;
    ++g->progress;
    if (++g->bar_arrived == g->alive) {
        g->bar_arrived = 0;
        ++g->bar_gen;
        return;
    }
    while (g->bar_gen == gen) yield();
}

WaveSlots& wave_slots() { return g->waves[g->cur >> 6].slots; }
int wave_parity() { return g->waves[g->cur >> 6].gen & 1; }
int wave_lanes() { return g->waves[g->cur >> 6].lanes; }
void wave_sync() {
    Wave& w = g->waves[g->cur >> 6];
    int gen = w.gen;
    ++g->progress;
    if (++w.arrived == w.lanes) {
        w.arrived = 0;
        ++w.gen;
        return;
    }
    while (w.gen == gen) yield();
}

// one workgroup, start to finish, on the calling OS thread
static void run_block(unsigned bx, unsigned by, unsigned bz) {
    const dim3 block = g_block;
    const int nt = block.x * block.y * block.z;
    blockIdx = dim3(bx, by, bz);
    g->nthreads = g->alive = nt;
    g->bar_arrived = 0;
    for (int t = 0; t < nt; ++t) {
        prepare(g->fib[t]);
        g->fib[t].tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    }
    for (int w = 0; w < (nt + 63) / 64; ++w) {
        g->waves[w].arrived = 0;
        g->waves[w].gen = 0;
        g->waves[w].lanes = (w * 64 + 64 <= nt) ? 64 : nt - w * 64;
    }
    while (g->alive > 0) {
        long before = g->progress;
        for (int t = 0; t < nt; ++t) {
            if (g->fib[t].done) continue;
            g->cur = t;
            threadIdx = g->fib[t].tid;
            emu_ctx_switch(&g->sched_sp, g->fib[t].sp);
        }
        if (g->progress == before) {
            fprintf(stderr, "emu: deadlock in block (%u,%u,%u): %d threads alive, none progressing "
                            "(divergent barrier / collective, or a lane exited early)\n", bx, by, bz, g->alive);
            abort();
        }
    }
}

// workgroups of the current launch until none is left (the caller and every pool thread run this)
static void run_blocks() {
    if (!g) g = new Ctx();
    gridDim = g_grid;
    blockDim = g_block;
    for (;;) {
        const long i = g_next.fetch_add(1, std::memory_order_relaxed);
        if (i >= g_total) break;
        const unsigned bx = (unsigned)(i % g_grid.x), by = (unsigned)((i / g_grid.x) % g_grid.y), bz = (unsigned)(i / ((long)g_grid.x * g_grid.y));
        run_block(bx, by, bz);
    }
}

// ---- the pool: created at the first launch that has more than one workgroup; a forked child starts without one
// (never destroyed: the detached workers wait on them until the process ends, and destroying a condition variable with waiters blocks)
static std::mutex& g_mu = *new std::mutex;
static std::condition_variable& g_cv_work = *new std::condition_variable;
static std::condition_variable& g_cv_done = *new std::condition_variable;
static int g_pool_size = -1;          // -1: not created yet
static long g_job = 0;                // generation of the launch the workers should run
static int g_busy = 0;                // workers that have not finished the current generation

static void worker_main() {
    long seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(g_mu);
            g_cv_work.wait(lk, [&] { return g_job != seen; });
            seen = g_job;
        }
        run_blocks();
        {
            std::lock_guard<std::mutex> lk(g_mu);
            if (--g_busy == 0) g_cv_done.notify_one();
        }
    }
}
static void forget_pool_in_child() {     // the worker threads do not exist in a forked child: it builds its own pool at its first launch
    g_pool_size = -1;
    g_job = 0;
    g_busy = 0;
    new (&g_mu) std::mutex();                       // (whatever state the parent's were in at the fork)
    new (&g_cv_work) std::condition_variable();
    new (&g_cv_done) std::condition_variable();
}
static void make_pool() {
    int n = (int)std::thread::hardware_concurrency();
    if (n > 8) n = 8;
    if (const char* e = getenv("NTTS_EMU_THREADS")) n = atoi(e);
    if (n < 1) n = 1;
    g_pool_size = n - 1;                 // the launching thread is one of the n
    static bool hooked = false;
    if (!hooked) { pthread_atfork(nullptr, nullptr, forget_pool_in_child); hooked = true; }
    for (int i = 0; i < g_pool_size; ++i) std::thread(worker_main).detach();
}

void launch(const std::function<void()>& body, dim3 grid, dim3 block) {
    int nt = block.x * block.y * block.z;
    if (nt > kMaxThreads || nt <= 0) { fprintf(stderr, "emu: bad block size %d\n", nt); abort(); }
    g_body = &body;
    g_grid = grid;
    g_block = block;
    g_total = (long)grid.x * grid.y * grid.z;
    g_next.store(0, std::memory_order_relaxed);
    if (g_total > 1 && g_pool_size < 0) make_pool();
    if (g_total > 1 && g_pool_size > 0) {
        {
            std::lock_guard<std::mutex> lk(g_mu);
            ++g_job;
            g_busy = g_pool_size;
        }
        g_cv_work.notify_all();
        run_blocks();
        std::unique_lock<std::mutex> lk(g_mu);
        g_cv_done.wait(lk, [&] { return g_busy == 0; });
    } else {
        run_blocks();
    }
    g_body = nullptr;
}

}  // namespace emu
