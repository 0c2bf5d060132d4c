// backbone.cpp -- the backbone engine behind include/neutts_hip.h (compiled as HIP for gfx950).
//
// Host side of the path that replaces  backbone.generate(...)  (ref:neutts/neutts.py:338-347):
// packed weight arena, paged KV pool + page allocator, decode slots (continuous batching), the
// prefill pass and the hipGraph-captured decode step.  All arithmetic is in csrc/kernels/*.h.
#include <ntts/dev.h>

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <set>
#include <string>
#include <vector>

#include "../../include/neutts_hip.h"
#include "kernels/attn_decode.h"
#include "kernels/attn_prefill.h"
#include "kernels/gemm.h"
#include "kernels/gemv.h"
#include "kernels/norm.h"
#include "kernels/qkv_rope.h"
#include "kernels/sample.h"

using namespace ntts;

static std::string g_create_err;

struct LayerW {
    bf16_t *ln1, *wqkv, *bqkv, *wo, *ln2, *wgu, *wd;
    bf16_t *qn = nullptr, *kn = nullptr;   // qk_norm models (Qwen3-style): self_attn.q_norm / k_norm weights [head_dim]
    // fp8 model (NTTS_W_FP8_E4M3): the four matrices hold e4m3 bytes; per-output-channel weight scales (device fp32, in the
    // arena) and the static input scales of the four GEMMs (device copy in the arena for the broadcast, host copy for launches)
    float *sqkv = nullptr, *so = nullptr, *sgu = nullptr, *sd = nullptr, *xs_dev = nullptr;
    float xs[4] = {1.f, 1.f, 1.f, 1.f};   // input scale of: 0 qkv, 1 o_proj, 2 gate/up, 3 down
};

struct HostSlot {
    bool sampling = false;      // do_sample=1 request (needs the bf16 logits row)
    int state = SLOT_FREE;      // host view: FREE / RUNNING (may already be finished on device)
    int prompt_len = 0, max_len = 0;
    int pos_upper = 0;          // upper bound of the device-side pos
    std::vector<int> pages;     // KV pages of positions [32k, 32k+32); the leading ones may be shared (page_ref > 1)
    std::vector<int> prompt;    // prompt ids (what a later prompt must match to share this slot's prefix pages)
    unsigned gen = 0;           // bumped whenever the slot changes hands (release, prefill): a snapshot entry only speaks for the occupant it saw
};

// What the engines that read ONE weight arena share on the host (ntts_backbone_share_arena): the reference count that frees the arena
// with its last reader -- atomic: twins may be destroyed from different threads -- and the time of every engine's most recent
// decode call, from which each engine tells how many decode chains run side by side right now (its decode shape follows: apply_gang_shape).
struct ArenaShare {
    static constexpr int kMaxEngines = 16;
    std::atomic<int> refs{1};
    std::atomic<int> next_idx{1};                              // the donor is engine 0
    std::atomic<long long> last_decode_ns[kMaxEngines];
    ArenaShare() { for (auto& t : last_decode_ns) t.store(0); }
};
static long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct ntts_backbone {
    ntts_backbone_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;      // the stream every launch goes to: the engine's own, or one the caller lent (ntts_backbone_set_stream)
    hipStream_t own_stream = nullptr;
    std::string err;
    int H = 0, F = 0, NQKV = 0, max_pages = 0;
    // Attention geometry (round 6): head_dim 64 (NeuTTS-Air; the fused / tiered kernels) or 128, optional per-head q / k RMSNorm -- what the
    // reference's AutoModelForCausalLM dispatch (ref:neutts/neutts.py:164) may hand over for another backbone (Qwen3-style).  `generic` = any of the
    // two differs from NeuTTS-Air: the QKV projection is then a plain GEMM followed by rope_norm_kv_write_kernel (attn_prefill.h), attention runs
    // on the head_dim-templated two-sweep / decode kernels, and the small-batch GEMV step, the context-split attention, the resident / deep prompt
    // tiers and fp8 are off (correct first; none of it is on the benchmark's path).
    int HD = 64;
    bool qk_norm = false, generic = false;
    // PARKING (round 6; ntts_backbone_config::park_slots): the LAST park_slots of the cfg.max_batch slots never decode -- the decode step's launches cover
    // rows [0, dec_rows) only.  A prompt pass may fill a parked slot like any other (its KV pages, its first token); ntts_backbone_activate moves it into
    // a free decode slot later.  What it is for: a continuous-batching scheduler admits prompts in efficient waves of 24-32 INTO the parking rows while
    // every decode row stays busy, instead of letting freed decode rows idle until the next wave (slot occupancy 0.92 -> ~0.99 of the decode rows).
    int dec_rows = 0;

    // weights
    bf16_t* arena = nullptr;
    size_t arena_elems = 0;
    bool arena_shared = false;         // the arena is another engine's (ntts_backbone_share_arena)
    ArenaShare* share = nullptr;       // host state shared by the engines that read `arena` (a donor and its readers): the LAST one to be destroyed frees
    int share_idx = 0;                 // the arena, in whatever order -- and from whatever threads -- the caller destroys them; this engine's index in it
    bf16_t *embed = nullptr, *final_norm = nullptr, *rope_cos = nullptr, *rope_sin = nullptr;
    bf16_t* embed_tm = nullptr;   // the lm_head's weight stream: tile-major copy of the tied embedding, the untied
                                  // "lm_head.weight", or (fp8) the e4m3 bytes of either
    bool fp8 = false, tied = true, has_bias = true;
    float* shead = nullptr;       // fp8: per-vocab-row scales of the head matrix
    float* xs_head_dev = nullptr;
    float xs_head = 1.f;          // fp8: input scale of the lm_head
    bool head_from_embed = false, head_loaded = false;
    std::set<std::string> needed; // tensor names finalize() insists on
    std::vector<LayerW> layers;
    std::set<std::string> loaded;
    std::set<std::string> quantised_here;   // fp8 model: matrices that came in bf16 / fp32 and were quantised on upload (no weight_scale may follow)
    float inv_freq[64];
    bool have_inv_freq = false, finalized = false;
    int* gu_map_gate = nullptr;  // device row maps for the gate/up packing
    int* gu_map_up = nullptr;

    // KV pool
    bf16_t* kv = nullptr;
    size_t layer_stride = 0, kv_half = 0;  // elements
    int num_pages = 0;
    std::vector<int> free_pages;
    std::vector<int> page_ref;   // owners per page: 1 private, > 1 a prefix shared by several slots (read-only by construction)
    std::vector<HostSlot> slots;

    // device slot state
    SlotArrays sl{};
    int* block_table = nullptr;
    int* ibuf = nullptr;  // backing store of the int arrays

    // decode workspaces
    bf16_t *h_dec = nullptr, *xn_dec = nullptr, *qkv_dec = nullptr, *attn_dec = nullptr, *act_dec = nullptr;
    float* slabs = nullptr;
    float* part_val = nullptr;
    int* part_idx = nullptr;
    int n_part = 0;
    float* logits = nullptr;  // debug
    bf16_t* logits_bf16 = nullptr;   // [B][ldl] processed logits for the top-k sampler (allocated on first sampling request)
    long ldl = 0;
    int n_sampling = 0;              // running slots with do_sample=1
    bool graph_has_logits = false;
    // ---- decode-step shape, fixed at create() from the batch size (every constant below was swept on MI355X; the losing variants and
    //      their knobs are gone -- DESIGN.md section 4 keeps the numbers, the git history the code)
    int ks_o = 1, ks_d = 1;          // split-K of o_proj / down_proj (fp32 slabs reduced by the norm kernel behind them)
    // "Tall" decode tiles (NTTS_TALL; round 5): ALL rows of a <= 256-row chain in one m-block, so that a weight tile goes through a
    // CU's load path once per chain instead of once per 64-row (128-row) m-block.  Alone on the chip such a GEMM has a quarter of the
    // workgroups and loses; in an engine gang (four chains side by side) the other chains fill the CUs and what
    // counts is the bytes every CU pulls.  bit 0: o_proj, bit 1: down_proj on the 256 x 64 / 8-wave tile; gu_tile picks the gate/up tile.
    int tall = 0;
    bool tall_xcd_split = true;      // the 256-row split-K tiles with ONE K slice per XCD (pair) (gemm.h xcd_nsplit; NTTS_TALL_XCD_SPLIT=0: slices in gridDim.y)
    bool qkv_wstat = false;          // QKV: column blocks dealt to XCDs (qkv_rope.h "W-stationary"; NTTS_QKV_WSTAT) -- used when the row-block placement is off
    // Opt-in restricted lm_head (ntts_backbone_set_logits_range; SURVEY 7 "hard parts": the reference only ever consumes <|speech_N|> ids
    // and the EOS, ref:neutts/neutts.py:276,336-341): a COMPACTED copy of the head -- rows [lr_lo, lr_hi) followed by the EOS row, padded to
    // whole 64-row groups -- is what the lm_head streams (NeuTTS-Air: 118 MB instead of 390 MB); columns are mapped back to token ids
    // in the sample kernel.  NOT the reference's arithmetic when the full-vocabulary argmax lies outside the range: never the default.
    int lr_lo = -1, lr_hi = -1, lr_eos = -1, lr_rows = 0;   // lr_rows = lr_hi - lr_lo + 1 valid columns (0 = full head)
    bf16_t* head_r = nullptr;        // compacted head matrix (tile-major; e4m3 bytes in the fp8 model)
    float* shead_r = nullptr;        // fp8: its per-row scales
    int n_part_full = 0;
    float* calib = nullptr;          // [num_layers * 4 + 1] running max |x| of every GEMM input seen by the prompt passes (ntts_backbone_calibrate)
    int gang = 1;                    // decode chains side by side on the GPU, this one included: the defaults of tall / xcd_affine follow it.  ABI 9: counted by
                                     // the engine itself at every decode call (engines of the same arena that decoded within kGangWindowNs) unless
    int gang_forced = 0;             // ntts_backbone_set_gang pinned it (> 0; tests, sweeps, profiling passes of one engine in the gang's shape)
    static constexpr long long kGangWindowNs = 50LL * 1000 * 1000;
    hipGraphExec_t graph_shape[2] = {nullptr, nullptr};   // the captured step of the shape not in use (index = side-by-side shape?), kept across switches
    bool graph_tried_shape[2] = {false, false};
    int shape = 0;                   // 0 = single-chain shape, 1 = side-by-side shape (what `graph` / `graph_tried` currently belong to)
    int tall_env = -1, affine_env = -1;   // NTTS_TALL / NTTS_XCD_AFFINE when set (sweeps, tests): they win over the gang's defaults
    // WIDE decode step (round 6; max_batch >= 512, e.g. the four 256-utterance batches of the static benchmark as ONE 1024-row chain instead of
    // four 256-row chains side by side): every GEMM of the step sees M = 1024 -- a weight tile enters a CU's LDS once per 64-256 rows, not once per
    // chain -- at the price of one chain's launch boundaries.  Tiles per GEMM (NTTS_WIDE_* override; DESIGN.md section 4l has the sweep):
    //   QKV + RoPE: 32 * wide_qkv batch rows per workgroup (qkv_rope.h TMQ);  o_proj: 1 = 64 x 64 tiles, whole K, residual add in the epilogue (no
    //   fp32 slabs; the norm behind it reads one bf16 stream), 0 = the split-K tiles of the narrow step;  down_proj: 1 = 128 x 128 / 8 waves split-K
    bool wide = false;
    int wide_qkv = 2, wide_o = 1, wide_down = 1;
    int gu_tile = 0;                 // gate/up tile (NTTS_GU_TILE): 0 = 128 x 128 / 8 waves / 3 slots (above batch 128; 64 x 64 below), 1 = 256 x 192 / 12 waves / 2 slots,
                                     // 2 = 256 x 256 / 16 waves / 2 slots, 3 = 128 x 128 / 2 slots (two workgroups per CU)
    // Tile path: the QKV projection with bias, rounding, RoPE and the K append in its epilogue (qkv_rope.h); the attention kernel
    // then has no prologue.  step_meta / rope_rows: the per-step row records that kernel reads (step_meta_kernel, once per step).
    int* step_meta = nullptr;
    bf16_t* rope_rows = nullptr;
    int head_tile = 0;      // lm_head tile (NTTS_HEAD_TILE): 0 = 64 x 64 skinny (batch <= 64), 1 = 128 x 128, 2 = 256 x 256 (16 waves; the fp8 model
                            // above batch 128), 4 = 256 x 288 natural-order tile, 12 waves (bf16 above batch 128: 756 tiles = 2.95 rounds of the 256 CUs
                            // instead of 850 = 3.32: 129.5 -> 118.7-123.8 us, profiles/r02k_sweep_lpt_head_gu_tiles.log).  Its weight stream uses the
                            // non-temporal policy (136.6 -> 132.3 us, profiles/r02a_sweep_nt_graphsteps.jsonl; slower on the skinny layer GEMMs)
    bool gu_128 = false;    // gate/up on the 128 x 128 / 8-wave tile (above batch 128: -2 % per step) instead of the 64 x 64 skinny tile
    int xcd_affine = 0;        // row-block placement per XCD group (gemm.h xcd_maffine; NTTS_XCD_AFFINE): bit 0: o_proj + the norm behind it, bit 1: down_proj + the
                               // norm behind it, bit 2: QKV GEMM + attention.  Batch 256 (profiles/r02k_sweep_xcd_affine*.log): 7 -> step 1.630 -> 1.615 ms
    int xcd_xps = 0;           // XCDs per 64-row m-block of the decode batch (8 / (max_batch / 64)); 0 = the batch does not split that way
    int xl_min_m = 0;          // > 0 (NTTS_XL_MIN_M, tests): rows from which the big-M GEMMs take the 256-row tiles; 0 = by tile count (gemm_large)
    // Split-K decode GEMMs below batch 129: XCD-aware slice placement of o_proj (gemm.h GemmArgs::xcd_nsplit): FETCH per skinny-GEMM launch
    // 15.0 -> 6.8 MB (profiles/r02f_*); above it the row-block placement (xcd_affine) takes over
    // Small-batch decode step (max_batch <= NTTS_SMALL_BATCH, default 8; BASELINE configs[1] = batch 1): wave-per-16-features
    // GEMV kernels with the slab-reduce + residual + RMSNorm fused into the consumer's prologue (gemv.h) -- 5 launches per
    // layer instead of 7.  Split-K factors 7 / 10 (o_proj / down_proj slabs, reduced in the next GEMV's prologue); QKV splits K inside its workgroups.
    bool small = false;
    static constexpr int kSksO = 7, kSksD = 10;
    bf16_t* h_alt = nullptr;     // second residual-stream buffer (the fused prologue writes the new stream while others still read the old)
    int attn_split = 0;          // context-split attention over this many workgroups per (sequence, kv-head) (NTTS_ATTN_SPLIT; 0 = off, the default
    int attn_split_ctx = 896;    //   since round 3), used once the longest running context reaches attn_split_ctx tokens (NTTS_ATTN_SPLIT_CTX).  Round 2, batch 1
                                 //   (profiles/r02i_sweep_split_*): context 1850: 1.257 -> 1.093 ms, break-even ~870 -- against an attention kernel that spent 3 us in
                                 //   a prologue and pulled a whole context through one CU.  Against the prologue-free kernel with the output dimensions split over
                                 //   workgroups (attn_decode.h DS) the two-launch form LOSES at every batch and context it was used for (round 3, one box, split 8 vs
                                 //   off, ms per step): batch 1 0.961 / 0.980 / 0.994 vs 0.871 / 0.917 / 0.972 at context 1010 / 1410 / 1810; batch 16 1.322 vs 1.184,
                                 //   batch 64 1.526 vs 1.287, batch 128 1.833 vs 1.482 at context 1010 (1.383 / 1.637 / 1.998 vs 1.252 / 1.374 / 1.684 at 1610).  The
                                 //   path stays available (and tested) behind the knob.
    bf16_t* as_scores = nullptr;  // [B * nkv][8][max_context + 16]
    float* as_stats = nullptr;    // [B * nkv][attn_split][8][2]
    float* as_oslabs = nullptr;   // [attn_split][B][nh * 64]
    float* slabs2 = nullptr;     // down_proj's slabs (read by the next layer's QKV prologue while that kernel writes `slabs`)
    int n_cu = 256;

    // prefill workspaces
    int Tmax = 0;
    bf16_t *h_pf = nullptr, *xn_pf = nullptr, *qkv_pf = nullptr, *attn_pf = nullptr, *o_pf = nullptr, *act_pf = nullptr;
    int* meta_dev = nullptr;
    size_t meta_cap = 0;
    int pf_res_cap = kPfResPages * kPage;   // prompt-pass attention: queries below this position take the resident kernel (attn_prefill.h); NTTS_PF_RES_CAP
    int pf_deep_cap = kPfDeepPages * kPage; // ... and from there up to this one the deep kernel; NTTS_PF_DEEP_CAP (<= pf_res_cap: tier off)

    // page-locked staging ring of the meta block: a host-to-device copy from pageable memory forced a stream synchronisation into
    // every prompt pass / decode call / code export (the host sat out the previous prompt pass before it could enqueue the next)
    static constexpr int kMetaStages = 4;
    int* meta_host[kMetaStages] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t meta_ev[kMetaStages] = {nullptr, nullptr, nullptr, nullptr};
    // asynchronous slot snapshot (ntts_backbone_poll_begin / _end): state | n_new of every slot as of a point of the stream, in
    // page-locked memory; the copy stream carries the reads of finished slots' ids past the decode steps still queued
    int* snap_host = nullptr;          // [2 * max_batch]
    std::vector<unsigned> snap_gen;    // HostSlot::gen of every slot when the open / last completed snapshot was enqueued
    hipEvent_t snap_ev = nullptr;
    bool snap_open = false, snap_valid = false;
    hipStream_t copy_stream = nullptr;
    bool meta_used[kMetaStages] = {false, false, false, false};
    int meta_next = 0;
    // ... and a deeper ring of SMALL slots for the uploads a scheduler makes per finished request / per step (slot lists of code exports, activations,
    // block-table updates): with four slots the fifth export of a poll waited for the engine's stream to reach the first one -- i.e. for the decode
    // step in flight -- and the launching thread sat out a whole gang round inside a hook while the other engines' queues ran dry
    // (bench.py --mode continuous on four 512-slot engines: 4.8 of 24 s in the on_finished hooks, profiles/r06j_*)
    static constexpr int kSmallStages = 32;
    size_t small_cap = 0;
    int* small_host = nullptr;                 // kSmallStages x small_cap ints, one page-locked block
    hipEvent_t small_ev[kSmallStages] = {};
    bool small_used[kSmallStages] = {};
    int small_next = 0;

    hipGraphExec_t graph = nullptr;
    hipGraphExec_t graph_split = nullptr;   // the small-batch step with context-split attention (long contexts)
    bool graph_split_tried = false, split_active = false;
    bool graph_tried = false, use_graph = true;
    hipEvent_t ev[4]{};
    // optional side stream for the prompt pass, restricted to a subset of the CUs (ntts_backbone_set_prefill_cu_mask): a prefill
    // on it leaves the other CUs to whatever else runs on the GPU -- another engine's decode steps, which are launch- and
    // latency-bound and lose little on fewer CUs, while an unrestricted prefill's 1024-thread workgroups would take every CU
    hipStream_t pf_stream = nullptr;
    bool pf_lent = false;              // pf_stream is the caller's (ntts_backbone_set_prefill_stream): not destroyed here
    hipEvent_t pf_ev[2]{};
    bool have_pf_time = false, have_dec_time = false;
    unsigned long long* attn_tl = nullptr;   // diagnostics (ntts_backbone_attn_timeline)
    unsigned long long* gemv_tl = nullptr;   // diagnostics (ntts_backbone_gemv_timeline)
    long long pf_tokens_computed = 0, pf_tokens_shared = 0;   // prompt tokens pushed through the layers / served from shared pages
};

static int fail(ntts_backbone* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf; else g_create_err = buf;
    return code;
}
#define HIPCHK(e, call)                                                                              \
    do {                                                                                             \
        hipError_t _s = (call);                                                                      \
        if (_s != hipSuccess) return fail(e, NTTS_EHIP, "%s failed: %s", #call, hipGetErrorString(_s)); \
    } while (0)

extern "C" int ntts_abi_version(void) { return NTTS_ABI_VERSION; }
extern "C" const char* ntts_last_error(const ntts_backbone* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// scratch device allocation released on every exit path of its scope (the HIPCHK early returns included)
struct DevScratch {
    void* p = nullptr;
    ~DevScratch() { if (p) (void)hipFree(p); }
};

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// (max, first index) pairs the lm_head leaves per row for a head of N columns (gemm.h / gemv.h EPI_ARGMAX: one per wave tile)
static int n_part_for(const ntts_backbone* e, int N) {
    return e->small ? (N + 15) / 16 : e->head_tile == 0 ? (N + 63) / 64 : e->head_tile == 4 ? ((N + 287) / 288) * 3 : e->head_tile == 2 ? ((N + 255) / 256) * 4 : ((N + 127) / 128) * 2;
}

// The decode step's shape as a function of how many chains share the chip (ntts_backbone_set_gang; measured on MI355X with
// tools/sweep_gang.py, profiles/r05a_sweep_gang_*: four 256-row chains, ms per 256-row step -- single-chain shape 0.988, XCD placement
// off 0.970, + o_proj / down_proj on the 256-row tile 0.953; the same tiles ALONE cost the single chain 1.64 -> 1.88 ms).
static void apply_gang_shape(ntts_backbone* e) {
    const int B = e->dec_rows;
    const bool side_by_side = e->gang >= 2 && B > 128 && B <= 256;
    e->tall = (e->tall_env >= 0 ? e->tall_env : (side_by_side ? 3 : 0)) & 3;
    e->xcd_affine = e->affine_env >= 0 ? e->affine_env : (B > 128 && !side_by_side ? 7 : 0);
    // QKV column blocks dealt to XCDs whenever the row-block placement is off for a gang (FETCH per launch 17.1 -> 6.1 MB, step -0.7 %:
    // profiles/r05g_*); NTTS_QKV_WSTAT = 0 / 1 overrides
    const int ws = env_int("NTTS_QKV_WSTAT", -1);
    e->qkv_wstat = ws >= 0 ? ws != 0 : side_by_side;
    e->tall_xcd_split = env_int("NTTS_TALL_XCD_SPLIT", 1) != 0;
}

extern "C" int ntts_backbone_create(const ntts_backbone_config* c, int device, ntts_backbone** out) {
    if (!c || !out) return fail(nullptr, NTTS_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device || device < 0)
        return fail(nullptr, NTTS_ENODEV, "no HIP device %d (found %d): this library has no CPU fallback", device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, NTTS_ENODEV, "device %d is '%s', kernels are built for gfx950 only", device, prop.gcnArchName);
    if (c->head_dim != 64 && c->head_dim != 128) return fail(nullptr, NTTS_EINVAL, "head_dim %d unsupported (64 or 128)", c->head_dim);
    if (c->num_heads % c->num_kv_heads || c->num_heads / c->num_kv_heads > kGroupMax)
        return fail(nullptr, NTTS_EINVAL, "GQA group %d/%d unsupported (<= %d)", c->num_heads, c->num_kv_heads, kGroupMax);
    if (c->hidden_size % 64 || c->intermediate_size % 64 || c->hidden_size > 2048)
        return fail(nullptr, NTTS_EINVAL, "hidden/intermediate size must be multiples of 64 (hidden <= 2048)");
    if (c->max_context > kAttnLMax || c->max_context % kPage) return fail(nullptr, NTTS_EINVAL, "max_context must be <= %d and a multiple of %d", kAttnLMax, kPage);
    if (c->max_batch < 1 || c->vocab_size < 2) return fail(nullptr, NTTS_EINVAL, "bad max_batch / vocab_size");
    if (c->park_slots < 0 || c->park_slots >= c->max_batch) return fail(nullptr, NTTS_EINVAL, "park_slots %d: need 0 <= park_slots < max_batch (%d)", c->park_slots, c->max_batch);
    if ((c->qk_norm || c->head_dim != 64) && c->weight_dtype != NTTS_W_BF16)
        return fail(nullptr, NTTS_EINVAL, "qk_norm / head_dim %d are bf16-only (the fp8 model variant covers the head_dim 64, no-qk-norm family)", c->head_dim);
    if (c->weight_dtype != NTTS_W_BF16 && c->weight_dtype != NTTS_W_FP8_E4M3) return fail(nullptr, NTTS_EINVAL, "unknown weight_dtype %d", c->weight_dtype);
    if (c->weight_dtype == NTTS_W_FP8_E4M3 && (c->hidden_size % 128 || c->intermediate_size % 128 || (c->num_heads * c->head_dim) % 128))
        return fail(nullptr, NTTS_EINVAL, "fp8 weights need hidden / intermediate / q width to be multiples of 128 (one 128-byte K tile)");

    ntts_backbone* e = new ntts_backbone();
    e->cfg = *c;
    e->device = device;
    e->H = c->hidden_size;
    e->F = c->intermediate_size;
    e->HD = c->head_dim;
    e->qk_norm = c->qk_norm != 0;
    e->generic = e->qk_norm || e->HD != 64;
    e->NQKV = (c->num_heads + 2 * c->num_kv_heads) * e->HD;
    e->max_pages = c->max_context / kPage;
    e->Tmax = c->max_prefill_tokens > 0 ? c->max_prefill_tokens : 16384;
    e->num_pages = c->num_pages > 0 ? c->num_pages : c->max_batch * e->max_pages;
    e->dec_rows = c->max_batch - c->park_slots;
    e->use_graph = env_int("NTTS_NO_GRAPH", 0) == 0;
    e->fp8 = c->weight_dtype == NTTS_W_FP8_E4M3;
    e->tied = c->tie_word_embeddings != 0;
    e->has_bias = c->attention_bias != 0;
    const int B = c->max_batch, H = e->H, F = e->F, L = c->num_layers, V = c->vocab_size;

#define CR_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t _s = (call);                                                               \
        if (_s != hipSuccess) {                                                               \
            int rc = fail(nullptr, _s == 2 ? NTTS_ENOMEM : NTTS_EHIP, "%s failed: %s", #call, hipGetErrorString(_s)); \
            ntts_backbone_destroy(e);                                                         \
            return rc;                                                                        \
        }                                                                                     \
    } while (0)

    CR_HIP(hipSetDevice(device));
    CR_HIP(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    e->stream = e->own_stream;
    for (auto& ev : e->ev) CR_HIP(hipEventCreate(&ev));

    // ---- weight arena (bf16), every tensor 256-byte aligned
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += align_up(n, 128); return o; };
    const size_t o_embed = take((size_t)V * H);
    // Weight layout: tile-major (gemm.h GemmArgs::w_tile_major: each workgroup's weight stream is one sequential run of HBM
    // addresses: lm_head -8 %, gate/up -3 % on the micro-benchmark, profiles/r01e_ubench_weight_layout.txt; neutral end to end,
    // r01g_ab_weight_layout.jsonl).  The embedding gather needs rows, the lm_head tiles: the tied matrix is kept in both layouts.
    // sizes in bf16 elements: a matrix of n weights takes n (bf16) or n / 2 (fp8 bytes); fp32 arrays take 2 per value
    auto take_w = [&](size_t n) { return take(e->fp8 ? (n + 1) / 2 : n); };
    auto take_f = [&](size_t n) { return take(2 * n); };
    const size_t o_embed_tm = take_w((size_t)((V + 63) / 64) * 64 * H);   // the head's own copy (layout / values / precision differ)
    const size_t o_shead = e->fp8 ? take_f((size_t)((V + 63) / 64) * 64) : 0, o_xs_head = e->fp8 ? take_f(4) : 0;
    struct LO { size_t ln1, wqkv, bqkv, wo, ln2, wgu, wd, sqkv, so, sgu, sd, xs, qn, kn; };
    std::vector<LO> lo(L);
    for (int i = 0; i < L; ++i) {
        lo[i].ln1 = take(H); lo[i].wqkv = take_w((size_t)e->NQKV * H); lo[i].bqkv = take(e->NQKV);
        lo[i].wo = take_w((size_t)H * c->num_heads * e->HD); lo[i].ln2 = take(H);
        if (e->qk_norm) { lo[i].qn = take(e->HD); lo[i].kn = take(e->HD); }
        lo[i].wgu = take_w((size_t)2 * F * H); lo[i].wd = take_w((size_t)H * F);
        if (e->fp8) { lo[i].sqkv = take_f(e->NQKV); lo[i].so = take_f(H); lo[i].sgu = take_f(2 * (size_t)F); lo[i].sd = take_f(H); lo[i].xs = take_f(4); }
    }
    const size_t o_fn = take(H), o_cos = take((size_t)c->max_context * (e->HD / 2)), o_sin = take((size_t)c->max_context * (e->HD / 2));
    e->arena_elems = off;
    CR_HIP(hipMalloc((void**)&e->arena, off * sizeof(bf16_t)));
    CR_HIP(hipMemset(e->arena, 0, off * sizeof(bf16_t)));
    e->share = new ArenaShare();
    e->embed = e->arena + o_embed;
    e->embed_tm = e->arena + o_embed_tm;
    if (e->fp8) { e->shead = (float*)(e->arena + o_shead); e->xs_head_dev = (float*)(e->arena + o_xs_head); }
    e->layers.resize(L);
    for (int i = 0; i < L; ++i) {
        LayerW& w = e->layers[i];
        w.ln1 = e->arena + lo[i].ln1; w.wqkv = e->arena + lo[i].wqkv; w.bqkv = e->arena + lo[i].bqkv; w.wo = e->arena + lo[i].wo;
        w.ln2 = e->arena + lo[i].ln2; w.wgu = e->arena + lo[i].wgu; w.wd = e->arena + lo[i].wd;
        if (e->qk_norm) { w.qn = e->arena + lo[i].qn; w.kn = e->arena + lo[i].kn; }
        if (e->fp8) {
            w.sqkv = (float*)(e->arena + lo[i].sqkv); w.so = (float*)(e->arena + lo[i].so); w.sgu = (float*)(e->arena + lo[i].sgu);
            w.sd = (float*)(e->arena + lo[i].sd); w.xs_dev = (float*)(e->arena + lo[i].xs);
        }
    }
    // what finalize() will insist on
    e->needed = {"model.embed_tokens.weight", "model.norm.weight", "rope.inv_freq"};
    if (!e->tied) e->needed.insert("lm_head.weight");
    if (e->fp8) e->needed.insert("lm_head.input_scale");
    for (int i = 0; i < L; ++i) {
        const std::string pre = "model.layers." + std::to_string(i) + ".";
        for (const char* t : {"input_layernorm.weight", "post_attention_layernorm.weight", "self_attn.q_proj.weight", "self_attn.k_proj.weight",
                              "self_attn.v_proj.weight", "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight"})
            e->needed.insert(pre + t);
        if (e->has_bias)
            for (const char* t : {"self_attn.q_proj.bias", "self_attn.k_proj.bias", "self_attn.v_proj.bias"}) e->needed.insert(pre + t);
        if (e->qk_norm)
            for (const char* t : {"self_attn.q_norm.weight", "self_attn.k_norm.weight"}) e->needed.insert(pre + t);
        if (e->fp8)
            for (const char* t : {"self_attn.q_proj.input_scale", "self_attn.o_proj.input_scale", "mlp.gate_proj.input_scale", "mlp.down_proj.input_scale"})
                e->needed.insert(pre + t);
    }
    e->final_norm = e->arena + o_fn;
    e->rope_cos = e->arena + o_cos;
    e->rope_sin = e->arena + o_sin;

    // gate/up packing maps: source feature f -> packed row (see gemm.h EPI_SILU_MUL)
    {
        std::vector<int> mg(F), mu(F);
        for (int f = 0; f < F; ++f) {
            const int b = f / 32, g = (f % 32) / 8, jj = (f % 8) / 4, r = f % 4;
            mg[f] = b * 64 + g * 16 + jj * 4 + r;
            mu[f] = b * 64 + g * 16 + (jj + 2) * 4 + r;
        }
        CR_HIP(hipMalloc((void**)&e->gu_map_gate, F * sizeof(int)));
        CR_HIP(hipMalloc((void**)&e->gu_map_up, F * sizeof(int)));
        CR_HIP(hipMemcpy(e->gu_map_gate, mg.data(), F * sizeof(int), hipMemcpyHostToDevice));
        CR_HIP(hipMemcpy(e->gu_map_up, mu.data(), F * sizeof(int), hipMemcpyHostToDevice));
    }

    // ---- KV pool: per layer [K pages | V^T pages], page-head = 32 tokens x 64 d
    e->kv_half = (size_t)e->num_pages * c->num_kv_heads * kPage * e->HD;
    e->layer_stride = 2 * e->kv_half;
    CR_HIP(hipMalloc((void**)&e->kv, (size_t)L * e->layer_stride * sizeof(bf16_t)));
    CR_HIP(hipMemset(e->kv, 0, (size_t)L * e->layer_stride * sizeof(bf16_t)));
    e->free_pages.reserve(e->num_pages);
    for (int p = e->num_pages - 1; p >= 0; --p) e->free_pages.push_back(p);
    e->page_ref.assign(e->num_pages, 0);
    e->slots.resize(B);

    // ---- slot state
    const size_t n_int = (size_t)B * 13 + (size_t)B * c->max_context + (size_t)B * e->max_pages;
    CR_HIP(hipMalloc((void**)&e->ibuf, n_int * sizeof(int)));
    CR_HIP(hipMemset(e->ibuf, 0, n_int * sizeof(int)));
    int* ip = e->ibuf;
    e->sl.state = ip; ip += B; e->sl.pos = ip; ip += B; e->sl.n_new = ip; ip += B; e->sl.cur_tok = ip; ip += B;
    e->sl.prompt_len = ip; ip += B; e->sl.min_new = ip; ip += B; e->sl.max_len = ip; ip += B; e->sl.eos = ip; ip += B;
    e->sl.mask_eos = ip; ip += B;
    e->sl.top_k = ip; ip += B;
    e->sl.temperature = (float*)ip; ip += B;
    e->sl.seed = (unsigned int*)ip; ip += 2 * B;
    e->sl.out_tokens = ip; ip += (size_t)B * c->max_context;
    e->sl.out_stride = c->max_context;
    e->block_table = ip;

    // ---- decode workspaces + tile choices (decode GEMMs are weight-streaming, M = max_batch)
    const int D = e->dec_rows;                 // rows of the decode step (the slots that are not parking rows): what the step's tiles are chosen for
    const int mblocks_s = (D + 63) / 64;
    auto pick_split = [&](int nblocks, int ktiles) {
        int blocks = nblocks * mblocks_s, ks = (224 + blocks - 1) / blocks;
        if (ks < 1) ks = 1;
        if (ks > 16) ks = 16;
        if (ks > ktiles) ks = ktiles;
        return ks;
    };
    const int ktile = e->fp8 ? 128 : 64;   // K extent of one 128-byte tile
    const int max_slabs = 16;
    e->ks_o = std::min(max_slabs, pick_split(H / 64, c->num_heads * e->HD / ktile));
    e->ks_d = std::min(max_slabs, pick_split(H / 64, F / ktile));
    if (int k = env_int("NTTS_KS_O", 0); k > 0) e->ks_o = std::min(std::min(max_slabs, k), c->num_heads * e->HD / ktile);
    if (int k = env_int("NTTS_KS_D", 0); k > 0) e->ks_d = std::min(std::min(max_slabs, k), F / ktile);
    e->tall_env = env_int("NTTS_TALL", -1);
    // (fp8 engines take the wide shape's gate/up tile, o_proj with the residual epilogue and the non-temporal K / V^T pages; their QKV and down_proj keep the
    //  narrow kernels: nano-fp8 at 4 x 512 slots 234.0 -> 241.1 k codec-tokens/s, fp8 parity tests unchanged -- round 6)
    e->wide = env_int("NTTS_WIDE", D >= 512 ? 1 : 0) != 0;
    e->wide_qkv = env_int("NTTS_WIDE_QKV", 2);
    if (e->wide_qkv != 1 && e->wide_qkv != 2 && e->wide_qkv != 4) e->wide_qkv = 2;
    e->wide_o = env_int("NTTS_WIDE_O", 1);
    e->wide_down = env_int("NTTS_WIDE_DOWN", 1);
    // (wide engines whose rows are a multiple of 128 but not of 256 -- 640: bench.py's engines for the driver's --steps 20 -- would leave half of the last
    //  256-row block empty: the 2-slot 128 x 128 tile instead, 3.02 -> 2.89 ms per step alone, 1.97 -> 1.95 in a gang of four; same bits, profiles/r06h_sweep_gang_640.txt)
    e->gu_tile = env_int("NTTS_GU_TILE", e->wide ? ((D % 256) == 128 ? 3 : 1) : 0);
    if (e->gu_tile < 0 || e->gu_tile > 3) e->gu_tile = 0;
    if (e->wide && e->wide_down == 1 && !e->fp8 && env_int("NTTS_KS_D", 0) <= 0) e->ks_d = 4;   // 8 x 7 tiles of 128 x 128 x 4 K slices = 224 workgroups
    e->xl_min_m = env_int("NTTS_XL_MIN_M", 0);
    {   // 0 = every query on the two-sweep kernel; whole pages, at most what the resident kernel holds
        int cap = env_int("NTTS_PF_RES_CAP", kPfResPages * kPage) / kPage * kPage;
        e->pf_res_cap = cap < 0 ? 0 : cap > kPfResPages * kPage ? kPfResPages * kPage : cap;
        int dcap = env_int("NTTS_PF_DEEP_CAP", kPfDeepPages * kPage) / kPage * kPage;
        dcap = dcap > kPfDeepPages * kPage ? kPfDeepPages * kPage : dcap;
        e->pf_deep_cap = dcap < e->pf_res_cap ? e->pf_res_cap : dcap;
        if (e->generic) e->pf_res_cap = e->pf_deep_cap = 0;     // every query on the (head_dim-templated) two-sweep kernel
    }
    e->xcd_xps = (D == 64 || D == 128 || D == 256 || D == 512) ? 8 / (D / 64) : 0;
    e->affine_env = env_int("NTTS_XCD_AFFINE", -1);
    apply_gang_shape(e);
    e->head_tile = env_int("NTTS_HEAD_TILE", D > 128 ? (e->fp8 || e->wide ? 2 : 4) : D > 64 ? 1 : 0);   // (1024 rows: 256 x 256 424 us, 256 x 288 438)
    if (e->head_tile != 0 && e->head_tile != 1 && e->head_tile != 2 && e->head_tile != 4) e->head_tile = 1;
    if (e->fp8 && e->head_tile == 4) e->head_tile = 2;          // (the natural-order tile is bf16 only)
    e->gu_128 = D > 128;
    e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    e->small = !e->generic && c->park_slots == 0 && B <= env_int("NTTS_SMALL_BATCH", 8) && B <= kGemvRows && H <= 1024 && V % 16 == 0 && e->NQKV % 16 == 0 && F % 32 == 0;   // (bf16 and fp8 models alike: gemv.h / qkv_rope.h F8 instantiations)
    static_assert(ntts_backbone::kSksO <= 16 && ntts_backbone::kSksD <= 16, "slab counts");
    {
        CR_HIP(hipMalloc((void**)&e->step_meta, (size_t)B * 4 * sizeof(int)));
        CR_HIP(hipMemset(e->step_meta, 0, (size_t)B * 4 * sizeof(int)));
        CR_HIP(hipMalloc((void**)&e->rope_rows, (size_t)B * e->HD * 2));
        CR_HIP(hipMemset(e->rope_rows, 0, (size_t)B * e->HD * 2));
    }
    e->n_part = e->n_part_full = n_part_for(e, V);
    CR_HIP(hipMalloc((void**)&e->h_dec, (size_t)B * H * 2));
    CR_HIP(hipMalloc((void**)&e->xn_dec, (size_t)B * H * 2));
    CR_HIP(hipMalloc((void**)&e->qkv_dec, (size_t)B * e->NQKV * 2));
    CR_HIP(hipMalloc((void**)&e->attn_dec, (size_t)B * c->num_heads * e->HD * 2));
    CR_HIP(hipMalloc((void**)&e->act_dec, (size_t)B * F * 2));
    CR_HIP(hipMalloc((void**)&e->slabs, (size_t)max_slabs * B * H * 4));
    if (e->small) {
        CR_HIP(hipMalloc((void**)&e->slabs2, (size_t)max_slabs * B * H * 4));
    }
    {
        // context-split attention (small-batch path; tile path below 2 workgroups per CU with split-K QKV slabs, bf16 engines)
        e->attn_split = env_int("NTTS_ATTN_SPLIT", 0);
        e->attn_split_ctx = env_int("NTTS_ATTN_SPLIT_CTX", 896);
        if (e->attn_split < 2 || e->attn_split > 32) e->attn_split = 0;
        if (e->fp8 || e->generic || (!e->small && B * c->num_kv_heads >= 512)) e->attn_split = 0;   // (fp8: the combine pass writes bf16 rows / fp32 chunk slabs)
        if (e->attn_split) {
            const size_t n_sc = (size_t)B * c->num_kv_heads * kGroupMax * (c->max_context + 16), n_st = (size_t)B * c->num_kv_heads * e->attn_split * kGroupMax * 2,
                         n_os = (size_t)e->attn_split * B * c->num_heads * 64;
            CR_HIP(hipMalloc((void**)&e->as_scores, n_sc * 2)); CR_HIP(hipMemset(e->as_scores, 0, n_sc * 2));
            CR_HIP(hipMalloc((void**)&e->as_stats, n_st * 4)); CR_HIP(hipMemset(e->as_stats, 0, n_st * 4));
            CR_HIP(hipMalloc((void**)&e->as_oslabs, n_os * 4)); CR_HIP(hipMemset(e->as_oslabs, 0, n_os * 4));
        }
        CR_HIP(hipMalloc((void**)&e->h_alt, (size_t)B * H * 2));
        CR_HIP(hipMemset(e->h_alt, 0, (size_t)B * H * 2));
    }
    CR_HIP(hipMalloc((void**)&e->part_val, (size_t)B * e->n_part * 4));
    CR_HIP(hipMalloc((void**)&e->part_idx, (size_t)B * e->n_part * 4));
    e->ldl = ((long)V + 7) / 8 * 8;
    CR_HIP(hipMemset(e->h_dec, 0, (size_t)B * H * 2));
    CR_HIP(hipMemset(e->xn_dec, 0, (size_t)B * H * 2));
    CR_HIP(hipMemset(e->qkv_dec, 0, (size_t)B * e->NQKV * 2));
    CR_HIP(hipMemset(e->attn_dec, 0, (size_t)B * c->num_heads * e->HD * 2));

    // ---- prefill workspaces
    const size_t T = e->Tmax;
    CR_HIP(hipMalloc((void**)&e->h_pf, T * H * 2));
    CR_HIP(hipMalloc((void**)&e->xn_pf, T * H * 2));
    CR_HIP(hipMalloc((void**)&e->qkv_pf, T * e->NQKV * 2));
    CR_HIP(hipMalloc((void**)&e->attn_pf, T * c->num_heads * e->HD * 2));
    CR_HIP(hipMalloc((void**)&e->o_pf, T * H * 2));
    CR_HIP(hipMalloc((void**)&e->act_pf, T * F * 2));
    e->meta_cap = 2 * T + (size_t)B * (16 + e->max_pages) + (T / 64 + 3 * B) * 2 + 6 * (size_t)B + 3 * (size_t)B * e->max_pages + 64;
    CR_HIP(hipMalloc((void**)&e->meta_dev, e->meta_cap * sizeof(int)));
    for (int i = 0; i < ntts_backbone::kMetaStages; ++i) {
        CR_HIP(hipHostMalloc((void**)&e->meta_host[i], e->meta_cap * sizeof(int), hipHostMallocDefault));
        CR_HIP(hipEventCreateWithFlags(&e->meta_ev[i], hipEventDisableTiming));
    }
    e->small_cap = 4 * (size_t)B + 64;
    CR_HIP(hipHostMalloc((void**)&e->small_host, ntts_backbone::kSmallStages * e->small_cap * sizeof(int), hipHostMallocDefault));
    for (int i = 0; i < ntts_backbone::kSmallStages; ++i) CR_HIP(hipEventCreateWithFlags(&e->small_ev[i], hipEventDisableTiming));
    CR_HIP(hipHostMalloc((void**)&e->snap_host, 2 * (size_t)B * sizeof(int), hipHostMallocDefault));
    CR_HIP(hipEventCreateWithFlags(&e->snap_ev, hipEventDisableTiming));
    CR_HIP(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    CR_HIP(hipDeviceSynchronize());
    *out = e;
    return NTTS_OK;
}

extern "C" void ntts_backbone_destroy(ntts_backbone* e) {
    if (!e) return;
    hipSetDevice(e->device);
    hipDeviceSynchronize();
    if (e->graph) hipGraphExecDestroy(e->graph);
    if (e->graph_split) hipGraphExecDestroy(e->graph_split);
    if (e->calib) hipFree(e->calib);
    if (e->head_r) hipFree(e->head_r);
    if (e->shead_r) hipFree(e->shead_r);
    for (hipGraphExec_t g : e->graph_shape)
        if (g && g != e->graph) hipGraphExecDestroy(g);
    bool last_reader = true;
    if (e->share) { last_reader = e->share->refs.fetch_sub(1) == 1; if (last_reader) delete e->share; else e->share->last_decode_ns[e->share_idx].store(0); }
    void* bufs[] = {last_reader ? e->arena : nullptr, e->gu_map_gate, e->gu_map_up, e->kv, e->ibuf, e->h_dec, e->xn_dec, e->qkv_dec, e->attn_dec,
                    e->act_dec, e->slabs, e->slabs2, e->h_alt, e->part_val, e->part_idx, e->logits, e->logits_bf16, e->h_pf, e->xn_pf, e->qkv_pf, e->attn_pf,
                    e->o_pf, e->act_pf, e->meta_dev, e->as_scores, e->as_stats, e->as_oslabs, e->step_meta, e->rope_rows};
    for (void* b : bufs)
        if (b) hipFree(b);
    for (auto& ev : e->ev)
        if (ev) hipEventDestroy(ev);
    for (int i = 0; i < ntts_backbone::kMetaStages; ++i) {
        if (e->meta_host[i]) hipHostFree(e->meta_host[i]);
        if (e->meta_ev[i]) hipEventDestroy(e->meta_ev[i]);
    }
    if (e->small_host) hipHostFree(e->small_host);
    for (int i = 0; i < ntts_backbone::kSmallStages; ++i)
        if (e->small_ev[i]) hipEventDestroy(e->small_ev[i]);
    if (e->snap_host) hipHostFree(e->snap_host);
    if (e->snap_ev) hipEventDestroy(e->snap_ev);
    if (e->copy_stream) hipStreamDestroy(e->copy_stream);
    if (e->pf_stream) { if (!e->pf_lent) hipStreamDestroy(e->pf_stream); hipEventDestroy(e->pf_ev[0]); hipEventDestroy(e->pf_ev[1]); }
    if (e->own_stream) hipStreamDestroy(e->own_stream);
    delete e;
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
static int put_rows(ntts_backbone* e, const void* data, int dtype, int is_device, long rows, long cols, bf16_t* dst,
                    const int* dst_rows) {
    const size_t esz = dtype == NTTS_DT_F32 ? 4 : 2;
    const void* src = data;
    DevScratch tmp;
    if (!is_device) {
        HIPCHK(e, hipMalloc(&tmp.p, (size_t)rows * cols * esz));
        HIPCHK(e, hipMemcpy(tmp.p, data, (size_t)rows * cols * esz, hipMemcpyHostToDevice));
        src = tmp.p;
    }
    NTTS_LAUNCH((pack_rows_kernel), dim3((unsigned)rows), dim3(256), e->stream, src, dtype == NTTS_DT_F32 ? 1 : 0, dst, dst_rows, cols);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return NTTS_OK;
}

// a GEMM weight: rows [row0, row0 + rows) of the packed matrix `dst` ([*, cols]), in the engine's weight layout
static int put_weight(ntts_backbone* e, const void* data, int dtype, int is_device, long rows, long cols, bf16_t* dst, long row0,
                      const int* dst_rows, int tile_major, float* wscale = nullptr) {
    if (dtype == NTTS_DT_FP8_E4M3) {      // a pre-quantised matrix: the bytes as they are (its scales come as "<module>.weight_scale")
        if (!wscale) return fail(e, NTTS_EINVAL, "fp8 bytes for an engine created with weight_dtype = NTTS_W_BF16");
        const void* srcb = data;
        DevScratch tmpb;
        if (!is_device) {
            HIPCHK(e, hipMalloc(&tmpb.p, (size_t)rows * cols));
            HIPCHK(e, hipMemcpy(tmpb.p, data, (size_t)rows * cols, hipMemcpyHostToDevice));
            srcb = tmpb.p;
        }
        NTTS_LAUNCH((pack_weight_fp8_raw_kernel), dim3((unsigned)rows), dim3(256), e->stream, (const unsigned char*)srcb, (unsigned char*)dst, dst_rows, row0, cols);
        HIPCHK(e, hipStreamSynchronize(e->stream));
        return NTTS_OK;
    }
    const size_t esz = dtype == NTTS_DT_F32 ? 4 : 2;
    const void* src = data;
    DevScratch tmp;
    if (!is_device) {
        HIPCHK(e, hipMalloc(&tmp.p, (size_t)rows * cols * esz));
        HIPCHK(e, hipMemcpy(tmp.p, data, (size_t)rows * cols * esz, hipMemcpyHostToDevice));
        src = tmp.p;
    }
    if (wscale)   // fp8 model: quantise per output channel on the way in
        NTTS_LAUNCH((pack_weight_fp8_kernel), dim3((unsigned)rows), dim3(256), e->stream, src, dtype == NTTS_DT_F32 ? 1 : 0, (unsigned char*)dst,
                    wscale, dst_rows, row0, cols);
    else
        NTTS_LAUNCH((pack_weight_kernel), dim3((unsigned)rows), dim3(256), e->stream, src, dtype == NTTS_DT_F32 ? 1 : 0, dst, dst_rows, row0,
                    cols, tile_major);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return NTTS_OK;
}

extern "C" int ntts_backbone_load_tensor(ntts_backbone* e, const char* name, const void* data, int dtype,
                                         const int64_t* shape, int ndim, int is_device) {
    if (!e || !name || !data || !shape) return fail(e, NTTS_EINVAL, "null argument");
    if (e->finalized) return fail(e, NTTS_ESTATE, "weights already finalised");
    HIPCHK(e, hipSetDevice(e->device));
    const ntts_backbone_config& c = e->cfg;
    const long H = e->H, F = e->F, QD = c.num_heads * e->HD, KD = c.num_kv_heads * e->HD;
    std::string n(name);
    auto want = [&](long r, long cc) -> bool {
        return (cc == 0 && ndim == 1 && shape[0] == r) || (cc != 0 && ndim == 2 && shape[0] == r && shape[1] == cc);
    };
    auto bad_shape = [&]() { return fail(e, NTTS_EINVAL, "tensor '%s': unexpected shape", name); };
    if (n == "rope.inv_freq") {
        const int nf = e->HD / 2;
        if (dtype != NTTS_DT_F32 || !want(nf, 0)) return fail(e, NTTS_EINVAL, "rope.inv_freq must be fp32 [%d] (head_dim / 2)", nf);
        if (is_device) HIPCHK(e, hipMemcpy(e->inv_freq, data, nf * 4, hipMemcpyDeviceToHost));
        else memcpy(e->inv_freq, data, nf * 4);
        e->have_inv_freq = true;
        e->loaded.insert(n);
        return NTTS_OK;
    }
    // ---- fp8 model: static input scales (fp32 scalars, naming of static-fp8 checkpoints); k_proj / v_proj / up_proj repeat
    //      the scale of the fused GEMM they belong to and must agree with it
    if (n.size() > 12 && n.compare(n.size() - 12, 12, ".input_scale") == 0) {
        if (!e->fp8) return fail(e, NTTS_EINVAL, "tensor '%s': input scales belong to the fp8 model (weight_dtype = NTTS_W_FP8_E4M3)", name);
        long cnt = 1;
        for (int d = 0; d < ndim; ++d) cnt *= shape[d];
        if (dtype != NTTS_DT_F32 || cnt != 1) return fail(e, NTTS_EINVAL, "tensor '%s' must be one fp32 value", name);
        float v = 0.f;
        if (is_device) HIPCHK(e, hipMemcpy(&v, data, 4, hipMemcpyDeviceToHost));
        else memcpy(&v, data, 4);
        if (!(v > 0.f) || !(v < 3.0e38f)) return fail(e, NTTS_EINVAL, "tensor '%s': scale must be positive and finite", name);
        float* slot = nullptr;
        std::string canon = n;
        if (n == "lm_head.input_scale") slot = &e->xs_head;
        else if (n.rfind("model.layers.", 0) == 0) {
            char* endp = nullptr;
            const long li = strtol(name + 13, &endp, 10);
            if (endp == name + 13 || *endp != '.' || li < 0 || li >= e->cfg.num_layers) return fail(e, NTTS_EINVAL, "tensor '%s': bad layer index", name);
            const std::string t(endp + 1), pre = "model.layers." + std::to_string(li) + ".";
            LayerW& w = e->layers[li];
            if (t == "self_attn.q_proj.input_scale" || t == "self_attn.k_proj.input_scale" || t == "self_attn.v_proj.input_scale") { slot = &w.xs[0]; canon = pre + "self_attn.q_proj.input_scale"; }
            else if (t == "self_attn.o_proj.input_scale") slot = &w.xs[1];
            else if (t == "mlp.gate_proj.input_scale" || t == "mlp.up_proj.input_scale") { slot = &w.xs[2]; canon = pre + "mlp.gate_proj.input_scale"; }
            else if (t == "mlp.down_proj.input_scale") slot = &w.xs[3];
        }
        if (!slot) return fail(e, NTTS_EINVAL, "unknown tensor '%s'", name);
        if (e->loaded.count(canon) && *slot != v)
            return fail(e, NTTS_EINVAL, "tensor '%s' = %g disagrees with the scale already loaded for the same fused GEMM input (%g)", name, v, *slot);
        *slot = v;
        e->loaded.insert(canon);
        return NTTS_OK;
    }
    // ---- pre-quantised fp8 checkpoints: "<module>.weight_scale" (fp32; one value per output channel or one per matrix)
    if (n.size() > 13 && n.compare(n.size() - 13, 13, ".weight_scale") == 0) {
        if (!e->fp8) return fail(e, NTTS_EINVAL, "tensor '%s': weight scales belong to the fp8 model (weight_dtype = NTTS_W_FP8_E4M3)", name);
        if (dtype != NTTS_DT_F32) return fail(e, NTTS_EINVAL, "tensor '%s' must be fp32", name);
        long cnt = 1;
        for (int d = 0; d < ndim; ++d) cnt *= shape[d];
        const std::string mod = n.substr(0, n.size() - 13), wname = mod + ".weight";
        if (e->quantised_here.count(wname)) return fail(e, NTTS_EINVAL, "tensor '%s': the matrix was given in bf16 / fp32 and quantised on upload; it has its scales", name);
        float* dst = nullptr; const int* rows_map = nullptr; long row0 = 0, rows_n = 0;
        if (mod == "lm_head" || mod == "model.embed_tokens") {
            if (mod == "model.embed_tokens" || e->tied) return fail(e, NTTS_EINVAL, "tensor '%s': the embedding stays bf16 (with tie_word_embeddings the head's fp8 copy is derived from it)", name);
            dst = e->shead; rows_n = c.vocab_size;
        } else if (mod.rfind("model.layers.", 0) == 0) {
            char* endp = nullptr;
            const long li = strtol(name + 13, &endp, 10);
            if (endp == name + 13 || *endp != '.' || li < 0 || li >= c.num_layers) return fail(e, NTTS_EINVAL, "tensor '%s': bad layer index", name);
            const std::string t(endp + 1, strlen(endp + 1) - 13);
            LayerW& w = e->layers[li];
            if (t == "self_attn.q_proj") { dst = w.sqkv; rows_n = QD; }
            else if (t == "self_attn.k_proj") { dst = w.sqkv; row0 = QD; rows_n = KD; }
            else if (t == "self_attn.v_proj") { dst = w.sqkv; row0 = QD + KD; rows_n = KD; }
            else if (t == "self_attn.o_proj") { dst = w.so; rows_n = H; }
            else if (t == "mlp.gate_proj") { dst = w.sgu; rows_map = e->gu_map_gate; rows_n = F; }
            else if (t == "mlp.up_proj") { dst = w.sgu; rows_map = e->gu_map_up; rows_n = F; }
            else if (t == "mlp.down_proj") { dst = w.sd; rows_n = H; }
        }
        if (!dst) return fail(e, NTTS_EINVAL, "unknown tensor '%s'", name);
        if (cnt != 1 && (cnt != rows_n || shape[0] != rows_n))      // [rows] or [rows, 1]; never [1, rows] or a transposed block
            return fail(e, NTTS_EINVAL, "tensor '%s': %ld values (leading dimension %ld), expected 1 or [%ld] / [%ld, 1] (one per output channel)", name, cnt, ndim ? (long)shape[0] : 0L, rows_n, rows_n);
        DevScratch tmp;
        const float* src = (const float*)data;
        if (!is_device) {
            HIPCHK(e, hipMalloc(&tmp.p, (size_t)cnt * 4));
            HIPCHK(e, hipMemcpy(tmp.p, data, (size_t)cnt * 4, hipMemcpyHostToDevice));
            src = (const float*)tmp.p;
        }
        NTTS_LAUNCH((scatter_scales_kernel), dim3((unsigned)((rows_n + 255) / 256)), dim3(256), e->stream, src, cnt == 1 ? 1 : 0, dst, rows_map, row0, rows_n);
        HIPCHK(e, hipStreamSynchronize(e->stream));
        e->loaded.insert(n);
        return NTTS_OK;
    }
    std::string mark_scale;
    bool mark_quant = false;
    if (dtype == NTTS_DT_FP8_E4M3) {
        if (!e->fp8) return fail(e, NTTS_EINVAL, "tensor '%s': fp8 bytes for an engine created with weight_dtype = NTTS_W_BF16", name);
        const bool is_matrix = n.size() > 7 && n.compare(n.size() - 7, 7, ".weight") == 0 && (n.find("_proj.weight") != std::string::npos || (n == "lm_head.weight" && !e->tied));
        if (!is_matrix) return fail(e, NTTS_EINVAL, "tensor '%s': only the projection matrices (and an untied lm_head) can be pre-quantised", name);
        mark_scale = n.substr(0, n.size() - 7) + ".weight_scale";          // finalize insists on its scales -- once the bytes are stored (stored())
    } else if (dtype != NTTS_DT_F32 && dtype != NTTS_DT_BF16) return fail(e, NTTS_EINVAL, "tensor '%s': dtype must be f32, bf16 or (fp8 model) e4m3 bytes", name);
    else if (e->fp8 && n.size() > 7 && n.compare(n.size() - 7, 7, ".weight") == 0) {
        if (e->loaded.count(n.substr(0, n.size() - 7) + ".weight_scale"))
            return fail(e, NTTS_EINVAL, "tensor '%s': its weight_scale was loaded, so the matrix must come as e4m3 bytes (NTTS_DT_FP8_E4M3)", name);
        mark_quant = true;
    }
    // Bookkeeping of a matrix of the fp8 model, applied only AFTER the tensor passed its shape check and was stored (a load
    // rejected for its shape used to leave the engine demanding a scale, or refusing one, for a matrix it never stored); a matrix that is
    // loaded again in the other form takes the other form's marker with it.
    auto stored = [&](const std::string& as) {
        e->loaded.insert(as);
        if (!mark_scale.empty()) { e->needed.insert(mark_scale); e->quantised_here.erase(n); }
        if (mark_quant) { e->quantised_here.insert(n); e->needed.erase(n.substr(0, n.size() - 7) + ".weight_scale"); }
    };
    int rc = NTTS_EINVAL;
    const int tm = 1;   // GEMM weights are stored tile-major
    if (n == "model.embed_tokens.weight") {
        if (!want(c.vocab_size, H)) return bad_shape();
        if (e->head_from_embed) {   // a tied "lm_head.weight" arrived first and was taken as the embedding: the two must agree
            DevScratch cnt, tmp;
            const void* src = data;
            const size_t esz = dtype == NTTS_DT_F32 ? 4 : 2;
            if (!is_device) {
                HIPCHK(e, hipMalloc(&tmp.p, (size_t)c.vocab_size * H * esz));
                HIPCHK(e, hipMemcpy(tmp.p, data, (size_t)c.vocab_size * H * esz, hipMemcpyHostToDevice));
                src = tmp.p;
            }
            HIPCHK(e, hipMalloc(&cnt.p, 4));
            unsigned int* cntp = (unsigned int*)cnt.p;     // (a plain pointer for the launch: the emulator's launch captures by value)
            const bf16_t* refp = e->embed;
            HIPCHK(e, hipMemsetAsync(cntp, 0, 4, e->stream));
            NTTS_LAUNCH((rows_mismatch_kernel), dim3((unsigned)c.vocab_size), dim3(256), e->stream, src, dtype == NTTS_DT_F32 ? 1 : 0,
                        refp, (long)H, cntp);
            unsigned int bad = 0;
            HIPCHK(e, hipMemcpyAsync(&bad, cntp, 4, hipMemcpyDeviceToHost, e->stream));
            HIPCHK(e, hipStreamSynchronize(e->stream));
            if (bad) return fail(e, NTTS_EINVAL, "tie_word_embeddings is set but 'lm_head.weight' differs from 'model.embed_tokens.weight' in %u values: "
                                                 "create the engine with tie_word_embeddings = 0 for an untied head", bad);
            stored(n);
            return NTTS_OK;
        }
        rc = put_rows(e, data, dtype, is_device, c.vocab_size, H, e->embed, nullptr);
        if (rc == NTTS_OK && e->tied)   // the tied head's own copy (tile-major and / or fp8)
            rc = put_weight(e, data, dtype, is_device, c.vocab_size, H, e->embed_tm, 0, nullptr, tm, e->fp8 ? e->shead : nullptr);
        if (rc == NTTS_OK) stored(n);
        return rc;
    }
    if (n == "lm_head.weight") {
        if (!want(c.vocab_size, H)) return bad_shape();
        if (!e->tied) {             // a separate head matrix
            rc = put_weight(e, data, dtype, is_device, c.vocab_size, H, e->embed_tm, 0, nullptr, tm, e->fp8 ? e->shead : nullptr);
            if (rc == NTTS_OK) stored(n);
            return rc;
        }
        // tied: the tensor may be present in a checkpoint (safetensors of some exporters keep both names) but it must BE the embedding
        if (!e->loaded.count("model.embed_tokens.weight")) {
            rc = put_rows(e, data, dtype, is_device, c.vocab_size, H, e->embed, nullptr);
            if (rc == NTTS_OK) rc = put_weight(e, data, dtype, is_device, c.vocab_size, H, e->embed_tm, 0, nullptr, tm, e->fp8 ? e->shead : nullptr);
            if (rc == NTTS_OK) { e->head_from_embed = true; stored("lm_head.weight"); }
            return rc;
        }
        {
            DevScratch cnt, tmp;
            const void* src = data;
            const size_t esz = dtype == NTTS_DT_F32 ? 4 : 2;
            if (!is_device) {
                HIPCHK(e, hipMalloc(&tmp.p, (size_t)c.vocab_size * H * esz));
                HIPCHK(e, hipMemcpy(tmp.p, data, (size_t)c.vocab_size * H * esz, hipMemcpyHostToDevice));
                src = tmp.p;
            }
            HIPCHK(e, hipMalloc(&cnt.p, 4));
            unsigned int* cntp = (unsigned int*)cnt.p;     // (a plain pointer for the launch: the emulator's launch captures by value)
            const bf16_t* refp = e->embed;
            HIPCHK(e, hipMemsetAsync(cntp, 0, 4, e->stream));
            NTTS_LAUNCH((rows_mismatch_kernel), dim3((unsigned)c.vocab_size), dim3(256), e->stream, src, dtype == NTTS_DT_F32 ? 1 : 0,
                        refp, (long)H, cntp);
            unsigned int bad = 0;
            HIPCHK(e, hipMemcpyAsync(&bad, cntp, 4, hipMemcpyDeviceToHost, e->stream));
            HIPCHK(e, hipStreamSynchronize(e->stream));
            if (bad) return fail(e, NTTS_EINVAL, "tie_word_embeddings is set but 'lm_head.weight' differs from 'model.embed_tokens.weight' in %u values: "
                                                 "create the engine with tie_word_embeddings = 0 for an untied head", bad);
        }
        stored("lm_head.weight");
        return NTTS_OK;
    }
    if (n == "model.norm.weight") {
        if (!want(H, 0)) return bad_shape();
        rc = put_rows(e, data, dtype, is_device, 1, H, e->final_norm, nullptr);
    } else if (n.rfind("model.layers.", 0) == 0) {
        const char* s = name + 13;
        char* endp = nullptr;
        const long li = strtol(s, &endp, 10);
        if (endp == s || *endp != '.' || li < 0 || li >= c.num_layers) return fail(e, NTTS_EINVAL, "tensor '%s': bad layer index", name);
        const std::string t(endp + 1);
        LayerW& w = e->layers[li];
        if (t == "input_layernorm.weight") { if (!want(H, 0)) return bad_shape(); rc = put_rows(e, data, dtype, is_device, 1, H, w.ln1, nullptr); }
        else if (t == "post_attention_layernorm.weight") { if (!want(H, 0)) return bad_shape(); rc = put_rows(e, data, dtype, is_device, 1, H, w.ln2, nullptr); }
        else if (t == "self_attn.q_proj.weight") { if (!want(QD, H)) return bad_shape(); rc = put_weight(e, data, dtype, is_device, QD, H, w.wqkv, 0, nullptr, tm, w.sqkv); }
        else if (t == "self_attn.k_proj.weight") { if (!want(KD, H)) return bad_shape(); rc = put_weight(e, data, dtype, is_device, KD, H, w.wqkv, QD, nullptr, tm, w.sqkv); }
        else if (t == "self_attn.v_proj.weight") { if (!want(KD, H)) return bad_shape(); rc = put_weight(e, data, dtype, is_device, KD, H, w.wqkv, QD + KD, nullptr, tm, w.sqkv); }
        else if (t == "self_attn.q_proj.bias" || t == "self_attn.k_proj.bias" || t == "self_attn.v_proj.bias") {
            if (!e->has_bias) return fail(e, NTTS_EINVAL, "tensor '%s': the engine was created with attention_bias = 0", name);
            const long rows_n = t[10] == 'q' ? QD : KD, at = t[10] == 'q' ? 0 : (t[10] == 'k' ? QD : QD + KD);
            if (!want(rows_n, 0)) return bad_shape();
            rc = put_rows(e, data, dtype, is_device, 1, rows_n, w.bqkv + at, nullptr);
        }
        else if (t == "self_attn.q_norm.weight" || t == "self_attn.k_norm.weight") {
            if (!e->qk_norm) return fail(e, NTTS_EINVAL, "tensor '%s': the engine was created with qk_norm = 0", name);
            if (!want(e->HD, 0)) return bad_shape();
            rc = put_rows(e, data, dtype, is_device, 1, e->HD, t[10] == 'q' ? w.qn : w.kn, nullptr);
        }
        else if (t == "self_attn.o_proj.weight") { if (!want(H, QD)) return bad_shape(); rc = put_weight(e, data, dtype, is_device, H, QD, w.wo, 0, nullptr, tm, w.so); }
        else if (t == "mlp.gate_proj.weight") { if (!want(F, H)) return bad_shape(); rc = put_weight(e, data, dtype, is_device, F, H, w.wgu, 0, e->gu_map_gate, tm, w.sgu); }
        else if (t == "mlp.up_proj.weight") { if (!want(F, H)) return bad_shape(); rc = put_weight(e, data, dtype, is_device, F, H, w.wgu, 0, e->gu_map_up, tm, w.sgu); }
        else if (t == "mlp.down_proj.weight") { if (!want(H, F)) return bad_shape(); rc = put_weight(e, data, dtype, is_device, H, F, w.wd, 0, nullptr, tm, w.sd); }
        else return fail(e, NTTS_EINVAL, "unknown tensor '%s'", name);
    } else {
        return fail(e, NTTS_EINVAL, "unknown tensor '%s'", name);
    }
    if (rc == NTTS_OK) stored(n);
    return rc;
}

static int build_rope(ntts_backbone* e) {
    // Qwen2RotaryEmbedding.forward hf:models/qwen2/modeling_qwen2.py:91-102: angle = fp32(pos * inv_freq),
    // cos/sin in fp32, cast to bf16.  cos/sin are evaluated in double and rounded once to fp32.
    const int n = e->cfg.max_context, nf = e->HD / 2;
    std::vector<bf16_t> c((size_t)n * nf), s((size_t)n * nf);
    auto tobf = [](float f) {
        uint32_t u;
        memcpy(&u, &f, 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (bf16_t)(u >> 16);
    };
    for (int p = 0; p < n; ++p)
        for (int i = 0; i < nf; ++i) {
            const float ang = (float)p * e->inv_freq[i];
            c[(size_t)p * nf + i] = tobf((float)cos((double)ang));
            s[(size_t)p * nf + i] = tobf((float)sin((double)ang));
        }
    HIPCHK(e, hipMemcpy(e->rope_cos, c.data(), c.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->rope_sin, s.data(), s.size() * 2, hipMemcpyHostToDevice));
    return NTTS_OK;
}

static int sync_input_scales(ntts_backbone* e, bool to_device) {   // fp8: host copies <-> the arena's copies (what the broadcast carries)
    if (!e->fp8) return NTTS_OK;
    const hipMemcpyKind k = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    for (auto& w : e->layers) {
        if (to_device) HIPCHK(e, hipMemcpy(w.xs_dev, w.xs, 16, k)); else HIPCHK(e, hipMemcpy(w.xs, w.xs_dev, 16, k));
    }
    if (to_device) HIPCHK(e, hipMemcpy(e->xs_head_dev, &e->xs_head, 4, k)); else HIPCHK(e, hipMemcpy(&e->xs_head, e->xs_head_dev, 4, k));
    return NTTS_OK;
}

extern "C" int ntts_backbone_finalize(ntts_backbone* e) {
    if (!e) return NTTS_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    if (!e->have_inv_freq) return fail(e, NTTS_ESTATE, "rope.inv_freq not loaded");
    std::string missing;
    int n_missing = 0;
    for (const auto& name : e->needed)
        if (!e->loaded.count(name) && !(name == "model.embed_tokens.weight" && e->head_from_embed)) {
            if (n_missing++ < 6) missing += (missing.empty() ? "" : ", ") + name;
        }
    if (n_missing) return fail(e, NTTS_ESTATE, "%d tensors not loaded: %s%s", n_missing, missing.c_str(), n_missing > 6 ? ", ..." : "");
    int rc = build_rope(e);
    if (rc) return rc;
    rc = sync_input_scales(e, true);
    if (rc) return rc;
    e->finalized = true;
    return NTTS_OK;
}

extern "C" int ntts_backbone_arena(ntts_backbone* e, void** dev_ptr, size_t* bytes) {
    if (!e || !dev_ptr || !bytes) return NTTS_EINVAL;
    *dev_ptr = e->arena;
    *bytes = e->arena_elems * sizeof(bf16_t);
    return NTTS_OK;
}
extern "C" int ntts_backbone_arena_copy(ntts_backbone* e, void* buf, size_t bytes, int to_arena) {
    if (!e || !buf || bytes != e->arena_elems * sizeof(bf16_t)) return fail(e, NTTS_EINVAL, "arena copy: bad buffer / size");
    if (to_arena && e->arena_shared) return fail(e, NTTS_ESTATE, "arena copy: this engine reads another engine's arena");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (to_arena) HIPCHK(e, hipMemcpy(e->arena, buf, bytes, hipMemcpyDeviceToDevice));
    else HIPCHK(e, hipMemcpy(buf, e->arena, bytes, hipMemcpyDeviceToDevice));
    return NTTS_OK;
}
// The part of the arena that is DERIVED from another part: the tied head's own copy of the embedding (tile-major and / or fp8
// with its scales).  A broadcast can skip [*off, *off + *bytes) and have the receiver rebuild it (ntts_backbone_adopt_arena does).
extern "C" int ntts_backbone_arena_derived(ntts_backbone* e, size_t* off, size_t* bytes) {
    if (!e || !off || !bytes) return NTTS_EINVAL;
    *off = 0; *bytes = 0;
    if (e->tied) {
        const size_t V64 = (size_t)((e->cfg.vocab_size + 63) / 64) * 64;
        *off = (size_t)((char*)e->embed_tm - (char*)e->arena);
        *bytes = e->fp8 ? ((V64 * e->H + 1) / 2) * 2 : V64 * e->H * 2;     // the matrix (its fp8 scales follow and are rebuilt too)
    }
    return NTTS_OK;
}

extern "C" int ntts_backbone_adopt_arena(ntts_backbone* e) {
    if (!e) return NTTS_EINVAL;
    if (e->tied) {   // rebuild the derived head copy from the (received) embedding: it need not travel
        HIPCHK(e, hipSetDevice(e->device));
        const int rc0 = put_weight(e, e->embed, NTTS_DT_BF16, 1, e->cfg.vocab_size, e->H, e->embed_tm, 0, nullptr, 1,
                                   e->fp8 ? e->shead : nullptr);
        if (rc0) return rc0;
    }
    HIPCHK(e, hipSetDevice(e->device));
    const int rc = sync_input_scales(e, false);   // fp8: the launches need the input scales on the host
    if (rc) return rc;
    e->finalized = true;  // arena (incl. the RoPE table) was filled by a broadcast from a finalised engine
    return NTTS_OK;
}

// A second engine on the SAME weights: e gives up its own (still empty) arena and reads the donor's -- weights, scales and the RoPE
// table are read-only once finalised.  Its KV pool, slot state, workspaces, stream and step graph stay its own.  What running
// several decode chains side by side needs (two 256-slot engines whose step graphs are replayed alternately: each chain fills the
// other's launch gaps, and the second chain finds the layer's weights in the memory-side cache).  Donor and readers may be destroyed in any order: the arena
// is reference-counted and goes with its last reader.
extern "C" int ntts_backbone_share_arena(ntts_backbone* e, ntts_backbone* donor) {
    if (!e || !donor || e == donor) return NTTS_EINVAL;
    if (!donor->finalized) return fail(e, NTTS_ESTATE, "share_arena: the donor engine is not finalised");
    if (e->finalized || e->arena_shared || e->graph || e->graph_split) return fail(e, NTTS_ESTATE, "share_arena: this engine already holds weights");
    const ntts_backbone_config &a = e->cfg, &b = donor->cfg;
    if (e->device != donor->device || e->arena_elems != donor->arena_elems || a.vocab_size != b.vocab_size || a.hidden_size != b.hidden_size ||
        a.intermediate_size != b.intermediate_size || a.num_layers != b.num_layers || a.num_heads != b.num_heads ||
        a.num_kv_heads != b.num_kv_heads || a.max_context != b.max_context || e->fp8 != donor->fp8 || e->tied != donor->tied || e->HD != donor->HD || e->qk_norm != donor->qk_norm ||
        e->has_bias != donor->has_bias)
        return fail(e, NTTS_EINVAL, "share_arena: the two engines differ in device, geometry, weight type or context length");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipStreamSynchronize(donor->stream));
    bf16_t* old = e->arena;
    auto mv = [&](auto*& ptr) { if (ptr) ptr = (std::remove_reference_t<decltype(ptr)>)((char*)donor->arena + ((char*)ptr - (char*)old)); };
    mv(e->embed); mv(e->embed_tm); mv(e->shead); mv(e->xs_head_dev); mv(e->final_norm); mv(e->rope_cos); mv(e->rope_sin);
    for (LayerW& w : e->layers) {
        mv(w.ln1); mv(w.wqkv); mv(w.bqkv); mv(w.wo); mv(w.ln2); mv(w.wgu); mv(w.wd); mv(w.qn); mv(w.kn);
        mv(w.sqkv); mv(w.so); mv(w.sgu); mv(w.sd); mv(w.xs_dev);
    }
    if (donor->share->next_idx.load() >= ArenaShare::kMaxEngines) return fail(e, NTTS_EINVAL, "share_arena: at most %d engines on one arena", ArenaShare::kMaxEngines);
    HIPCHK(e, hipFree(old));
    delete e->share;
    e->arena = donor->arena;
    e->share = donor->share;
    e->share->refs.fetch_add(1);
    e->share_idx = e->share->next_idx.fetch_add(1);
    e->arena_shared = true;
    const int rc = sync_input_scales(e, false);   // fp8: the launches need the input scales on the host
    if (rc) return rc;
    e->finalized = true;
    return NTTS_OK;
}

static void drop_graphs(ntts_backbone* e) {
    for (hipGraphExec_t& g : e->graph_shape) {           // (the other shape's parked capture)
        if (g && g != e->graph) hipGraphExecDestroy(g);
        g = nullptr;
    }
    e->graph_tried_shape[0] = e->graph_tried_shape[1] = false;
    if (e->graph) { hipGraphExecDestroy(e->graph); e->graph = nullptr; }
    if (e->graph_split) { hipGraphExecDestroy(e->graph_split); e->graph_split = nullptr; }
    e->graph_tried = e->graph_split_tried = false;
}

// Opt-in: the lm_head over the token ids [lo, hi) and `eos_id` only (ABI 8; SURVEY.md section 7 "hard parts").  Everything outside gets
// logit -inf: greedy ids equal the full head's whenever the full-vocabulary argmax lies in the range (what a trained NeuTTS checkpoint
// emits after its prompt: ref:neutts/neutts.py:276 keeps only <|speech_N|> ids, :336-341 stops at the one EOS id), and they DIFFER
// otherwise -- this is a serving option, never the parity configuration.  lo < 0 restores the full head.  Requests must then carry
// eos_token_id == eos_id (refused at prefill otherwise).  No slot may be running.
extern "C" int ntts_backbone_set_logits_range(ntts_backbone* e, int32_t lo, int32_t hi, int32_t eos_id) {
    if (!e) return NTTS_EINVAL;
    if (!e->finalized) return fail(e, NTTS_ESTATE, "set_logits_range: weights not finalised");
    const int V = e->cfg.vocab_size, H = e->H;
    for (const HostSlot& sl : e->slots)
        if (sl.state != SLOT_FREE) return fail(e, NTTS_ESTATE, "set_logits_range: a slot is in use");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    drop_graphs(e);
    if (e->head_r) { hipFree(e->head_r); e->head_r = nullptr; }
    if (e->shead_r) { hipFree(e->shead_r); e->shead_r = nullptr; }
    e->lr_lo = e->lr_hi = e->lr_eos = -1; e->lr_rows = 0; e->n_part = e->n_part_full;
    if (lo < 0) return NTTS_OK;
    if (hi <= lo || hi > V || eos_id < 0 || eos_id >= V || (eos_id >= lo && eos_id < hi))
        return fail(e, NTTS_EINVAL, "set_logits_range: need 0 <= lo < hi <= vocab_size (%d) and an eos id outside [lo, hi): got [%d, %d), eos %d", V, lo, hi, eos_id);
    const int rows = hi - lo + 1, padded = (rows + 63) / 64 * 64;
    std::vector<int> map(padded);
    for (int r = 0; r < padded; ++r) map[r] = r < rows - 1 ? lo + r : eos_id;       // (padding rows repeat the EOS row; their columns are never valid)
    DevScratch dmap;
    HIPCHK(e, hipMalloc(&dmap.p, (size_t)padded * sizeof(int)));
    HIPCHK(e, hipMemcpy(dmap.p, map.data(), (size_t)padded * sizeof(int), hipMemcpyHostToDevice));
    const size_t kb = (size_t)H * (e->fp8 ? 1 : 2);
    HIPCHK(e, hipMalloc((void**)&e->head_r, (size_t)padded * kb));
    if (e->fp8) HIPCHK(e, hipMalloc((void**)&e->shead_r, (size_t)padded * sizeof(float)));
    {   // (plain pointers for the launch: the emulator's launch captures its arguments by value)
        const unsigned char* srcp = (const unsigned char*)e->embed_tm;
        unsigned char* dstp = (unsigned char*)e->head_r;
        const int* mapp = (const int*)dmap.p;
        const float* scs = e->fp8 ? e->shead : nullptr;
        float* scd = e->shead_r;
        const long kbl = (long)kb;
        NTTS_LAUNCH((gather_head_rows_kernel), dim3((unsigned)padded), dim3(64), e->stream, srcp, dstp, mapp, kbl, scs, scd);
    }
    // (the rows kept for the top-k sampler and the debug tap are in COLUMN order while a range is set: the sampler looks at the first
    //  lr_rows columns, ntts_backbone_read_logits scatters them back to token ids)
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipGetLastError());
    e->lr_lo = lo; e->lr_hi = hi; e->lr_eos = eos_id; e->lr_rows = rows;
    e->n_part = n_part_for(e, rows);
    return NTTS_OK;
}

// ---- fp8 activation-scale calibration (ABI 8; VERDICT r4 missing 4).  The reference's quantised builds are ready-made files
// (ref:README.md:59-64); a user holding a bf16 checkpoint gets the static `input_scale`s of the fp8 model from data: calibration mode
// on a BF16 engine records max |x| of every GEMM's input over the prompt passes that follow, ntts_backbone_read_amax hands the
// 4 * num_layers + 1 values out ([layer][QKV, o_proj, gate/up, down_proj], lm_head last); scale = amax / 448 (tools/calibrate_fp8.py).
extern "C" int ntts_backbone_calibrate(ntts_backbone* e, int32_t enable) {
    if (!e) return NTTS_EINVAL;
    if (e->fp8) return fail(e, NTTS_EINVAL, "calibrate: the activations of the fp8 model are already quantised; calibrate on a bf16 engine of the same checkpoint");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    const size_t nslot = (size_t)e->cfg.num_layers * 4 + 1;
    if (enable && !e->calib) {
        HIPCHK(e, hipMalloc((void**)&e->calib, nslot * sizeof(float)));
    }
    if (enable) HIPCHK(e, hipMemset(e->calib, 0, nslot * sizeof(float)));
    else if (e->calib) { HIPCHK(e, hipFree(e->calib)); e->calib = nullptr; }
    return NTTS_OK;
}
extern "C" int ntts_backbone_read_amax(ntts_backbone* e, float* out, int32_t n) {
    if (!e || !out) return NTTS_EINVAL;
    if (!e->calib) return fail(e, NTTS_ESTATE, "read_amax: calibration mode is off (ntts_backbone_calibrate(e, 1) first)");
    if (n != e->cfg.num_layers * 4 + 1) return fail(e, NTTS_EINVAL, "read_amax: %d values asked, the engine records %d (4 per layer + lm_head)", n, e->cfg.num_layers * 4 + 1);
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(out, e->calib, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return NTTS_OK;
}

// The decode shape for `chains` chains side by side, with that shape's captured step: both shapes keep their capture, so an engine that
// leaves a gang for a while (NeuTTS.infer on engine 0 of a gang: ADVICE r5) and comes back re-captures nothing.  Same arithmetic and the
// same summation order per output element in both shapes (tests/test_gpu_parity_matrix.py asserts the logits rows bit-identical).
static void use_shape_for(ntts_backbone* e, int chains) {
    const int B = e->dec_rows;
    const int want = (chains >= 2 && B > 128 && B <= 256) ? 1 : 0;
    e->gang = chains;
    if (want == e->shape) return;
    e->graph_shape[e->shape] = e->graph; e->graph_tried_shape[e->shape] = e->graph_tried;
    e->shape = want;
    e->graph = e->graph_shape[want]; e->graph_tried = e->graph_tried_shape[want];
    e->graph_shape[want] = nullptr;
    apply_gang_shape(e);
}
// engines of this arena with a decode call inside the window, this one included
static int chains_now(ntts_backbone* e) {
    const long long t = now_ns();
    e->share->last_decode_ns[e->share_idx].store(t);
    int n = 1;
    const int hi = e->share->next_idx.load();
    for (int i = 0; i < hi && i < ArenaShare::kMaxEngines; ++i) {
        if (i == e->share_idx) continue;
        const long long o = e->share->last_decode_ns[i].load();
        if (o != 0 && t - o < ntts_backbone::kGangWindowNs) ++n;
    }
    return n;
}

// chains > 0 pins the count (the engine keeps that shape whoever else decodes; in effect at once, e.g. for ntts_backbone_time_kernel);
// 0 returns to counting (the default since ABI 9)
extern "C" int ntts_backbone_set_gang(ntts_backbone* e, int32_t chains) {
    if (!e) return NTTS_EINVAL;
    if (chains < 0) return fail(e, NTTS_EINVAL, "set_gang: %d chains", chains);
    e->gang_forced = chains;
    if (chains > 0) use_shape_for(e, chains);
    return NTTS_OK;
}

// ------------------------------------------------------------------------------------------------
// model passes
// ------------------------------------------------------------------------------------------------
// fp8 model: X and W hold e4m3 bytes (ldx / ldw / K still count elements), wscale = the matrix's per-row scales, xscale = the
// static scale its input was quantised with
static GemmArgs gemm_args(const ntts_backbone* e, const bf16_t* X, long ldx, const bf16_t* W, long ldw, const bf16_t* bias, void* out,
                          long ldo, int M, int N, int K, const float* wscale = nullptr, float xscale = 1.f) {
    GemmArgs a{};
    a.w_tile_major = 1;   // every W this engine hands to a GEMM is in its weight layout
    a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.bias = bias; a.out = out; a.ldo = ldo; a.M = M; a.N = N; a.K = K;
    a.wscale = wscale; a.xscale = xscale;
    a.tl = e->gemv_tl;
    return a;
}
static int ktile_of(const ntts_backbone* e) { return e->fp8 ? 128 : 64; }

// ---- the decode step's launches, one helper per kernel (shared by decode_step and ntts_backbone_time_kernel)
// 64 x 64 tile, 4 waves; LDS ring of 4 slots (3 for gate/up): latency-bound kernels, <= 1 block per CU (profiles/r01_sweep_decode.jsonl)
template <int EPI, int NS = 4>
static void gemm_skinny(const GemmArgs& a, int ks, hipStream_t st) {
    if (a.wscale) gemm_launch<4, 1, 1, EPI, NS, 0, 64, false, true>(a, ks, st);   // fp8 operands
    else gemm_launch<4, 1, 1, EPI, NS>(a, ks, st);
}

// 256 x 64 tile, 8 waves (32 x 64 each), 3-slot ring of 40 KB: the whole decode batch of a <= 256-row chain is one m-block (ntts_backbone::tall)
template <int EPI>
static void gemm_tall(const GemmArgs& a, int ks, hipStream_t st) {
    if (a.wscale) gemm_launch<8, 1, 2, EPI, 3, 0, 64, false, true>(a, ks, st);
    else gemm_launch<8, 1, 2, EPI, 3>(a, ks, st);
}

template <int EPI>
static void gemm_large(ntts_backbone* e, const GemmArgs& a, hipStream_t st) {
    // Tile by how many tiles the GEMM has, per GEMM (a prompt pass of P x 500 rows; profiles/r03i_sweep_prefill_tiles.txt): 256 x 256 / 16 waves
    // once that grid has >= 140 tiles (gate/up from ~1 000 rows, the N = 896 GEMMs from ~9 000), else 128 x 128 / 4 waves once THAT grid has
    // >= 240, else the decode step's 64 x 64 skinny tile -- a 500-row pass is 28 tiles of 128 x 128 on 256 CUs, 112 of 64 x 64: one prompt
    // 5.36 -> 3.72 ms, 4 prompts 6.12 -> 5.06, 16 prompts 9.99 -> 9.13; from 20 prompts on nothing changes.  Same k order per output element
    // on every tile: same bits.  NTTS_XL_MIN_M > 0 (tests): the 256-row tiles from that many rows on, whatever the tile count.
    const long tiles_xl = (long)((a.M + 255) / 256) * ((a.N + 255) / 256), tiles_l = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const bool xl = a.N >= 256 && (e->xl_min_m > 0 ? a.M >= e->xl_min_m : tiles_xl >= 140);
    if (!xl && tiles_l < 240 && e->xl_min_m == 0) { gemm_skinny<EPI, 4>(a, 1, st); return; }   // (bf16 and fp8 alike)
    if (e->fp8) {
        if (xl) gemm_launch<4, 4, 4, EPI, 2, 0, 64, false, true>(a, 1, st);
        else gemm_launch<2, 2, 4, EPI, 2, 0, 64, false, true>(a, 1, st);
        return;
    }
    if (xl) {
        // prefill QKV (N = 1152 = 4.5 x 256): the 256 x 256 tile needs 5 column blocks, the last one half empty -- 625 tiles
        // = 2.44 rounds of the CUs per 32 000-token chunk; the natural-order 256 x 288 tile needs 4: 500 tiles = 1.95 rounds
        if constexpr (EPI == EPI_BF16) {
            if (a.N % 288 == 0 && a.N % 256 != 0) { gemm_launch<4, 3, 4, EPI_BF16, 2, 0, 64, false, false, 6>(a, 1, st); return; }
        }
        NTTS_GEMM_XL(EPI, a, 1, st);
        return;
    }
    gemm_launch<2, 2, 4, EPI, 2>(a, 1, st);
}

static void ks_lm_head(ntts_backbone* e, bool keep_logits);
// rows: e->dec_rows for a decode step; every slot (cfg.max_batch, parked ones included) for the first token behind a prompt pass
static void k_lm_head(ntts_backbone* e, bool keep_logits, int rows) {
    if (e->small) { ks_lm_head(e, keep_logits); return; }
    const int B = rows, H = e->H, V = e->cfg.vocab_size;
    GemmArgs a = gemm_args(e, e->xn_dec, H, e->lr_rows ? e->head_r : e->embed_tm, H, nullptr, nullptr, 0, B, e->lr_rows ? e->lr_rows : V, H,
                           e->lr_rows ? e->shead_r : e->shead, e->xs_head);
    if (e->lr_rows) a.eos_col1 = e->lr_rows;          // compacted head: columns = [range | EOS]
    a.part_val = e->part_val; a.part_idx = e->part_idx; a.mask_eos = e->sl.mask_eos;
    a.logits = keep_logits ? e->logits : nullptr; a.ld_logits = V;
    a.logits_bf16 = (keep_logits && e->n_sampling > 0) ? e->logits_bf16 : nullptr; a.ld_logits_bf16 = e->ldl;
    switch (e->head_tile) {     // (the 256-row tiles stream W with the non-temporal policy: read once per step)
        case 0: gemm_skinny<EPI_ARGMAX>(a, 1, e->stream); break;
        case 4: gemm_launch<4, 3, 4, EPI_ARGMAX, 2, 0, 64, true, false, 6>(a, 1, e->stream); break;   // natural-order 256 x 288 (bf16)
        case 2:
            if (e->fp8) gemm_launch<4, 4, 4, EPI_ARGMAX, 2, 0, 64, true, true>(a, 1, e->stream);
            else gemm_launch<4, 4, 4, EPI_ARGMAX, 2, 0, 64, true>(a, 1, e->stream);
            break;
        default:
            if (e->fp8) gemm_launch<2, 2, 4, EPI_ARGMAX, 2, 0, 64, false, true>(a, 1, e->stream);
            else gemm_launch<2, 2, 4, EPI_ARGMAX, 2>(a, 1, e->stream);
    }
}

static void lm_head_and_sample(ntts_backbone* e, int phase) {
    const int rows = phase == SLOT_PREFILLED ? e->cfg.max_batch : e->dec_rows;
    k_lm_head(e, true, rows);
    SampleArgs s{};
    s.part_val = e->part_val; s.part_idx = e->part_idx; s.n_part = e->n_part; s.sl = e->sl; s.phase = phase;
    s.part_width = e->small ? 16 : e->head_tile == 4 ? 96 : 64;
    s.logits = e->n_sampling > 0 ? e->logits_bf16 : nullptr; s.ld_logits = e->ldl; s.vocab = e->lr_rows ? e->lr_rows : e->cfg.vocab_size;
    if (e->lr_rows) { s.n_range = e->lr_rows - 1; s.id_base = e->lr_lo; s.id_tail = e->lr_eos; }
    NTTS_LAUNCH((sample_greedy_kernel), dim3(rows), dim3(256), e->stream, s);
}

static StepMetaArgs step_meta_args(ntts_backbone* e) {
    StepMetaArgs m{};
    m.pos = e->sl.pos; m.state = e->sl.state; m.block_table = e->block_table; m.max_pages = e->max_pages; m.max_ctx = e->cfg.max_context;
    m.M = e->dec_rows; m.rope_cos = e->rope_cos; m.rope_sin = e->rope_sin; m.meta = e->step_meta; m.rope_rows = e->rope_rows;
    return m;
}
static void k_step_meta(ntts_backbone* e) {   // once per decode step, before the first fused QKV kernel
    NTTS_LAUNCH((step_meta_kernel), dim3((e->dec_rows + 3) / 4), dim3(256), e->stream, step_meta_args(e));
}

// QKV projection + bias + rounding + RoPE + K append (qkv_rope.h); 3 ring slots, 2 K slices per workgroup (swept: 4 slices / 4 and
// 6 slots are equal or slower, profiles/r03a_sweep_qkv_fused.log)
// the rope / qk-norm / KV-append step of the generic attention geometry (attn_prefill.h rope_norm_kv_write_kernel), decode rows
static void k_rope_norm_decode(ntts_backbone* e, int i) {
    const ntts_backbone_config& c = e->cfg;
    RopeNormArgs r{};
    r.qkv = e->qkv_dec; r.ld_qkv = e->NQKV; r.kpool = e->kv + (size_t)i * e->layer_stride; r.vpool = r.kpool + e->kv_half;
    r.block_table = e->block_table; r.max_pages = e->max_pages; r.dec_pos = e->sl.pos; r.dec_state = e->sl.state;
    r.rope_cos = e->rope_cos; r.rope_sin = e->rope_sin; r.q_norm = e->layers[i].qn; r.k_norm = e->layers[i].kn; r.eps = c.rms_eps;
    r.nh = c.num_heads; r.nkv = c.num_kv_heads; r.rows = e->dec_rows; r.write_v = 0;
    const long items = (long)r.rows * (c.num_heads + 2 * c.num_kv_heads);
    if (e->HD == 128) NTTS_LAUNCH((rope_norm_kv_write_kernel<128>), dim3((unsigned)((items + 3) / 4)), dim3(256), e->stream, r);
    else NTTS_LAUNCH((rope_norm_kv_write_kernel<64>), dim3((unsigned)((items + 3) / 4)), dim3(256), e->stream, r);
}

static void k_qkv(ntts_backbone* e, int i) {
    const int B = e->dec_rows, H = e->H;
    const LayerW& w = e->layers[i];
    if (e->generic) {     // plain GEMM (bias, one rounding) into the q|k|v row, then norm + RoPE + K append as one small launch
        GemmArgs g = gemm_args(e, e->xn_dec, H, w.wqkv, H, w.bqkv, e->qkv_dec, e->NQKV, B, e->NQKV, H);
        if (B > 128) gemm_launch<2, 2, 4, EPI_BF16, 2>(g, 1, e->stream); else gemm_skinny<EPI_BF16>(g, 1, e->stream);
        k_rope_norm_decode(e, i);
        return;
    }
    QkvRopeArgs a{};
    a.X = e->xn_dec; a.ldx = H; a.W = w.wqkv; a.bias = w.bqkv; a.wscale = w.sqkv; a.xscale = w.xs[0];
    a.M = B; a.N = e->NQKV; a.K = H; a.meta = e->step_meta; a.rope_rows = e->rope_rows;
    a.q_out = e->qkv_dec; a.ld_q = e->NQKV; a.kpool = e->kv + (size_t)i * e->layer_stride;
    a.nh = e->cfg.num_heads; a.nkv = e->cfg.num_kv_heads;
    a.tl = e->gemv_tl;
    if (e->wide && !e->fp8 && e->wide_qkv > 1) {   // (fp8: the 64-row kernel measured the same, 241.7 vs 241.3 k on nano-fp8: kept narrow)
        if (e->wide_qkv == 4) qkv_rope_launch_wide<false, 4>(a, e->stream);
        else qkv_rope_launch_wide<false, 2>(a, e->stream);
        return;
    }
    const bool place = (e->xcd_affine & 4) && e->xcd_xps;
    if (e->fp8) qkv_rope_launch<true>(a, place, e->stream, e->qkv_wstat);
    else qkv_rope_launch<false>(a, place, e->stream, e->qkv_wstat);
}

static void k_attn(ntts_backbone* e, int i) {
    const ntts_backbone_config& c = e->cfg;
    AttnDecodeArgs a{};
    a.qkv = e->qkv_dec; a.ld_qkv = e->NQKV; a.out = e->attn_dec; a.ld_out = c.num_heads * e->HD;
    a.kpool = e->kv + (size_t)i * e->layer_stride; a.vpool = a.kpool + e->kv_half;
    a.block_table = e->block_table; a.max_pages = e->max_pages; a.pos = e->sl.pos; a.state = e->sl.state;
    a.rope_cos = e->rope_cos; a.rope_sin = e->rope_sin; a.nh = c.num_heads; a.nkv = c.num_kv_heads;
    a.tl = e->attn_tl;
    if (e->HD == 128) { attn_decode_launch_hd128(a, e->dec_rows, e->stream, c.max_context); return; }
    a.xcd_rows = ((e->xcd_affine & 4) && !e->attn_tl) ? e->xcd_xps : 0;
    a.nt_pages = e->wide && !e->attn_tl;
    if (e->fp8) a.out_fp8_inv = 1.0f / e->layers[i].xs[1];   // attention output = o_proj's input
    if (e->split_active && !e->attn_tl) {       // long contexts below 2 workgroups per CU: context-split attention + combine
        AttnSplitArgs q{};
        q.a = a; q.scores = e->as_scores; q.ld_scores = c.max_context + 16; q.stats = e->as_stats; q.oslabs = e->as_oslabs; q.nsplit = e->attn_split;
        q.a.slab_rows = e->dec_rows;
        attn_split_launch(q, e->dec_rows, e->stream, true);
        return;
    }
    attn_decode_launch(a, e->dec_rows, e->stream, c.max_context);
}

static void k_o_proj(ntts_backbone* e, int i) {
    const int B = e->dec_rows, H = e->H, QD = e->cfg.num_heads * e->HD;
    GemmArgs a = gemm_args(e, e->attn_dec, QD, e->layers[i].wo, QD, nullptr, e->slabs, H, B, H, QD, e->layers[i].so, e->layers[i].xs[1]);
    if (e->wide && e->wide_o == 1) {   // whole K per tile: h = bf16(h + bf16(acc)) in the epilogue (hf:models/qwen2/modeling_qwen2.py:233,291), in place
        a.out = e->h_dec; a.resid_bf16 = e->h_dec; a.ldrb = H;
        gemm_skinny<EPI_RESID>(a, 1, e->stream);
        return;
    }
    if (e->tall & 1) { a.xcd_nsplit = e->tall_xcd_split ? -1 : 0; gemm_tall<EPI_SPLITK>(a, e->ks_o, e->stream); return; }   // (one K slice per XCD pair: its X columns enter two L2s, not eight)
    if ((e->xcd_affine & 1) && e->xcd_xps) a.xcd_maffine = -1;
    gemm_skinny<EPI_SPLITK>(a, e->ks_o, e->stream);
}

static void k_gate_up(ntts_backbone* e, int i) {
    const int B = e->dec_rows, H = e->H, F = e->F;
    GemmArgs gu = gemm_args(e, e->xn_dec, H, e->layers[i].wgu, H, nullptr, e->act_dec, F, B, 2 * F, H, e->layers[i].sgu, e->layers[i].xs[2]);
    if (e->fp8) gu.out_fp8_inv = 1.0f / e->layers[i].xs[3];            // the activation is down_proj's input
    // (a 4-slot ring on the 128 x 128 tile, 96 KB in flight per CU instead of 64: 13.3 vs 13.4 us -- ring depth is not what
    //  bounds this kernel; profiles/r02h_sweep_gate_up_ring.log)
    // (natural-order gate/up tiles that use more CUs -- 128 x 80 as 244 workgroups of 4 or 8 waves, 128 x 96 as 204 -- measured
    //  15.5 / 13.7 / 14.1 vs 13.5 us and were removed: profiles/r02k_sweep_lpt_head_gu_tiles.log)
    if (e->gu_tile == 1) {          // 256 x 192, 12 waves
        if (e->fp8) gemm_launch<4, 3, 4, EPI_SILU_MUL, 2, 0, 64, false, true>(gu, 1, e->stream);
        else gemm_launch<4, 3, 4, EPI_SILU_MUL, 2>(gu, 1, e->stream);
    } else if (e->gu_tile == 2) {   // 256 x 256, 16 waves
        if (e->fp8) gemm_launch<4, 4, 4, EPI_SILU_MUL, 2, 0, 64, false, true>(gu, 1, e->stream);
        else gemm_launch<4, 4, 4, EPI_SILU_MUL, 2>(gu, 1, e->stream);
    } else if (e->gu_tile == 3) {   // 128 x 128, 8 waves, 2 slots
        if (e->fp8) gemm_launch<4, 2, 2, EPI_SILU_MUL, 2, 0, 64, false, true>(gu, 1, e->stream);
        else gemm_launch<4, 2, 2, EPI_SILU_MUL, 2>(gu, 1, e->stream);
    } else if (e->gu_128) {   // 128 x 128, 8 waves
        if (e->fp8) gemm_launch<4, 2, 2, EPI_SILU_MUL, 3, 0, 64, false, true>(gu, 1, e->stream);
        else gemm_launch<4, 2, 2, EPI_SILU_MUL, 3>(gu, 1, e->stream);
    } else gemm_skinny<EPI_SILU_MUL, 3>(gu, 1, e->stream);
}

static void k_down(ntts_backbone* e, int i) {
    const int B = e->dec_rows, H = e->H, F = e->F;
    GemmArgs a = gemm_args(e, e->act_dec, F, e->layers[i].wd, F, nullptr, e->slabs, H, B, H, F, e->layers[i].sd, e->layers[i].xs[3]);
    if (e->wide && e->wide_down == 1 && !e->fp8) {   // 128 x 128 / 8 waves / 3-slot ring: a W tile enters LDS once per 128 rows (64 x 64: 622 KB through a CU per tile); fp8: no gain (241.1 vs 241.3 k), kept narrow
        a.xcd_nsplit = -1;
        gemm_launch<4, 2, 2, EPI_SPLITK, 3>(a, e->ks_d, e->stream);
        return;
    }
    if (e->tall & 2) { a.xcd_nsplit = e->tall_xcd_split ? -1 : 0; gemm_tall<EPI_SPLITK>(a, e->ks_d, e->stream); return; }
    a.xcd_nsplit = -1;   // one K slice per XCD (pair) unless the row-block placement below applies (FETCH 15.0 -> 6.8 MB per launch, profiles/r02f_*)
    if ((e->xcd_affine & 2) && e->xcd_xps) a.xcd_maffine = -1;
    gemm_skinny<EPI_SPLITK>(a, e->ks_d, e->stream);
}

// residual += reduce(slabs of a K-deep split-K GEMM); normed = rmsnorm(residual) * norm_w
// next_scale: fp8 model, the static input scale of the GEMM that consumes the normalised rows (0 = bf16 output)
static void k_add_norm(ntts_backbone* e, int K, int ks, const bf16_t* norm_w, bf16_t* resid_out, bf16_t* normed_out, float next_scale = 0.f, int affine_bit = 0) {
    NormArgs n{};
    if (e->wide && e->wide_o == 1 && affine_bit == 1) {   // behind the wide o_proj: the residual stream is already summed (EPI_RESID), this pass only normalises it
        n.o_bf16 = e->h_dec; n.norm_w = norm_w; n.normed_out = normed_out; n.M = e->dec_rows; n.H = e->H; n.eps = e->cfg.rms_eps;
        if (e->fp8 && next_scale > 0.f) n.out_fp8_inv = 1.0f / next_scale;
        add_rmsnorm_launch(n, e->stream, true);
        return;
    }
    n.xcd_rows = (e->xcd_affine & affine_bit) ? e->xcd_xps : 0;
    n.slabs = e->slabs; n.nslab = gemm_nsplit(K, ks, ktile_of(e)); n.slab_rows = e->dec_rows; n.resid_in = e->h_dec; n.resid_out = resid_out;
    n.norm_w = norm_w; n.normed_out = normed_out; n.M = e->dec_rows; n.H = e->H; n.eps = e->cfg.rms_eps;
    if (e->fp8 && next_scale > 0.f) n.out_fp8_inv = 1.0f / next_scale;
    add_rmsnorm_launch(n, e->stream, true);   // one row per workgroup: 256 CUs pull the slabs instead of 64 (5.5 -> 4.0 us per launch)
}

// ---- small-batch step (gemv.h): per layer  [norm -> QKV]  attention  [o_proj]  [norm -> gate/up -> SiLU*mul]  [down]
static GemvArgs gemv_args(const ntts_backbone* e, const bf16_t* X, long ldx, const bf16_t* W, long ldw, void* out, long ldo, int N, int K,
                          const float* wscale = nullptr, float xscale = 1.f) {
    GemvArgs a{};
    a.wscale = wscale; a.xscale = xscale;
    a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.w_tile_major = 1; a.out = out; a.ldo = ldo;
    a.slab_rows = e->dec_rows; a.M = e->dec_rows; a.N = N; a.K = K;
    a.tl = e->gemv_tl;
    a.n_valid = N;
    return a;
}
// the fused prologue of layer i's QKV GEMV: h = (i == 0 ? embed[cur_tok] : h + bf16(sum of the previous down_proj's slabs));
// x = rmsnorm(h) * ln1.  The residual stream alternates between h_dec and h_alt (block 0 writes, every block reads).
static NormArgs pro_qkv(ntts_backbone* e, int i) {
    NormArgs n{};
    n.M = e->dec_rows; n.H = e->H; n.eps = e->cfg.rms_eps; n.norm_w = e->layers[i].ln1;
    if (i == 0) { n.gather_ids = e->sl.cur_tok; n.embed = e->embed; }
    else { n.slabs = e->slabs2; n.nslab = gemv_nsplit(e->F, ntts_backbone::kSksD, e->fp8); n.slab_rows = e->dec_rows; n.resid_in = e->h_dec; }
    n.resid_out = e->h_alt;
    if (e->fp8) n.out_fp8_inv = 1.0f / e->layers[i].xs[0];       // the panel holds the QKV GEMV's e4m3 input
    return n;
}
// QKV + bias + rounding + RoPE + K append in one GEMV launch (qkv_rope.h gemv_qkv_rope_kernel): batch 1 step 1.003 -> 0.962 ms together
// with the 8-wave prologue-free attention (profiles/r03c_sweep_b1_fused_qkv_attn_waves.log)
static void ks_qkv(ntts_backbone* e, int i) {
    GemvQkvArgs a{};
    a.pro = pro_qkv(e, i); a.W = e->layers[i].wqkv; a.bias = e->layers[i].bqkv; a.M = e->dec_rows; a.N = e->NQKV; a.K = e->H;
    if (e->fp8) { a.wscale = e->layers[i].sqkv; a.xscale = e->layers[i].xs[0]; }
    a.meta = e->step_meta; a.rope_rows = e->rope_rows; a.q_out = e->qkv_dec; a.ld_q = e->NQKV;
    a.kpool = e->kv + (size_t)i * e->layer_stride; a.nh = e->cfg.num_heads; a.nkv = e->cfg.num_kv_heads;
    a.tl = e->gemv_tl;
    gemv_qkv_rope_launch(a, e->stream);
}
static void ks_attn(ntts_backbone* e, int i) {
    const ntts_backbone_config& c = e->cfg;
    AttnDecodeArgs a{};
    a.qkv = e->qkv_dec; a.ld_qkv = e->NQKV; a.out = e->attn_dec; a.ld_out = c.num_heads * 64;
    a.slab_rows = e->dec_rows;
    a.kpool = e->kv + (size_t)i * e->layer_stride; a.vpool = a.kpool + e->kv_half;
    a.block_table = e->block_table; a.max_pages = e->max_pages; a.pos = e->sl.pos; a.state = e->sl.state;
    a.rope_cos = e->rope_cos; a.rope_sin = e->rope_sin; a.nh = c.num_heads; a.nkv = c.num_kv_heads;
    a.tl = e->attn_tl;
    if (e->fp8) a.out_fp8_inv = 1.0f / e->layers[i].xs[1];       // attention output = o_proj's e4m3 input
    if (e->split_active && !e->attn_tl) {
        AttnSplitArgs q{};
        q.a = a; q.scores = e->as_scores; q.ld_scores = c.max_context + 16; q.stats = e->as_stats; q.oslabs = e->as_oslabs; q.nsplit = e->attn_split;
        attn_split_launch(q, e->dec_rows, e->stream);
        return;
    }
    attn_decode_launch_small(a, e->dec_rows, e->stream);
}
static void ks_o_proj(ntts_backbone* e, int i) {
    const int QD = e->cfg.num_heads * 64;
    GemvArgs a = gemv_args(e, e->attn_dec, QD, e->layers[i].wo, QD, e->slabs, e->H, e->H, QD, e->layers[i].so, e->layers[i].xs[1]);
    if (e->split_active && !e->attn_tl) { a.xslabs = e->as_oslabs; a.n_xslab = e->attn_split; }   // context-split attention: chunk outputs summed here
    if (e->fp8) gemv_launch<EPI_SPLITK, false, 4, true>(a, ntts_backbone::kSksO, e->stream);
    else gemv_launch<EPI_SPLITK, false>(a, ntts_backbone::kSksO, e->stream);
}
static void ks_gate_up(ntts_backbone* e, int i) {
    GemvArgs a = gemv_args(e, nullptr, 0, e->layers[i].wgu, e->H, e->act_dec, e->F, 2 * e->F, e->H, e->layers[i].sgu, e->layers[i].xs[2]);
    NormArgs n{};
    n.M = e->dec_rows; n.H = e->H; n.eps = e->cfg.rms_eps; n.norm_w = e->layers[i].ln2;
    n.slabs = e->slabs; n.nslab = gemv_nsplit(e->cfg.num_heads * 64, ntts_backbone::kSksO, e->fp8); n.slab_rows = e->dec_rows;
    n.resid_in = e->h_alt; n.resid_out = e->h_dec;
    if (e->fp8) { n.out_fp8_inv = 1.0f / e->layers[i].xs[2]; a.out_fp8_inv = 1.0f / e->layers[i].xs[3]; }   // e4m3 panel in, e4m3 activation out (down_proj's input)
    a.pro = n;
    if (e->fp8) { gemv_launch<EPI_SILU_MUL, true, 3, true>(a, 1, e->stream); return; }
    gemv_launch<EPI_SILU_MUL, true, 3>(a, 1, e->stream);   // 3 feature waves: 203 workgroups, every CU streams <= 86 KB of weights (gemv.h)
}
static void ks_down(ntts_backbone* e, int i) {
    const GemvArgs a = gemv_args(e, e->act_dec, e->F, e->layers[i].wd, e->F, e->slabs2, e->H, e->H, e->F, e->layers[i].sd, e->layers[i].xs[3]);
    if (e->fp8) gemv_launch<EPI_SPLITK, false, 4, true>(a, ntts_backbone::kSksD, e->stream);
    else gemv_launch<EPI_SPLITK, false>(a, ntts_backbone::kSksD, e->stream);
}
static void ks_final_norm(ntts_backbone* e) {   // h += down (last layer); xn = rmsnorm(h) * final_norm  -> the lm_head's input
    NormArgs n{};
    n.slabs = e->slabs2; n.nslab = gemv_nsplit(e->F, ntts_backbone::kSksD, e->fp8); n.slab_rows = e->dec_rows; n.resid_in = e->h_dec; n.resid_out = e->h_dec;
    n.norm_w = e->final_norm; n.normed_out = e->xn_dec; n.M = e->dec_rows; n.H = e->H; n.eps = e->cfg.rms_eps;
    if (e->fp8) n.out_fp8_inv = 1.0f / e->xs_head;
    add_rmsnorm_launch(n, e->stream, true);
}
static void ks_lm_head(ntts_backbone* e, bool keep_logits) {
    const int H = e->H, V = e->cfg.vocab_size;
    GemvArgs a = gemv_args(e, e->xn_dec, H, e->lr_rows ? e->head_r : e->embed_tm, H, nullptr, 0, e->lr_rows ? (e->lr_rows + 15) / 16 * 16 : V, H,
                           e->lr_rows ? e->shead_r : e->shead, e->xs_head);
    if (e->lr_rows) { a.eos_col1 = e->lr_rows; a.n_valid = e->lr_rows; }   // compacted head: columns = [range | EOS | padding to 16]
    a.part_val = e->part_val; a.part_idx = e->part_idx; a.mask_eos = e->sl.mask_eos;
    a.logits = keep_logits ? e->logits : nullptr; a.ld_logits = V;
    a.logits_bf16 = (keep_logits && e->n_sampling > 0) ? e->logits_bf16 : nullptr; a.ld_logits_bf16 = e->ldl;
    if (e->fp8) { gemv_launch<EPI_ARGMAX, false, 4, true>(a, 1, e->stream); return; }
    gemv_launch<EPI_ARGMAX, false>(a, 1, e->stream);
}

static void decode_step_small(ntts_backbone* e) {
    k_step_meta(e);
    for (int i = 0; i < e->cfg.num_layers; ++i) {
        ks_qkv(e, i);
        ks_attn(e, i);
        ks_o_proj(e, i);
        ks_gate_up(e, i);
        ks_down(e, i);
    }
    ks_final_norm(e);
    lm_head_and_sample(e, SLOT_RUNNING);
}

static void decode_step(ntts_backbone* e) {
    if (e->small) { decode_step_small(e); return; }
    const ntts_backbone_config& c = e->cfg;
    const int B = e->dec_rows, H = e->H, F = e->F, QD = c.num_heads * e->HD;
    NormArgs n0{};
    n0.gather_ids = e->sl.cur_tok; n0.embed = e->embed; n0.resid_out = e->h_dec; n0.norm_w = e->layers[0].ln1;
    n0.normed_out = e->xn_dec; n0.M = B; n0.H = H; n0.eps = c.rms_eps;
    if (e->fp8) n0.out_fp8_inv = 1.0f / e->layers[0].xs[0];
    if (H > 512 && H <= 1024) {   // (the geometry add_rmsnorm_launch gives one row per workgroup) the step record rides in the same launch
        NTTS_LAUNCH((embed_norm_meta_kernel), dim3(B), dim3(128), e->stream, n0, step_meta_args(e));
    } else {
        add_rmsnorm_launch(n0, e->stream, true);
        k_step_meta(e);
    }
    for (int i = 0; i < c.num_layers; ++i) {
        const bool last = i + 1 == c.num_layers;
        k_qkv(e, i);
        k_attn(e, i);
        k_o_proj(e, i);
        k_add_norm(e, QD, e->ks_o, e->layers[i].ln2, e->h_dec, e->xn_dec, e->layers[i].xs[2], 1);
        k_gate_up(e, i);
        k_down(e, i);
        k_add_norm(e, F, e->ks_d, last ? e->final_norm : e->layers[i + 1].ln1, e->h_dec, e->xn_dec, last ? e->xs_head : e->layers[i + 1].xs[0], 2);
    }
    lm_head_and_sample(e, SLOT_RUNNING);
}

static int alloc_pages(ntts_backbone* e, HostSlot& s, int tokens) {
    const int need = (tokens + kPage - 1) / kPage;
    while ((int)s.pages.size() < need) {
        if (e->free_pages.empty()) return NTTS_ENOMEM;
        const int pg = e->free_pages.back();
        e->free_pages.pop_back();
        e->page_ref[pg] = 1;
        s.pages.push_back(pg);
    }
    return NTTS_OK;
}
// give back the slot's pages from index `keep` on; a page returns to the pool when its last owner lets go
static void drop_pages(ntts_backbone* e, HostSlot& s, size_t keep = 0) {
    while (s.pages.size() > keep) {
        const int pg = s.pages.back();
        s.pages.pop_back();
        if (--e->page_ref[pg] == 0) e->free_pages.push_back(pg);
    }
}

// meta block -> device, asynchronously on `st`: through the next slot of the page-locked ring (waits only if the copy that last
// used that slot -- four uploads ago -- has not executed yet)
static hipError_t upload_meta(ntts_backbone* e, const int* src, size_t n, hipStream_t st) {
    if (n > e->meta_cap) return hipErrorInvalidValue;   // (every caller checks first; the ring slots hold meta_cap ints)
    if (n <= e->small_cap && e->small_host) {            // a scheduler's small upload: the deep ring (the device copies execute in stream order either way)
        const int k = e->small_next;
        e->small_next = (k + 1) % ntts_backbone::kSmallStages;
        if (e->small_used[k]) { const hipError_t rc = hipEventSynchronize(e->small_ev[k]); if (rc != hipSuccess) return rc; }
        int* h = e->small_host + (size_t)k * e->small_cap;
        memcpy(h, src, n * sizeof(int));
        hipError_t rc = hipMemcpyAsync(e->meta_dev, h, n * sizeof(int), hipMemcpyHostToDevice, st);
        if (rc != hipSuccess) return rc;
        e->small_used[k] = true;
        return hipEventRecord(e->small_ev[k], st);
    }
    const int k = e->meta_next;
    e->meta_next = (k + 1) % ntts_backbone::kMetaStages;
    if (e->meta_used[k]) { const hipError_t rc = hipEventSynchronize(e->meta_ev[k]); if (rc != hipSuccess) return rc; }
    memcpy(e->meta_host[k], src, n * sizeof(int));
    hipError_t rc = hipMemcpyAsync(e->meta_dev, e->meta_host[k], n * sizeof(int), hipMemcpyHostToDevice, st);
    if (rc != hipSuccess) return rc;
    e->meta_used[k] = true;
    return hipEventRecord(e->meta_ev[k], st);
}

// Prompt pass.  With donor_slot / shared_len (ntts_backbone_prefill_shared): prompt i re-uses the KV pages that hold the
// first pos0[i] = floor(shared_len[i] / 32) * 32 tokens of its donor's prompt -- only the remaining tokens are packed,
// embedded and pushed through the layers; their queries attend to the shared pages exactly as they would to their own.
static int prefill_impl(ntts_backbone* e, int32_t n, const int32_t* ids, const int32_t* lens, const int32_t* slots,
                        const ntts_sampling* samp, const int32_t* donor_slot, const int32_t* shared_len) {
    if (!e || n < 1 || !ids || !lens || !slots || !samp) return fail(e, NTTS_EINVAL, "null/empty argument");
    if ((donor_slot == nullptr) != (shared_len == nullptr)) return fail(e, NTTS_EINVAL, "donor_slot and shared_len go together");
    if (!e->finalized) return fail(e, NTTS_ESTATE, "weights not finalised");
    HIPCHK(e, hipSetDevice(e->device));
    const ntts_backbone_config& c = e->cfg;
    const int B = c.max_batch, H = e->H, F = e->F, QD = c.num_heads * e->HD;
    long T = 0, Tfull = 0;
    std::vector<int> pos0(n, 0), id_off(n, 0);
    for (int i = 0; i < n; ++i) { id_off[i] = (int)Tfull; Tfull += lens[i] > 0 ? lens[i] : 0; }
    for (int i = 0; i < n; ++i) {
        if (slots[i] < 0 || slots[i] >= B) return fail(e, NTTS_EINVAL, "slot %d out of range", slots[i]);
        if (e->slots[slots[i]].state != SLOT_FREE) return fail(e, NTTS_ESTATE, "slot %d is busy", slots[i]);
        for (int j = 0; j < i; ++j)
            if (slots[j] == slots[i]) return fail(e, NTTS_EINVAL, "slot %d given twice", slots[i]);
        if (lens[i] < 1) return fail(e, NTTS_EINVAL, "empty prompt %d", i);
        if (samp[i].max_length > c.max_context || samp[i].max_length <= lens[i])
            return fail(e, NTTS_EINVAL, "prompt %d: need len < max_length <= max_context (%d, %d)", i, lens[i], samp[i].max_length);
        if (samp[i].do_sample && (samp[i].top_k < 1 || !(samp[i].temperature > 0.f)))
            return fail(e, NTTS_EINVAL, "prompt %d: do_sample needs top_k >= 1 and temperature > 0 (got %d, %g)", i, samp[i].top_k, samp[i].temperature);
        if (samp[i].eos_token_id < 0 || samp[i].eos_token_id >= c.vocab_size) return fail(e, NTTS_EINVAL, "eos id out of range");
        if (e->lr_rows && samp[i].eos_token_id != e->lr_eos)
            return fail(e, NTTS_EINVAL, "prompt %d: eos id %d, but the restricted lm_head was set up for eos id %d (ntts_backbone_set_logits_range)", i, samp[i].eos_token_id, e->lr_eos);
        if (donor_slot && donor_slot[i] >= 0) {
            // the donor is a running slot, or a prompt given EARLIER in this call (its pages are filled by the same
            // launches: the rope/KV-write kernel of a layer completes before that layer's attention kernel starts)
            const int d = donor_slot[i];
            if (d >= B) return fail(e, NTTS_EINVAL, "prompt %d: donor slot %d out of range", i, d);
            const int* dp = nullptr;
            int dlen = 0;
            for (int j = 0; j < i; ++j)
                if (slots[j] == d) { dp = ids + id_off[j]; dlen = lens[j]; }
            if (!dp) {
                if (e->slots[d].state != SLOT_RUNNING) return fail(e, NTTS_ESTATE, "prompt %d: donor slot %d holds no prompt", i, d);
                dp = e->slots[d].prompt.data();
                dlen = (int)e->slots[d].prompt.size();
            }
            int sl = shared_len[i];
            if (sl < 0) return fail(e, NTTS_EINVAL, "prompt %d: negative shared_len", i);
            if (sl > dlen) sl = dlen;
            if (sl > lens[i] - 1) sl = lens[i] - 1;          // the last position is always computed (it yields the logits)
            for (int t = 0; t < sl; ++t)
                if (dp[t] != ids[id_off[i] + t])
                    return fail(e, NTTS_EINVAL, "prompt %d does not start with the first %d tokens of slot %d's prompt (token %d differs)", i, sl, d, t);
            pos0[i] = sl / kPage * kPage;                       // whole pages only: a page is never written by two slots
        }
        T += lens[i] - pos0[i];
    }
    if (T > e->Tmax) return fail(e, NTTS_EINVAL, "%ld prompt tokens exceed max_prefill_tokens %d", T, e->Tmax);
    for (long t = 0; t < Tfull; ++t)
        if (ids[t] < 0 || ids[t] >= c.vocab_size) return fail(e, NTTS_EINVAL, "token id %d out of range", ids[t]);

    // ---- pages (roll back on exhaustion)
    for (int i = 0; i < n; ++i) {
        HostSlot& s = e->slots[slots[i]];
        if (pos0[i] > 0) {                                      // borrow the donor's leading pages
            const HostSlot& d = e->slots[donor_slot[i]];
            for (int k = 0; k < pos0[i] / kPage; ++k) { s.pages.push_back(d.pages[k]); e->page_ref[d.pages[k]]++; }
        }
        if (alloc_pages(e, s, lens[i]) != NTTS_OK) {
            for (int j = 0; j <= i; ++j) drop_pages(e, e->slots[slots[j]]);
            return fail(e, NTTS_ENOMEM, "KV page pool exhausted (%d pages)", e->num_pages);
        }
    }
    // ---- meta block: [ids T][tok_seq T][tok_base n][seq_len n][slot n][min_new n][max_len n][eos n][last_row n]
    //                  [tile_seq nt][tile_q0 nt][bt_rows n*max_pages]
    std::vector<int> tile_seq, tile_q0, rtile_seq, rtile_q0, rtile_key, dtile_seq, dtile_q0, dtile_key;
    std::vector<int> m;
    m.reserve(2 * T + 16 * n);
    for (int i = 0; i < n; ++i) m.insert(m.end(), ids + id_off[i] + pos0[i], ids + id_off[i] + lens[i]);   // packed: new tokens only
    const size_t o_tok_seq = m.size();
    for (int i = 0; i < n; ++i) m.insert(m.end(), lens[i] - pos0[i], i);
    const size_t o_base = m.size();
    long acc = 0;
    for (int i = 0; i < n; ++i) {
        m.push_back((int)acc);
        // attention work lists, split by POSITION into three tiers (attn_prefill.h): queries below pf_res_cap (512) go to the resident kernel, those
        // below pf_deep_cap (1024) to the deep one -- both take work items (prompt, k) of 256 queries whose 16-query blocks the kernel deals out from
        // both ends of the tier -- the rest to the two-sweep kernel in 64-query tiles.  Which kernel computes a query depends on nothing but its position
        const int cap = e->pf_res_cap, dcap = e->pf_deep_cap;
        auto items = [&](int lo, int hi, std::vector<int>& seq, std::vector<int>& q0, std::vector<int>& key) {
            const int b0 = pos0[i] > lo ? pos0[i] : lo, a_end = lens[i] < hi ? lens[i] : hi;
            if (a_end <= b0) return;
            const int nb = (a_end - b0 + 15) / 16, nwg = (nb + 15) / 16;
            for (int k = 0; k < nwg; ++k) { seq.push_back(i); q0.push_back(k); key.push_back(nb); }
        };
        items(0, cap, rtile_seq, rtile_q0, rtile_key);
        items(cap, dcap, dtile_seq, dtile_q0, dtile_key);
        for (int q = pos0[i] > dcap ? pos0[i] : dcap; q < lens[i]; q += 64) { tile_seq.push_back(i); tile_q0.push_back(q); }
        acc += lens[i] - pos0[i];
    }
    // Causal attention: a 64-query tile that starts at position q0 sweeps (q0 + 64) / 32 KV pages, 2 .. 16 for a 500-token
    // prompt.  In prompt order the LAST workgroups dispatched are the deepest tiles of the last prompt and the pass ends on
    // them; sorted by descending depth (stable: ties keep prompt order) the shallow tiles fill the tail instead.
    auto deepest_first = [](std::vector<int>& seq, std::vector<int>& q0, const std::vector<int>& key) {
        std::vector<int> ord(seq.size());
        for (size_t k = 0; k < ord.size(); ++k) ord[k] = (int)k;
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key[a] > key[b]; });
        std::vector<int> ts(ord.size()), tq(ord.size());
        for (size_t k = 0; k < ord.size(); ++k) { ts[k] = seq[ord[k]]; tq[k] = q0[ord[k]]; }
        seq.swap(ts); q0.swap(tq);
    };
    deepest_first(tile_seq, tile_q0, std::vector<int>(tile_q0));
    deepest_first(rtile_seq, rtile_q0, rtile_key);   // (the work items of one prompt weigh the same: longest prompts first)
    deepest_first(dtile_seq, dtile_q0, dtile_key);
    const size_t o_len = m.size();   m.insert(m.end(), lens, lens + n);
    const size_t o_pos0 = m.size();  m.insert(m.end(), pos0.begin(), pos0.end());
    const size_t o_slot = m.size();  m.insert(m.end(), slots, slots + n);
    const size_t o_min = m.size();   for (int i = 0; i < n; ++i) m.push_back(samp[i].min_new_tokens);
    const size_t o_max = m.size();   for (int i = 0; i < n; ++i) m.push_back(samp[i].max_length);
    const size_t o_eos = m.size();   for (int i = 0; i < n; ++i) m.push_back(samp[i].eos_token_id);
    const size_t o_topk = m.size();  for (int i = 0; i < n; ++i) m.push_back(samp[i].do_sample ? samp[i].top_k : 0);
    const size_t o_temp = m.size();
    for (int i = 0; i < n; ++i) { int b; const float t = samp[i].do_sample ? samp[i].temperature : 1.0f; memcpy(&b, &t, 4); m.push_back(b); }
    const size_t o_seed = m.size();
    for (int i = 0; i < n; ++i) { m.push_back((int)(uint32_t)samp[i].seed); m.push_back((int)(uint32_t)(samp[i].seed >> 32)); }
    const size_t o_last = m.size();
    acc = 0;
    for (int i = 0; i < n; ++i) { acc += lens[i] - pos0[i]; m.push_back((int)acc - 1); }
    const size_t o_tseq = m.size();  m.insert(m.end(), tile_seq.begin(), tile_seq.end());
    const size_t o_tq0 = m.size();   m.insert(m.end(), tile_q0.begin(), tile_q0.end());
    const size_t o_rtseq = m.size(); m.insert(m.end(), rtile_seq.begin(), rtile_seq.end());
    const size_t o_rtq0 = m.size();  m.insert(m.end(), rtile_q0.begin(), rtile_q0.end());
    const size_t o_dtseq = m.size(); m.insert(m.end(), dtile_seq.begin(), dtile_seq.end());
    const size_t o_dtq0 = m.size();  m.insert(m.end(), dtile_q0.begin(), dtile_q0.end());
    // work lists of the LAST layer's attention: the one tile / work item per prompt that holds its last position (same split by position)
    std::vector<int> lt_seq, lt_q0, lrt_seq, lrt_q0, ldt_seq, ldt_q0;
    for (int i = 0; i < n; ++i) {
        const int last = lens[i] - 1, cap = e->pf_res_cap, dcap = e->pf_deep_cap;
        // the work item that holds a tier's last 16-query block: blocks below nbp / 2 are "lo" blocks of item b / 8, the others "hi" blocks
        auto last_item = [&](int lo, int hi) {
            const int b0 = pos0[i] > lo ? pos0[i] : lo, a_end = lens[i] < hi ? lens[i] : hi;
            const int nb = (a_end - b0 + 15) / 16, nbp = (nb + 15) / 16 * 16, b = nb - 1;
            return b < nbp / 2 ? b / 8 : (nbp - 1 - b) / 8;
        };
        if (last < cap) { lrt_seq.push_back(i); lrt_q0.push_back(last_item(0, cap)); }
        else if (last < dcap) { ldt_seq.push_back(i); ldt_q0.push_back(last_item(cap, dcap)); }
        else { const int b0 = pos0[i] > dcap ? pos0[i] : dcap; lt_seq.push_back(i); lt_q0.push_back(b0 + (last - b0) / 64 * 64); }
    }
    const size_t o_ltseq = m.size();  m.insert(m.end(), lt_seq.begin(), lt_seq.end());
    const size_t o_ltq0 = m.size();   m.insert(m.end(), lt_q0.begin(), lt_q0.end());
    const size_t o_lrtseq = m.size(); m.insert(m.end(), lrt_seq.begin(), lrt_seq.end());
    const size_t o_lrtq0 = m.size();  m.insert(m.end(), lrt_q0.begin(), lrt_q0.end());
    const size_t o_ldtseq = m.size(); m.insert(m.end(), ldt_seq.begin(), ldt_seq.end());
    const size_t o_ldtq0 = m.size();  m.insert(m.end(), ldt_q0.begin(), ldt_q0.end());
    const size_t o_bt = m.size();
    for (int i = 0; i < n; ++i) {
        const HostSlot& s = e->slots[slots[i]];
        for (int k = 0; k < e->max_pages; ++k) m.push_back(k < (int)s.pages.size() ? s.pages[k] : 0);
    }
    if (m.size() > e->meta_cap) {
        for (int i = 0; i < n; ++i) drop_pages(e, e->slots[slots[i]]);
        return fail(e, NTTS_EINVAL, "prefill meta block too large");
    }
    {   // sampling requests need the bf16 logits rows: allocate on first use; the captured decode step bakes the pointer
        int add = 0;
        for (int i = 0; i < n; ++i) add += samp[i].do_sample ? 1 : 0;
        if (add && !e->logits_bf16) {
            HIPCHK(e, hipStreamSynchronize(e->stream));
            HIPCHK(e, hipMalloc((void**)&e->logits_bf16, (size_t)B * e->ldl * sizeof(bf16_t)));
            HIPCHK(e, hipMemset(e->logits_bf16, 0, (size_t)B * e->ldl * sizeof(bf16_t)));
        }
        for (int i = 0; i < n; ++i) e->slots[slots[i]].sampling = samp[i].do_sample != 0;
        e->n_sampling += add;
    }
    // With a side stream the whole pass (meta upload included) runs there, ordered behind the work already on the engine's
    // stream and followed by that stream; every launch helper reads e->stream, so it is swapped for the duration of the call.
    // If the ordering events cannot be recorded the pass stays on the engine's own stream (correct, merely not CU-masked);
    // if the closing event fails the side stream is drained on the host before the engine's stream goes on.
    struct SideStream {
        ntts_backbone* e; hipStream_t main; bool on = false;
        explicit SideStream(ntts_backbone* e_) : e(e_), main(e_->stream) {
            if (!e->pf_stream || !e->pf_ev[0] || !e->pf_ev[1]) return;
            if (hipEventRecord(e->pf_ev[0], main) != hipSuccess || hipStreamWaitEvent(e->pf_stream, e->pf_ev[0], 0) != hipSuccess) {
                (void)hipGetLastError();
                return;
            }
            e->stream = e->pf_stream;
            on = true;
        }
        ~SideStream() {
            if (!on) return;
            if (hipEventRecord(e->pf_ev[1], e->pf_stream) != hipSuccess || hipStreamWaitEvent(main, e->pf_ev[1], 0) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipStreamSynchronize(e->pf_stream);
            }
            e->stream = main;
        }
    } side(e);
    hipStream_t st = e->stream;
    HIPCHK(e, upload_meta(e, m.data(), m.size(), st));
    const int* md = e->meta_dev;
    PrefillMeta meta{md + o_base, md + o_len, md + o_pos0, md + o_slot, md + o_tok_seq, md + o_tseq, md + o_tq0};

    HIPCHK(e, hipEventRecord(e->ev[0], st));
    PrefillInit pi{};
    pi.slot = md + o_slot; pi.seq_len = md + o_len; pi.min_new = md + o_min; pi.max_len = md + o_max; pi.eos = md + o_eos;
    pi.top_k = md + o_topk; pi.temp_bits = md + o_temp; pi.seed = md + o_seed;
    pi.bt_rows = md + o_bt; pi.block_table = e->block_table; pi.max_pages = e->max_pages; pi.n = n; pi.sl = e->sl;
    NTTS_LAUNCH((prefill_init_kernel), dim3(n), dim3(64), st, pi);

    const int Ti = (int)T;
    NormArgs n0{};
    n0.gather_ids = md; n0.embed = e->embed; n0.resid_out = e->h_pf; n0.norm_w = e->layers[0].ln1; n0.normed_out = e->xn_pf;
    n0.M = Ti; n0.H = H; n0.eps = c.rms_eps;
    if (e->fp8) n0.out_fp8_inv = 1.0f / e->layers[0].xs[0];
    add_rmsnorm_launch(n0, st);
    // fp8 calibration (ntts_backbone_calibrate, bf16 engines): running max |x| of every GEMM's input rows, slot = 4 * layer + {0: QKV, 1: o_proj,
    // 2: gate/up, 3: down_proj}, last slot = lm_head.  One small launch behind each producer, in calibration mode only.
    auto tap = [&](const bf16_t* x, long rows, int cols, int slot) {   // x: DENSE rows (leading dimension == cols, as every workspace tapped below is)
        if (!e->calib || rows <= 0) return;
        const long nvec = rows * cols / 8;
        float* dst = e->calib + slot;
        NTTS_LAUNCH((amax_bf16_kernel), dim3((unsigned)std::min<long>((nvec + 255) / 256, 1024)), dim3(256), st, x, nvec, dst);
    };
    for (int i = 0; i < c.num_layers; ++i) {
        const LayerW& w = e->layers[i];
        const bool last = i + 1 == c.num_layers;
        tap(e->xn_pf, Ti, H, 4 * i);
        gemm_large<EPI_BF16>(e, gemm_args(e, e->xn_pf, H, w.wqkv, H, w.bqkv, e->qkv_pf, e->NQKV, Ti, e->NQKV, H, w.sqkv, w.xs[0]), st);
        RopeWriteArgs r{};
        r.qkv = e->qkv_pf; r.ld_qkv = e->NQKV; r.kpool = e->kv + (size_t)i * e->layer_stride; r.vpool = r.kpool + e->kv_half;
        r.block_table = e->block_table; r.max_pages = e->max_pages; r.meta = meta; r.rope_cos = e->rope_cos; r.rope_sin = e->rope_sin;
        r.nh = c.num_heads; r.nkv = c.num_kv_heads; r.T = Ti;
        // the q heads are rotated by the attention kernel as it loads them (one read + one write of T x 896 values less per layer:
        // prompt pass 28.16 -> 27.94 ms per chunk, profiles/r02k_sweep_pf_rope_q_fused.log); this kernel rotates k and scatters v
        r.skip_q = 1;
        if (e->generic) {   // head_dim 128 and / or qk-norm: q AND k normalised + rotated here (q in place), v scattered (attn_prefill.h)
            RopeNormArgs g{};
            g.qkv = e->qkv_pf; g.ld_qkv = e->NQKV; g.kpool = r.kpool; g.vpool = r.vpool; g.block_table = e->block_table; g.max_pages = e->max_pages;
            g.meta = meta; g.rope_cos = e->rope_cos; g.rope_sin = e->rope_sin; g.q_norm = w.qn; g.k_norm = w.kn; g.eps = c.rms_eps;
            g.nh = c.num_heads; g.nkv = c.num_kv_heads; g.rows = Ti; g.write_v = 1;
            const long items = (long)Ti * (c.num_heads + 2 * c.num_kv_heads);
            if (e->HD == 128) NTTS_LAUNCH((rope_norm_kv_write_kernel<128>), dim3((unsigned)((items + 3) / 4)), dim3(256), st, g);
            else NTTS_LAUNCH((rope_norm_kv_write_kernel<64>), dim3((unsigned)((items + 3) / 4)), dim3(256), st, g);
        } else
        NTTS_LAUNCH((rope_kv_write_vec_kernel), dim3((Ti + kRopeTokPerBlock - 1) / kRopeTokPerBlock), dim3(256), st, r);
        AttnPrefillArgs a{};
        a.qkv = e->qkv_pf; a.ld_qkv = e->NQKV; a.out = e->attn_pf; a.ld_out = QD; a.kpool = r.kpool; a.vpool = r.vpool;
        a.block_table = e->block_table; a.max_pages = e->max_pages; a.meta = meta; a.nh = c.num_heads; a.nkv = c.num_kv_heads;
        if (e->fp8) a.out_fp8_inv = 1.0f / w.xs[1];     // attn_pf rows hold QD e4m3 BYTES (ld_out counts bytes then)
        a.rope_cos = e->rope_cos; a.rope_sin = e->rope_sin;
        if (e->generic) a.rope_cos = a.rope_sin = nullptr;   // (q is already normalised and rotated)
        // Last layer: the KV pages are complete after the rope/KV-write above, and nothing but each prompt's LAST position
        // is read afterwards (it alone feeds the lm_head).  Attention runs on the one query tile per prompt that holds
        // it, then that row and its residual row are compacted and o_proj / the MLP run on n rows instead of T.
        // Row-wise results are unchanged (every GEMM / norm row is computed from that row's operands alone).
        const bool prune = last && T >= 4L * n && !e->calib;    // (calibration mode looks at EVERY position's GEMM inputs, the last layer's included)
        int n_tiles = (int)tile_seq.size(), n_rtiles = (int)rtile_seq.size();
        if (prune) { a.meta.tile_seq = md + o_ltseq; a.meta.tile_q0 = md + o_ltq0; n_tiles = (int)lt_seq.size(); }
        if (n_tiles) { if (e->HD == 128) attn_prefill_launch_hd128(a, n_tiles, st); else attn_prefill_launch(a, n_tiles, st); }
        a.meta.tile_seq = md + o_rtseq; a.meta.tile_q0 = md + o_rtq0;
        if (prune) { a.meta.tile_seq = md + o_lrtseq; a.meta.tile_q0 = md + o_lrtq0; n_rtiles = (int)lrt_seq.size(); }
        if (n_rtiles) attn_prefill_res_launch(a, n_rtiles, e->pf_res_cap, prune, st);
        int n_dtiles = (int)dtile_seq.size();
        a.meta.tile_seq = md + o_dtseq; a.meta.tile_q0 = md + o_dtq0;
        if (prune) { a.meta.tile_seq = md + o_ldtseq; a.meta.tile_q0 = md + o_ldtq0; n_dtiles = (int)ldt_seq.size(); }
        if (n_dtiles) attn_prefill_deep_launch(a, n_dtiles, e->pf_res_cap, e->pf_deep_cap, prune, st);
        const int Mi = prune ? n : Ti;                       // rows from here on
        const bf16_t* attn_in = e->attn_pf;
        bf16_t* hres = e->h_pf;
        if (prune) {   // qkv_pf is free once attention has run: T * NQKV >= 4 n * NQKV > n * (QD + H) elements
            bf16_t* attn_c = e->qkv_pf;
            bf16_t* h_c = e->qkv_pf + (size_t)n * QD;
            const int qe = e->fp8 ? QD / 2 : QD;   // attention rows in bf16 units (fp8: QD bytes)
            NTTS_LAUNCH((gather_rows_kernel), dim3(n), dim3(256), st, (const bf16_t*)e->attn_pf, (long)qe, (const int*)(md + o_last), attn_c, (long)qe, qe);
            NTTS_LAUNCH((gather_rows_kernel), dim3(n), dim3(256), st, (const bf16_t*)e->h_pf, (long)H, (const int*)(md + o_last), h_c, (long)H, H);
            attn_in = attn_c;
            hres = h_c;
        }
        // the residual add rides in the o_proj / down_proj epilogue (EPI_RESID: h = bf16(h + bf16(acc)), in place -- the same two
        // roundings the norm kernel applied), so the norm pass reads one row stream instead of two and writes one
        // (131.8 -> 128.3 ms per batch, profiles/r02i_ab_prefill_resid_epilogue.jsonl)
        NormArgs n1{};
        tap(attn_in, Mi, QD, 4 * i + 1);
        {
            GemmArgs ao = gemm_args(e, attn_in, QD, w.wo, QD, nullptr, hres, H, Mi, H, QD, w.so, w.xs[1]);
            ao.resid_bf16 = hres; ao.ldrb = H;
            gemm_large<EPI_RESID>(e, ao, st);
            n1.o_bf16 = hres;
        }
        n1.norm_w = w.ln2; n1.normed_out = e->xn_pf;
        n1.M = Mi; n1.H = H; n1.eps = c.rms_eps;
        if (e->fp8) n1.out_fp8_inv = 1.0f / w.xs[2];
        add_rmsnorm_launch(n1, st);
        tap(e->xn_pf, Mi, H, 4 * i + 2);
        GemmArgs gu = gemm_args(e, e->xn_pf, H, w.wgu, H, nullptr, e->act_pf, F, Mi, 2 * F, H, w.sgu, w.xs[2]);
        if (e->fp8) gu.out_fp8_inv = 1.0f / w.xs[3];
        gemm_large<EPI_SILU_MUL>(e, gu, st);
        tap(e->act_pf, Mi, F, 4 * i + 3);
        NormArgs n2{};
        {
            GemmArgs ad = gemm_args(e, e->act_pf, F, w.wd, F, nullptr, hres, H, Mi, H, F, w.sd, w.xs[3]);
            ad.resid_bf16 = hres; ad.ldrb = H;
            gemm_large<EPI_RESID>(e, ad, st);
            n2.o_bf16 = hres;
        }
        n2.eps = c.rms_eps; n2.H = H;
        if (!last) {
            n2.norm_w = e->layers[i + 1].ln1; n2.normed_out = e->xn_pf; n2.M = Ti;
            if (e->fp8) n2.out_fp8_inv = 1.0f / e->layers[i + 1].xs[0];
        } else {  // only each prompt's last position feeds the lm_head: gather it into its decode-slot row
            n2.resid_out = e->h_dec; n2.norm_w = e->final_norm; n2.normed_out = e->xn_dec; n2.M = n;
            n2.in_rows = prune ? nullptr : md + o_last; n2.out_rows = md + o_slot;
            if (e->fp8) n2.out_fp8_inv = 1.0f / e->xs_head;
        }
        add_rmsnorm_launch(n2, st);
    }
    if (e->calib) {        // the lm_head's input: the final norm of EVERY position (the decode steps feed the head one row per step and
        NormArgs nc{};     // sequence -- a record of the prompts' last positions only would be 64 rows for 64 calibration prompts); scratch output
        nc.o_bf16 = e->h_pf; nc.norm_w = e->final_norm;   // (no pruning in calibration mode: h_pf holds every row)
        nc.norm_w = e->final_norm; nc.normed_out = e->xn_pf; nc.M = Ti; nc.H = H; nc.eps = c.rms_eps;
        add_rmsnorm_launch(nc, st);
        tap(e->xn_pf, Ti, H, 4 * c.num_layers);
    }
    lm_head_and_sample(e, SLOT_PREFILLED);
    HIPCHK(e, hipEventRecord(e->ev[1], st));
    e->have_pf_time = true;
    HIPCHK(e, hipGetLastError());
    for (int i = 0; i < n; ++i) {
        HostSlot& s = e->slots[slots[i]];
        s.state = SLOT_RUNNING; s.prompt_len = lens[i]; s.max_len = samp[i].max_length; s.pos_upper = lens[i];
        s.gen++;
        s.prompt.assign(ids + id_off[i], ids + id_off[i] + lens[i]);
    }
    e->pf_tokens_computed += T;
    e->pf_tokens_shared += Tfull - T;
    return NTTS_OK;
}

extern "C" int ntts_backbone_prefill(ntts_backbone* e, int32_t n, const int32_t* ids, const int32_t* lens,
                                     const int32_t* slots, const ntts_sampling* samp) {
    return prefill_impl(e, n, ids, lens, slots, samp, nullptr, nullptr);
}

extern "C" int ntts_backbone_prefill_shared(ntts_backbone* e, int32_t n, const int32_t* ids, const int32_t* lens,
                                            const int32_t* slots, const ntts_sampling* samp, const int32_t* donor_slot,
                                            const int32_t* shared_len) {
    if (!donor_slot || !shared_len) return fail(e, NTTS_EINVAL, "null argument");
    return prefill_impl(e, n, ids, lens, slots, samp, donor_slot, shared_len);
}

extern "C" int ntts_backbone_kv_stats(ntts_backbone* e, int32_t* free_pages, int32_t* total_pages, int64_t* prompt_tokens_computed,
                                      int64_t* prompt_tokens_shared) {
    if (!e) return NTTS_EINVAL;
    if (free_pages) *free_pages = (int32_t)e->free_pages.size();
    if (total_pages) *total_pages = e->num_pages;
    if (prompt_tokens_computed) *prompt_tokens_computed = e->pf_tokens_computed;
    if (prompt_tokens_shared) *prompt_tokens_shared = e->pf_tokens_shared;
    return NTTS_OK;
}

extern "C" int ntts_backbone_decode(ntts_backbone* e, int32_t n_steps) {
    if (!e || n_steps < 1) return fail(e, NTTS_EINVAL, "bad n_steps");
    if (!e->finalized) return fail(e, NTTS_ESTATE, "weights not finalised");
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = e->stream;
    {   // how many chains share the chip right now decides the step's tiles (both shapes' graphs are kept)
        const int counted = chains_now(e);
        use_shape_for(e, e->gang_forced > 0 ? e->gang_forced : counted);
    }
    // ---- reserve KV pages for the positions these steps can write (pos <= max_len - 2)
    std::vector<int> trip;
    std::vector<std::pair<int, size_t>> undo;
    int ctx_now = 0;                                             // longest context a running slot may have at the first of these steps
    for (int b = 0; b < e->dec_rows; ++b) {                      // (parked slots do not decode: nothing to reserve for them)
        HostSlot& s = e->slots[b];
        if (s.state != SLOT_RUNNING) continue;
        if (s.pos_upper > ctx_now) ctx_now = s.pos_upper;
        int upto = s.pos_upper + n_steps;
        if (upto > s.max_len - 1) upto = s.max_len - 1;
        const size_t before = s.pages.size();
        if (alloc_pages(e, s, upto) != NTTS_OK) {
            undo.emplace_back(b, before);
            for (auto& u : undo) drop_pages(e, e->slots[u.first], u.second);
            return fail(e, NTTS_ENOMEM, "KV page pool exhausted (%d pages)", e->num_pages);
        }
        undo.emplace_back(b, before);
        for (size_t k = before; k < s.pages.size(); ++k) { trip.push_back(b); trip.push_back((int)k); trip.push_back(s.pages[k]); }
        s.pos_upper = upto;
    }
    if (!trip.empty()) {
        if (trip.size() > e->meta_cap) return fail(e, NTTS_EINVAL, "block-table update too large");
        HIPCHK(e, upload_meta(e, trip.data(), trip.size(), st));
        const int nt = (int)trip.size() / 3;
        NTTS_LAUNCH((bt_update_kernel), dim3((nt + 63) / 64), dim3(64), st, (const int*)e->meta_dev, nt, e->block_table, e->max_pages);
    }
    if ((e->graph || e->graph_split || e->graph_shape[0] || e->graph_shape[1]) && e->graph_has_logits != (e->n_sampling > 0))   // the step's launch arguments changed:
        drop_graphs(e);                                                                                                            // every capture is stale
    // small-batch path: steps whose longest context has reached attn_split_ctx run the context-split attention (its own graph)
    const bool can_split = e->attn_split > 0 && !e->attn_tl;
    auto wants_split = [&](int step) { return can_split && ctx_now + step >= e->attn_split_ctx; };
    auto capture = [&](int steps, hipGraphExec_t* out) {
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            for (int k = 0; k < steps; ++k) decode_step(e);
            if (hipStreamEndCapture(st, &g) == hipSuccess && g) {
                if (hipGraphInstantiate(out, g, nullptr, nullptr, 0) != hipSuccess) *out = nullptr;
                hipGraphDestroy(g);
            }
        }
        (void)hipGetLastError();
    };
    if (e->use_graph && can_split && wants_split(n_steps - 1) && !e->graph_split_tried) {
        e->graph_split_tried = true;
        e->graph_has_logits = e->n_sampling > 0;
        e->split_active = true;
        capture(1, &e->graph_split);
        e->split_active = false;
    }
    if (e->use_graph && !e->graph_tried) {
        e->graph_tried = true;
        e->graph_has_logits = e->n_sampling > 0;
        capture(1, &e->graph);   // several steps per graph were measured: -0.3 % per step (profiles/r02a_sweep_nt_graphsteps.jsonl), not kept
    }
    HIPCHK(e, hipEventRecord(e->ev[2], st));
    for (int s = 0; s < n_steps;) {
        const bool sp = wants_split(s);
        hipGraphExec_t gx = sp ? e->graph_split : e->graph;
        if (gx) HIPCHK(e, hipGraphLaunch(gx, st));
        else { e->split_active = sp; decode_step(e); e->split_active = false; }
        ++s;
    }
    HIPCHK(e, hipEventRecord(e->ev[3], st));
    e->have_dec_time = true;
    HIPCHK(e, hipGetLastError());
    return NTTS_OK;
}

extern "C" int ntts_backbone_sync(ntts_backbone* e) {
    if (!e) return NTTS_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return NTTS_OK;
}

extern "C" int ntts_backbone_poll(ntts_backbone* e, int32_t* state, int32_t* n_new) {
    if (!e || !state) return NTTS_EINVAL;
    const int B = e->cfg.max_batch;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(state, e->sl.state, B * sizeof(int), hipMemcpyDeviceToHost));
    if (n_new) HIPCHK(e, hipMemcpy(n_new, e->sl.n_new, B * sizeof(int), hipMemcpyDeviceToHost));
    return NTTS_OK;
}

// ---- the same poll without stopping the stream: the scheduler enqueues the NEXT burst of decode steps, then looks at the slot
//      states as they were BEFORE that burst -- the GPU never waits for the host to read, release and refill
extern "C" int ntts_backbone_poll_begin(ntts_backbone* e) {
    if (!e) return NTTS_EINVAL;
    if (e->snap_open) return fail(e, NTTS_ESTATE, "a snapshot is already open (ntts_backbone_poll_end first)");
    const int B = e->cfg.max_batch;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemcpyAsync(e->snap_host, e->sl.state, B * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(e->snap_host + B, e->sl.n_new, B * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipEventRecord(e->snap_ev, e->stream));
    e->snap_gen.resize(B);
    for (int b = 0; b < B; ++b) e->snap_gen[b] = e->slots[b].gen;
    e->snap_open = true;
    e->snap_valid = false;
    return NTTS_OK;
}

extern "C" int ntts_backbone_poll_end(ntts_backbone* e, int32_t* state, int32_t* n_new) {
    if (!e || !state) return NTTS_EINVAL;
    if (!e->snap_open) return fail(e, NTTS_ESTATE, "no snapshot is open (ntts_backbone_poll_begin first)");
    const int B = e->cfg.max_batch;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipEventSynchronize(e->snap_ev));     // waits for the work enqueued BEFORE poll_begin only
    memcpy(state, e->snap_host, B * sizeof(int));
    if (n_new) memcpy(n_new, e->snap_host + B, B * sizeof(int));
    e->snap_open = false;
    e->snap_valid = true;
    return NTTS_OK;
}

extern "C" int ntts_backbone_read_finished(ntts_backbone* e, int32_t slot, int32_t* out_ids, int32_t cap, int32_t* n_out) {
    if (!e || slot < 0 || slot >= e->cfg.max_batch || !n_out) return fail(e, NTTS_EINVAL, "bad argument");
    const int B = e->cfg.max_batch;
    // The snapshot entry speaks for the request that occupied the slot when poll_begin was enqueued: a release (+ a new prefill) since
    // then -- before or after poll_end -- makes it stale, and the rows it points at are being overwritten by the new occupant
    if (!e->snap_valid || e->snap_host[slot] != SLOT_FINISHED || e->slots[slot].state == SLOT_FREE || e->snap_gen[slot] != e->slots[slot].gen)
        return fail(e, NTTS_ESTATE, "slot %d was not finished in the last completed snapshot (or was released / refilled since it was taken)", slot);
    HIPCHK(e, hipSetDevice(e->device));
    // A finished slot's ids do not change until it is released, and the snapshot that showed it finished has completed: the
    // copy needs no ordering against the decode steps still queued on the engine's stream, so it goes around them
    const int nn = e->snap_host[B + slot];
    const int k = nn < cap ? nn : cap;
    if (out_ids && k > 0) {
        HIPCHK(e, hipMemcpyAsync(out_ids, e->sl.out_tokens + (size_t)slot * e->sl.out_stride, k * sizeof(int), hipMemcpyDeviceToHost, e->copy_stream));
        HIPCHK(e, hipStreamSynchronize(e->copy_stream));
    }
    *n_out = nn;
    return NTTS_OK;
}

extern "C" int ntts_backbone_read(ntts_backbone* e, int32_t slot, int32_t* out_ids, int32_t cap, int32_t* n_out,
                                  int32_t* finished) {
    if (!e || slot < 0 || slot >= e->cfg.max_batch || !n_out) return fail(e, NTTS_EINVAL, "bad argument");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    int st = 0, nn = 0;
    HIPCHK(e, hipMemcpy(&st, e->sl.state + slot, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(&nn, e->sl.n_new + slot, sizeof(int), hipMemcpyDeviceToHost));
    if (e->slots[slot].state == SLOT_FREE) { st = SLOT_FREE; nn = 0; }
    const int k = nn < cap ? nn : cap;
    if (out_ids && k > 0)
        HIPCHK(e, hipMemcpy(out_ids, e->sl.out_tokens + (size_t)slot * e->sl.out_stride, k * sizeof(int), hipMemcpyDeviceToHost));
    *n_out = nn;
    if (finished) *finished = (st == SLOT_FINISHED) ? 1 : 0;
    return NTTS_OK;
}

extern "C" int ntts_backbone_read_all(ntts_backbone* e, int32_t* out_ids, int32_t cap, int32_t* n_out, int32_t* finished) {
    if (!e || !out_ids || !n_out || cap < 1) return fail(e, NTTS_EINVAL, "bad argument");
    const int B = e->cfg.max_batch;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    std::vector<int> st(B);
    HIPCHK(e, hipMemcpy(st.data(), e->sl.state, B * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(n_out, e->sl.n_new, B * sizeof(int), hipMemcpyDeviceToHost));
    const int w = cap < e->sl.out_stride ? cap : e->sl.out_stride;
    HIPCHK(e, hipMemcpy2D(out_ids, (size_t)cap * sizeof(int), e->sl.out_tokens, (size_t)e->sl.out_stride * sizeof(int),
                          (size_t)w * sizeof(int), B, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) {
        if (e->slots[b].state == SLOT_FREE) { st[b] = SLOT_FREE; n_out[b] = 0; }
        if (finished) finished[b] = st[b] == SLOT_FINISHED ? 1 : 0;
    }
    return NTTS_OK;
}

extern "C" int ntts_backbone_stream(ntts_backbone* e, void** stream) {
    if (!e || !stream) return NTTS_EINVAL;
    *stream = (void*)e->stream;
    return NTTS_OK;
}

// "all work enqueued on a caller-provided hipStream_t".  The engine creates a stream of its own; a caller that wants the
// engine's work ordered inside ITS stream (e.g. torch's current stream) lends it here: every later launch, copy and graph replay goes
// there (a captured decode step replays on any stream).  nullptr returns to the engine's own stream.  Blocking: drains the stream in use.
extern "C" int ntts_backbone_set_stream(ntts_backbone* e, void* stream) {
    if (!e) return NTTS_EINVAL;
    if (e->snap_open) return fail(e, NTTS_ESTATE, "a snapshot is open on the current stream (ntts_backbone_poll_end first)");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->stream = stream ? (hipStream_t)stream : e->own_stream;
    return NTTS_OK;
}

// Drop the side stream of the prompt passes (drained first; a lent one goes back to its owner).
static int drop_prefill_stream(ntts_backbone* e) {
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (e->pf_stream) {
        HIPCHK(e, hipStreamSynchronize(e->pf_stream));
        if (!e->pf_lent) hipStreamDestroy(e->pf_stream);
        hipEventDestroy(e->pf_ev[0]); hipEventDestroy(e->pf_ev[1]);
        e->pf_stream = nullptr;
        e->pf_lent = false;
    }
    return NTTS_OK;
}
static int make_prefill_events(ntts_backbone* e) {
    e->pf_ev[0] = e->pf_ev[1] = nullptr;
    if (hipEventCreateWithFlags(&e->pf_ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->pf_ev[1], hipEventDisableTiming) != hipSuccess) {   // no ordering events, no side stream
        if (e->pf_ev[0]) hipEventDestroy(e->pf_ev[0]);
        if (e->pf_ev[1]) hipEventDestroy(e->pf_ev[1]);
        e->pf_ev[0] = e->pf_ev[1] = nullptr;
        if (!e->pf_lent) hipStreamDestroy(e->pf_stream);
        e->pf_stream = nullptr;
        e->pf_lent = false;
        return fail(e, NTTS_EHIP, "could not create the side stream's ordering events");
    }
    return NTTS_OK;
}

extern "C" int ntts_backbone_set_prefill_cu_mask(ntts_backbone* e, const uint32_t* mask, int32_t n_words) {
    if (!e || n_words < 0 || (n_words > 0 && !mask)) return fail(e, NTTS_EINVAL, "bad CU mask");
    const int rc = drop_prefill_stream(e);
    if (rc) return rc;
    if (n_words == 0) return NTTS_OK;
    long bits = 0;
    for (int i = 0; i < n_words; ++i) bits += __builtin_popcount(mask[i]);
    if (bits < 1) return fail(e, NTTS_EINVAL, "empty CU mask");
    HIPCHK(e, hipExtStreamCreateWithCUMask(&e->pf_stream, (uint32_t)n_words, mask));
    return make_prefill_events(e);
}

// The prompt passes on a stream of the CALLER's, ordered behind and before the engine's stream by events (nothing blocks): a server
// that runs several engines side by side keeps every matrix-core-bound pass (prompt passes, codec passes) in ONE hardware queue and
// each engine's decode chain in a queue of its own -- the runtime maps streams onto four hardware queues, and a prompt pass that
// lands in the queue of another engine's decode chain stalls that chain for its whole length.  nullptr: back to the engine's stream.
extern "C" int ntts_backbone_set_prefill_stream(ntts_backbone* e, void* stream) {
    if (!e) return NTTS_EINVAL;
    const int rc = drop_prefill_stream(e);
    if (rc || !stream) return rc;
    e->pf_stream = (hipStream_t)stream;
    e->pf_lent = true;
    return make_prefill_events(e);
}

extern "C" int ntts_backbone_export_codes(ntts_backbone* e, int32_t n, const int32_t* slots, int32_t speech_base, int32_t n_codes,
                                          int32_t modulo, int32_t* codes_dev, int32_t stride, int32_t* lens_dev) {
    if (!e || n < 1 || !slots || !codes_dev || !lens_dev || stride < 1 || n_codes < 1) return fail(e, NTTS_EINVAL, "bad argument");
    if (n > e->cfg.max_batch || (size_t)n > e->meta_cap) return fail(e, NTTS_EINVAL, "%d slots given, the engine has %d", n, e->cfg.max_batch);
    {
        std::vector<char> seen(e->cfg.max_batch, 0);
        for (int i = 0; i < n; ++i) {
            if (slots[i] < 0 || slots[i] >= e->cfg.max_batch) return fail(e, NTTS_EINVAL, "slot %d out of range", slots[i]);
            if (seen[slots[i]]++) return fail(e, NTTS_EINVAL, "slot %d given twice", slots[i]);
        }
    }
    HIPCHK(e, hipSetDevice(e->device));
    // the slot list travels through the engine's meta block: stream-ordered behind whatever still reads it
    HIPCHK(e, upload_meta(e, slots, (size_t)n, e->stream));
    ExportCodesArgs a{};
    a.slots = e->meta_dev; a.sl = e->sl; a.speech_base = speech_base; a.n_codes = n_codes; a.modulo = modulo;
    a.codes = codes_dev; a.stride = stride; a.lens = lens_dev;
    NTTS_LAUNCH((export_codes_kernel), dim3(n), dim3(256), e->stream, a);
    HIPCHK(e, hipGetLastError());
    return NTTS_OK;
}

extern "C" int ntts_backbone_append_codes(ntts_backbone* e, int32_t n, const int32_t* slots, int32_t speech_base, int32_t n_codes, int32_t modulo,
                                          int32_t* cache_dev, int32_t stride, int32_t* clen_dev, int32_t* seen_dev, int32_t* fin_dev) {
    if (!e || n < 1 || !slots || !cache_dev || !clen_dev || !seen_dev || !fin_dev || stride < 1 || n_codes < 1) return fail(e, NTTS_EINVAL, "bad argument");
    if (n > e->cfg.max_batch || (size_t)n > e->meta_cap) return fail(e, NTTS_EINVAL, "%d slots given, the engine has %d", n, e->cfg.max_batch);
    {
        std::vector<char> seen(e->cfg.max_batch, 0);
        for (int i = 0; i < n; ++i) {
            if (slots[i] < 0 || slots[i] >= e->cfg.max_batch) return fail(e, NTTS_EINVAL, "slot %d out of range", slots[i]);
            if (seen[slots[i]]++) return fail(e, NTTS_EINVAL, "slot %d given twice", slots[i]);
        }
    }
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, upload_meta(e, slots, (size_t)n, e->stream));     // (stream-ordered behind whatever still reads the meta block)
    AppendCodesArgs a{};
    a.slots = e->meta_dev; a.sl = e->sl; a.speech_base = speech_base; a.n_codes = n_codes; a.modulo = modulo;
    a.cache = cache_dev; a.stride = stride; a.clen = clen_dev; a.seen = seen_dev; a.fin = fin_dev;
    NTTS_LAUNCH((append_codes_kernel), dim3(n), dim3(256), e->stream, a);
    HIPCHK(e, hipGetLastError());
    return NTTS_OK;
}

static int release_host(ntts_backbone* e, int32_t slot) {   // host half of a release: pages back to the pool, slot FREE
    HostSlot& s = e->slots[slot];
    drop_pages(e, s);
    s.prompt.clear();
    if (s.sampling) { s.sampling = false; e->n_sampling--; }
    s.state = SLOT_FREE;
    s.gen++;
    return NTTS_OK;
}

extern "C" int ntts_backbone_release(ntts_backbone* e, int32_t slot) {
    if (!e || slot < 0 || slot >= e->cfg.max_batch) return fail(e, NTTS_EINVAL, "bad slot");
    HIPCHK(e, hipSetDevice(e->device));
    // Stream-ordered, no host sync: the slot's pages go back to the pool now, but anything that re-uses them is
    // enqueued on the same stream behind the work that still reads them.
    release_host(e, slot);
    static_assert(SLOT_FREE == 0, "release writes the state with a memset");
    HIPCHK(e, hipMemsetAsync(e->sl.state + slot, 0, sizeof(int), e->stream));
    return NTTS_OK;
}

// ntts_backbone_release for `n` slots at once: ONE stream operation instead of n (a batch server frees a whole batch -- 256 one-int
// fills were 256 launches on the engine's stream between two batches).  Slots must be distinct and valid; nothing is released if one is not.
extern "C" int ntts_backbone_release_many(ntts_backbone* e, int32_t n, const int32_t* slots) {
    if (!e || n < 1 || !slots) return fail(e, NTTS_EINVAL, "bad argument");
    if ((size_t)n > e->meta_cap) return fail(e, NTTS_EINVAL, "too many slots");
    std::vector<char> seen(e->cfg.max_batch, 0);
    int lo = e->cfg.max_batch, hi = -1;
    for (int i = 0; i < n; ++i) {
        const int s = slots[i];
        if (s < 0 || s >= e->cfg.max_batch || seen[s]) return fail(e, NTTS_EINVAL, "bad or repeated slot %d", s);
        seen[s] = 1;
        if (s < lo) lo = s;
        if (s > hi) hi = s;
    }
    HIPCHK(e, hipSetDevice(e->device));
    for (int i = 0; i < n; ++i) release_host(e, slots[i]);
    if (hi - lo + 1 == n) {                                   // a contiguous range (the usual case): one fill
        HIPCHK(e, hipMemsetAsync(e->sl.state + lo, 0, (size_t)n * sizeof(int), e->stream));
    } else {
        HIPCHK(e, upload_meta(e, slots, (size_t)n, e->stream));
        NTTS_LAUNCH((zero_slots_kernel), dim3((n + 63) / 64), dim3(64), e->stream, (const int*)e->meta_dev, n, e->sl.state);
    }
    return NTTS_OK;
}

extern "C" int ntts_backbone_activate(ntts_backbone* e, int32_t n, const int32_t* park_slots, const int32_t* slots) {
    if (!e || n < 1 || !park_slots || !slots) return fail(e, NTTS_EINVAL, "null/empty argument");
    const int B = e->cfg.max_batch, D = e->dec_rows;
    if ((size_t)(2 * n) > e->meta_cap) return fail(e, NTTS_EINVAL, "too many slots");
    std::vector<char> seen(B, 0);
    for (int i = 0; i < n; ++i) {
        const int ps = park_slots[i], s = slots[i];
        if (ps < D || ps >= B || s < 0 || s >= D || seen[ps] || seen[s]) return fail(e, NTTS_EINVAL, "activate: parking row %d -> decode slot %d (parking rows are %d..%d)", ps, s, D, B - 1);
        seen[ps] = seen[s] = 1;
        if (e->slots[ps].state != SLOT_RUNNING) return fail(e, NTTS_ESTATE, "activate: parking row %d holds no request", ps);
        if (e->slots[s].state != SLOT_FREE) return fail(e, NTTS_ESTATE, "activate: decode slot %d is in use", s);
    }
    HIPCHK(e, hipSetDevice(e->device));
    std::vector<int> pairs(2 * (size_t)n);
    for (int i = 0; i < n; ++i) { pairs[2 * i] = park_slots[i]; pairs[2 * i + 1] = slots[i]; }
    HIPCHK(e, upload_meta(e, pairs.data(), pairs.size(), e->stream));
    ActivateArgs a{};
    a.pairs = e->meta_dev; a.sl = e->sl; a.block_table = e->block_table; a.max_pages = e->max_pages;
    NTTS_LAUNCH((activate_slots_kernel), dim3(n), dim3(64), e->stream, a);
    for (int i = 0; i < n; ++i) {
        HostSlot& src = e->slots[park_slots[i]];
        HostSlot& dst = e->slots[slots[i]];
        const unsigned gd = dst.gen, gs = src.gen;
        dst = std::move(src);
        dst.gen = gd + 1;             // (both rows changed hands: open snapshots no longer speak for either)
        src = HostSlot{};
        src.gen = gs + 1;
    }
    HIPCHK(e, hipGetLastError());
    return NTTS_OK;
}

extern "C" int ntts_backbone_set_debug(ntts_backbone* e, int32_t keep_logits) {
    if (!e) return NTTS_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (keep_logits && !e->logits) {
        HIPCHK(e, hipMalloc((void**)&e->logits, (size_t)e->cfg.max_batch * e->cfg.vocab_size * sizeof(float)));
        HIPCHK(e, hipMemset(e->logits, 0, (size_t)e->cfg.max_batch * e->cfg.vocab_size * sizeof(float)));
    } else if (!keep_logits && e->logits) {
        HIPCHK(e, hipFree(e->logits));
        e->logits = nullptr;
    }
    drop_graphs(e);  // the logits pointer is baked into the captured step
    return NTTS_OK;
}

extern "C" int ntts_backbone_debug_force(ntts_backbone* e, int32_t slot, int32_t token) {
    if (!e || slot < 0 || slot >= e->cfg.max_batch || token < 0 || token >= e->cfg.vocab_size) return fail(e, NTTS_EINVAL, "bad argument");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    int nn = 0;
    HIPCHK(e, hipMemcpy(&nn, e->sl.n_new + slot, sizeof(int), hipMemcpyDeviceToHost));
    if (nn < 1) return fail(e, NTTS_ESTATE, "slot %d has generated nothing yet", slot);
    HIPCHK(e, hipMemcpy(e->sl.cur_tok + slot, &token, sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->sl.out_tokens + (size_t)slot * e->sl.out_stride + nn - 1, &token, sizeof(int), hipMemcpyHostToDevice));
    return NTTS_OK;
}

extern "C" int ntts_backbone_read_logits(ntts_backbone* e, int32_t slot, float* out, int32_t n) {
    if (!e || !out || slot < 0 || slot >= e->cfg.max_batch) return NTTS_EINVAL;
    if (!e->logits) return fail(e, NTTS_ESTATE, "logits are not kept: call ntts_backbone_set_debug(e, 1) first");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (n > e->cfg.vocab_size) n = e->cfg.vocab_size;
    if (e->lr_rows) {        // restricted head: the kept row is in column order [range | EOS]; hand it out by token id, -inf elsewhere
        std::vector<float> row(e->lr_rows);
        HIPCHK(e, hipMemcpy(row.data(), e->logits + (size_t)slot * e->cfg.vocab_size, (size_t)e->lr_rows * sizeof(float), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) out[i] = -INFINITY;
        for (int c = 0; c < e->lr_rows; ++c) {
            const int id = c < e->lr_rows - 1 ? e->lr_lo + c : e->lr_eos;
            if (id < n) out[id] = row[c];
        }
        return NTTS_OK;
    }
    HIPCHK(e, hipMemcpy(out, e->logits + (size_t)slot * e->cfg.vocab_size, n * sizeof(float), hipMemcpyDeviceToHost));
    return NTTS_OK;
}

extern "C" int ntts_backbone_last_timing(ntts_backbone* e, float* prefill_ms, float* decode_ms) {
    if (!e) return NTTS_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (prefill_ms) { *prefill_ms = 0; if (e->have_pf_time) HIPCHK(e, hipEventElapsedTime(prefill_ms, e->ev[0], e->ev[1])); }
    if (decode_ms) { *decode_ms = 0; if (e->have_dec_time) HIPCHK(e, hipEventElapsedTime(decode_ms, e->ev[2], e->ev[3])); }
    return NTTS_OK;
}

extern "C" int ntts_backbone_step_bytes(ntts_backbone* e, double* bytes) {
    if (!e || !bytes) return NTTS_EINVAL;
    const ntts_backbone_config& c = e->cfg;
    const int B = c.max_batch;
    std::vector<int> st(B), pos(B);
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(st.data(), e->sl.state, B * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(pos.data(), e->sl.pos, B * sizeof(int), hipMemcpyDeviceToHost));
    const double H = e->H, F = e->F, QD = c.num_heads * e->HD, KD = c.num_kv_heads * e->HD;
    const double wb = e->fp8 ? 1.0 : 2.0;      // bytes per matrix weight (+ 4 per output channel for the fp8 scales)
    const double mats = (QD + 2 * KD) * H + H * QD + 3 * F * H, small = (QD + 2 * KD) + 2 * H;
    const double sc = e->fp8 ? ((QD + 2 * KD) + 2 * H + 2 * F) * 4.0 : 0.0;
    const double w_layers = (mats * wb + small * 2.0 + sc) * c.num_layers + H * 2.0;
    const double head_rows = e->lr_rows ? e->lr_rows : c.vocab_size;     // (restricted head: the rows it streams)
    const double w_head = head_rows * H * wb + (e->fp8 ? head_rows * 4.0 : 0.0);
    const double kv_tok = (double)c.num_layers * 2 * KD * 2.0;  // bytes per cached token (K+V, all layers)
    double kv = 0;
    for (int b = 0; b < B; ++b)
        if (st[b] == SLOT_RUNNING) kv += (double)pos[b] * kv_tok + kv_tok;  // read L tokens, write 1
    *bytes = w_layers + w_head + kv;
    return NTTS_OK;
}

// Diagnostics: one launch of the decode attention kernel of `layer` at the current slot state with its phase
// timestamps recorded (attn_decode.h `mark`): out[((b * nkv + kvh) * 4 + wave) * 8 + phase], 100 MHz ticks.
extern "C" int ntts_backbone_attn_timeline(ntts_backbone* e, int32_t layer, uint64_t* out, int64_t cap) {
    if (!e || !out || layer < 0 || layer >= e->cfg.num_layers) return fail(e, NTTS_EINVAL, "bad argument");
    const size_t n = (size_t)e->cfg.max_batch * e->cfg.num_kv_heads * 4 * 8;
    if (cap < (int64_t)n) return fail(e, NTTS_EINVAL, "timeline needs %zu entries", n);
    HIPCHK(e, hipSetDevice(e->device));
    DevScratch buf;
    HIPCHK(e, hipMalloc(&buf.p, n * 8));
    unsigned long long* tl = (unsigned long long*)buf.p;
    HIPCHK(e, hipMemsetAsync(tl, 0, n * 8, e->stream));
    auto attn = [&](int i) { if (e->small) ks_attn(e, i); else k_attn(e, i); };
    auto qkv = [&](int i) { if (e->small) ks_qkv(e, i); else k_qkv(e, i); };
    k_step_meta(e);
    // the fused QKV kernel writes this step's K entry and the q | v row the attention kernel reads (the real step runs it first, too)
    qkv((layer + 1) % e->cfg.num_layers);
    qkv(layer);
    attn((layer + 1) % e->cfg.num_layers);     // another layer first: this launch is neither the first nor cache-warm
    e->attn_tl = tl;
    attn(layer);
    e->attn_tl = nullptr;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(out, tl, n * 8, hipMemcpyDeviceToHost));
    return NTTS_OK;
}

// Phase timestamps of one small-batch GEMV launch at the current slot state (tools/gemv_timeline.py): which = 1 QKV (+ RoPE + K append), 2 o_proj, 3 gate/up,
// 4 down_proj of `layer`; out[workgroup][16] (gemv.h GemvArgs::tl), *n_wg = workgroups of the launch.
extern "C" int ntts_backbone_gemv_timeline(ntts_backbone* e, int32_t which, int32_t layer, uint64_t* out, int64_t cap, int32_t* n_wg) {
    if (!e || !out || !n_wg || layer < 0 || layer >= e->cfg.num_layers || which < 1 || which > 4) return fail(e, NTTS_EINVAL, "bad argument");
    const size_t n = 4096 * 16;
    if (cap < (int64_t)n) return fail(e, NTTS_EINVAL, "timeline needs %zu entries", n);
    HIPCHK(e, hipSetDevice(e->device));
    DevScratch buf;
    HIPCHK(e, hipMalloc(&buf.p, n * 8));
    unsigned long long* tl = (unsigned long long*)buf.p;
    HIPCHK(e, hipMemsetAsync(tl, 0, n * 8, e->stream));
    auto run = [&](int i) {
        if (e->small) { if (which == 1) ks_qkv(e, i); else if (which == 2) ks_o_proj(e, i); else if (which == 3) ks_gate_up(e, i); else ks_down(e, i); }
        else { if (which == 1) k_qkv(e, i); else if (which == 2) k_o_proj(e, i); else if (which == 3) k_gate_up(e, i); else k_down(e, i); }
    };
    if (which == 1) k_step_meta(e);
    run((layer + 1) % e->cfg.num_layers);      // another layer first: this launch is neither the first nor cache-warm
    e->gemv_tl = tl;
    run(layer);
    e->gemv_tl = nullptr;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(out, tl, n * 8, hipMemcpyDeviceToHost));
    int wg = 0;
    for (size_t i = 0; i < 4096; ++i) if (out[i * 16] || out[i * 16 + 8]) wg = (int)i + 1;
    *n_wg = wg;
    return NTTS_OK;
}

// ------------------------------------------------------------------------------------------------
// per-kernel timing at the CURRENT slot state (bench.py roofline leg).  Every kernel of the decode
// step is idempotent with respect to the slot state (the fused QKV kernel re-writes the K entry of the
// current position, the attention kernel its V^T entry), so replaying one in isolation does not disturb
// generation.  The K entry of the current position is written by the QKV kernel, not by attention: the
// attention replays below are preceded by ONE untimed QKV pass over every layer, as in the real step --
// otherwise they would read a page slot no kernel has written yet (stale data of the page's last owner).
// ------------------------------------------------------------------------------------------------
extern "C" int ntts_backbone_time_kernel(ntts_backbone* e, int32_t which, int32_t iters, float* avg_ms, double* alg_bytes,
                                         int32_t* launches_per_step) {
    if (!e || !avg_ms || !alg_bytes || !launches_per_step || iters < 1) return fail(e, NTTS_EINVAL, "bad argument");
    if (!e->finalized) return fail(e, NTTS_ESTATE, "weights not finalised");
    HIPCHK(e, hipSetDevice(e->device));
    const ntts_backbone_config& c = e->cfg;
    const int B = c.max_batch, H = e->H, F = e->F, QD = c.num_heads * e->HD, KD = c.num_kv_heads * e->HD, L = c.num_layers;
    hipStream_t st = e->stream;
    HIPCHK(e, hipStreamSynchronize(st));
    std::vector<int> sst(B), pos(B);
    HIPCHK(e, hipMemcpy(sst.data(), e->sl.state, B * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(pos.data(), e->sl.pos, B * sizeof(int), hipMemcpyDeviceToHost));
    double kv_layer = 0;  // bytes one attention launch must read (+ write): K and V of every cached token
    for (int b = 0; b < B; ++b)
        if (sst[b] == SLOT_RUNNING) kv_layer += ((double)pos[b] + 1) * 2 * KD * 2.0;
    const double wb = e->fp8 ? 1.0 : 2.0;      // bytes per matrix weight / per GEMM-input activation element
    const double act = (double)B * wb;
    const int kt_ = ktile_of(e);
    // Every replay works on the NEXT layer's weights / KV pool, as consecutive launches of this kernel do inside the
    // decode step: one layer's operands (84 MB of KV at batch 256) would sit in the 256 MB Infinity Cache when replayed
    // alone, all layers together (2 GB) do not -- so the timing below is HBM-cold like the in-graph launches.
    auto run = [&](int k, int i) {
        if (e->small) {
            switch (k) {
                case 0: ks_attn(e, i); break;
                case 1: ks_qkv(e, i); break;         // incl. the fused slab-reduce + residual + RMSNorm prologue
                case 2: ks_o_proj(e, i); break;
                case 3: ks_gate_up(e, i); break;     // incl. its fused prologue
                case 4: ks_down(e, i); break;
                case 5: k_lm_head(e, false, e->dec_rows); break;
                case 6: ks_final_norm(e); break;
                default: break;
            }
            return;
        }
        switch (k) {
            case 0: k_attn(e, i); break;                        // paged decode attention (+RoPE, +KV append)
            case 1: k_qkv(e, i); break;
            case 2: k_o_proj(e, i); break;
            case 3: k_gate_up(e, i); break;
            case 4: k_down(e, i); break;
            case 5: k_lm_head(e, false, e->dec_rows); break;
            case 6: k_add_norm(e, QD, e->ks_o, e->layers[i].ln2, e->o_pf, e->xn_pf, e->layers[i].xs[2]); break;   // scratch outputs
            default: break;
        }
    };
    // algorithmic bytes per launch: the weights of the GEMM + its activations in/out; for attention K/V only
    const int kso = e->small ? ntts_backbone::kSksO : e->ks_o, ksd = e->small ? ntts_backbone::kSksD : e->ks_d;
    switch (which) {
        case 0: *alg_bytes = kv_layer;   // SURVEY 8(d), strictly: K and V of every cached token (+ the appended one); the q/k/v
                                         // inputs (bf16 row or the QKV GEMM's fp32 slabs) and the output are the builder's own
                *launches_per_step = L; break;
        case 1: *alg_bytes = (double)e->NQKV * H * wb + e->NQKV * 2.0 + act * H +
                             (double)B * e->NQKV * 2.0;   // bf16 q | v rows + the K entry
                *launches_per_step = L; break;
        case 2: *alg_bytes = (double)H * QD * wb + act * QD + (double)gemm_nsplit(QD, kso, kt_) * B * H * 4.0;
                *launches_per_step = L; break;
        case 3: *alg_bytes = (double)2 * F * H * wb + act * (H + F); *launches_per_step = L; break;
        case 4: *alg_bytes = (double)H * F * wb + act * F + (double)gemm_nsplit(F, ksd, kt_) * B * H * 4.0; *launches_per_step = L; break;
        case 5: *alg_bytes = (double)(e->lr_rows ? e->lr_rows : c.vocab_size) * H * wb + act * H; *launches_per_step = 1; break;
        case 6: *alg_bytes = (double)gemm_nsplit(e->small ? F : QD, e->small ? ksd : kso, kt_) * B * H * 4.0 + (double)B * 2.0 * H * 2 + act * H;
                *launches_per_step = e->small ? 1 : 2 * L; break;
        default: return fail(e, NTTS_EINVAL, "unknown kernel id %d", which);
    }
    if (which == 6 && B > e->Tmax) return fail(e, NTTS_EINVAL, "scratch too small");
    k_step_meta(e);   // the fused QKV kernel appends at the CURRENT position (not yet written), like the step it replays
    if (which == 0)
        for (int i = 0; i < L; ++i) run(1, i);   // every layer's K entry + q | v row for the attention replays (untimed)
    run(which, L - 1);  // warm (code, TLBs); the timed replays start from layer 0
    HIPCHK(e, hipEventRecord(e->ev[2], st));
    for (int i = 0; i < iters; ++i) run(which, i % L);
    HIPCHK(e, hipEventRecord(e->ev[3], st));
    HIPCHK(e, hipStreamSynchronize(st));
    float ms = 0;
    HIPCHK(e, hipEventElapsedTime(&ms, e->ev[2], e->ev[3]));
    *avg_ms = ms / iters;
    return NTTS_OK;
}
