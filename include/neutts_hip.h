/* neutts_hip.h -- C ABI of libneutts_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for the NeuTTS synthesis hot path.  The reference has no FFI of its own:
 * its seam is two third-party Python calls, and each entry point below names the one it replaces.
 *
 *   backbone:  self.backbone.generate(prompt[1,S], max_length, eos_token_id, do_sample, temperature,
 *              top_k, use_cache, min_new_tokens) -> ids            ref:neutts/neutts.py:338-347
 *              (transformers Qwen2ForCausalLM + GenerationMixin._sample, un-vendored dependency)
 *   codec:     self.codec.decode_code(codes[1,1,T]) -> wav[1,1,480*T]   ref:neutts/neutts.py:288-291
 *              (neucodec NeuCodec.decode_code, un-vendored dependency)
 *
 * Conventions
 *   - plain C types, raw pointers and sizes; no torch / C++ types cross the boundary.
 *   - every function returns 0 on success, a negative NTTS_E* code on failure, and never throws;
 *     ntts_last_error() returns the message of the most recent failure on that engine
 *     (pass NULL for failures of ntts_*_create itself).
 *   - the CALLER owns every buffer it passes in; the library owns only what it allocates itself
 *     (weight arena, KV page pool, workspaces), all in the HBM of the device given at create time.
 *   - an engine is bound to ONE device, is not re-entrant and not thread-safe: serialise calls per
 *     engine (the Python host holds the GIL across each call).  One process per GPU; there is no
 *     cross-GPU collective inside any entry point (utterances are independent).
 *   - work is enqueued on the engine's HIP stream -- its own, or one the caller lends with ntts_backbone_set_stream /
 *     ntts_codec_set_stream (ABI 6); functions named *_sync or documented as "blocking" wait for it, the others return
 *     once the work is enqueued.
 *   - there is NO CPU fallback: without a gfx950 device ntts_*_create fails with NTTS_ENODEV.
 */
#ifndef NEUTTS_HIP_H
#define NEUTTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTTS_ABI_VERSION 9

enum {
    NTTS_OK = 0,
    NTTS_EINVAL = -1,   /* bad argument / unknown tensor name / shape mismatch */
    NTTS_ENODEV = -2,   /* no usable gfx950 device */
    NTTS_ENOMEM = -3,   /* HBM allocation or KV page pool exhausted */
    NTTS_ESTATE = -4,   /* call out of order (weights not finalised, slot busy, ...) */
    NTTS_EHIP = -5      /* a HIP runtime call failed; see ntts_last_error */
};

enum { NTTS_DT_F32 = 0, NTTS_DT_BF16 = 1, NTTS_DT_I32 = 2, NTTS_DT_FP8_E4M3 = 3 /* ABI 6: pre-quantised matrices of an fp8 checkpoint (bytes) */ };

/* ------------------------------------------------------------------------------------------ */
/* Backbone engine: Qwen2-style decoder, paged KV cache, continuous batching over `max_batch`   */
/* decode slots.  Replaces transformers' Qwen2ForCausalLM.forward + GenerationMixin._sample     */
/* (hf:models/qwen2/modeling_qwen2.py:342-477, hf:generation/utils.py:2783-2973) behind         */
/* ref:neutts/neutts.py:338-347.                                                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct ntts_backbone ntts_backbone;

typedef struct ntts_backbone_config {
    int32_t vocab_size;         /* V (lm_head tied to the embedding) */
    int32_t hidden_size;        /* 896 */
    int32_t intermediate_size;  /* 4864 */
    int32_t num_layers;         /* 24 */
    int32_t num_heads;          /* 14 */
    int32_t num_kv_heads;       /* 2  (num_heads / num_kv_heads must be <= 16) */
    int32_t head_dim;           /* 64 (NeuTTS-Air: the fused / tiered attention kernels) or 128 (ABI 9: the general attention path -- plain QKV GEMM +
                                 * a norm / RoPE / KV-append pass, head_dim-templated two-sweep prompt attention and decode attention; bf16 only;
                                 * the small-batch GEMV step, context-split attention and the resident / deep prompt tiers are off) */
    float rms_eps;              /* 1e-6 */
    int32_t max_context;        /* ref:neutts/neutts.py:85  (2048) */
    int32_t max_batch;          /* number of slots; the decode step runs over the first max_batch - park_slots of them */
    int32_t num_pages;          /* KV pool size in pages of NTTS_PAGE_TOKENS tokens; 0 = max_batch * max_context / page */
    int32_t max_prefill_tokens; /* workspace rows for one packed prefill call; 0 = 16384 */
    /* ABI 2: architecture / precision switches of the decoder family the reference's AutoModelForCausalLM dispatch covers
     * (ref:neutts/neutts.py:164: NeuTTS-Air = Qwen2; NeuTTS-Nano ships under the same class surface). */
    int32_t tie_word_embeddings; /* 1: lm_head = embedding (Qwen2.5-0.5B / NeuTTS-Air); 0: a separate "lm_head.weight" must be loaded */
    int32_t attention_bias;      /* 1: q/k/v_proj carry a bias (Qwen2); 0: none (Llama-style) */
    int32_t qk_norm;             /* ABI 9: 1 = Qwen3-style per-head RMSNorm of q and k (over head_dim, weights "self_attn.q_norm.weight" / "k_norm.weight",
                                  * before RoPE: hf:models/qwen3/modeling_qwen3.py Qwen3Attention.forward); takes the general attention path like head_dim 128 */
    int32_t weight_dtype;        /* NTTS_W_BF16 | NTTS_W_FP8_E4M3: fp8 weights with per-output-channel scales, fp8 GEMM inputs with
                                    static per-tensor scales ("*.input_scale" tensors), bf16 residual stream / KV / attention */
    int32_t park_slots;          /* ABI 9.  The LAST park_slots slots are PARKING rows: a prompt pass fills them like any slot (KV pages, first token) but no
                                  * decode step touches them; ntts_backbone_activate moves a parked request into a free decode slot.  A continuous-batching
                                  * scheduler admits prompts in efficient waves into the parking rows while every decode row stays busy (the reference has no
                                  * scheduler: ref:neutts/neutts.py:335 runs one utterance at a time).  0 = none (a zeroed field: ABI 8 behaviour) */
} ntts_backbone_config;
enum { NTTS_W_BF16 = 0, NTTS_W_FP8_E4M3 = 1 };

#define NTTS_PAGE_TOKENS 32

int ntts_abi_version(void);
const char* ntts_last_error(const ntts_backbone* e);

int ntts_backbone_create(const ntts_backbone_config* cfg, int device, ntts_backbone** out);
void ntts_backbone_destroy(ntts_backbone* e);

/* Weight upload.  `name` is the HF state-dict key of Qwen2ForCausalLM ("model.embed_tokens.weight",
 * "model.layers.{i}.self_attn.q_proj.weight|bias", ..., "model.norm.weight"; "lm_head.weight": with tie_word_embeddings
 * it must hold the embedding's values -- anything else is NTTS_EINVAL, never silently dropped -- without, it is the
 * separate head matrix) plus "rope.inv_freq" (fp32 [head_dim/2], the buffer hf:models/qwen2/modeling_qwen2.py:86 computes).
 * NTTS_W_FP8_E4M3: the matrices are quantised on upload (per output channel: scale = max|w| / 448, w_q = e4m3(w / scale)) and
 * the static activation scales of the four GEMM inputs of a layer and of the head are loaded as fp32 scalars named
 * "model.layers.{i}.self_attn.q_proj.input_scale" (= k_proj / v_proj), "...self_attn.o_proj.input_scale",
 * "...mlp.gate_proj.input_scale" (= up_proj), "...mlp.down_proj.input_scale", "lm_head.input_scale" (the naming of
 * static-fp8 checkpoints).  ABI 6, PRE-QUANTISED fp8 checkpoints: a matrix may arrive as NTTS_DT_FP8_E4M3 bytes ([N][K] row-major e4m3fn) together with
 * its "<module>.weight_scale" tensor (fp32: one value per output channel, [N] or [N, 1], or ONE value for the whole matrix) -- in either order; it is
 * then stored as it is, not re-quantised (finalize insists on the scale; a scale for a matrix that was quantised on upload is NTTS_EINVAL).  The
 * embedding stays bf16 / fp32 (it is gathered, not multiplied); with tie_word_embeddings the head's fp8 copy is derived from it.
 * `data` may be a host or a device pointer
 * (is_device); dtype NTTS_DT_F32 or NTTS_DT_BF16 (fp32 is rounded to bf16 RNE, like `.to(bfloat16)`).
 * The tensor is copied into the engine's packed arena (fused QKV rows, gate/up interleaved in 16-row
 * groups) -- the caller's buffer can be freed on return.  Blocking. */
int ntts_backbone_load_tensor(ntts_backbone* e, const char* name, const void* data, int dtype,
                              const int64_t* shape, int ndim, int is_device);
/* All tensors present?  Builds the RoPE table and freezes the arena. */
int ntts_backbone_finalize(ntts_backbone* e);
/* The packed arena (device pointer + bytes): what rank 0 broadcasts over RCCL/xGMI to the other
 * ranks at start-up instead of every rank reading the checkpoint.  Valid after
 * finalize on the sender; a receiver calls ntts_backbone_adopt_arena() after the broadcast landed. */
int ntts_backbone_arena(ntts_backbone* e, void** dev_ptr, size_t* bytes);
int ntts_backbone_adopt_arena(ntts_backbone* e);
/* The byte range of the arena that is derived from another part of it (the tied lm_head's own tile-major / fp8 copy of the
 * embedding; *bytes = 0 if there is none): a broadcast may skip it, ntts_backbone_adopt_arena rebuilds it on the receiver. */
int ntts_backbone_arena_derived(ntts_backbone* e, size_t* off, size_t* bytes);
/* Device-to-device copy between the arena and a caller buffer of exactly the arena size (the staging
 * tensor torch.distributed broadcasts): to_arena = 0 arena -> buf, 1 buf -> arena.  Blocking. */
int ntts_backbone_arena_copy(ntts_backbone* e, void* buf, size_t bytes, int to_arena);
/* (ABI 7) A second engine on the SAME weights: `e` (created with the donor's configuration on the donor's device, nothing loaded
 * yet) drops its own arena and reads the finalised donor's; KV pool, slots, workspaces, stream and step graph stay its own.  For
 * several decode chains side by side on one GPU (ref:neutts/neutts.py:338-347 run for more than one batch at a time: each engine
 * replays its own step graph on its own stream, the launching thread alternates between them).  Weight loads into either
 * engine are refused from then on (NTTS_ESTATE); at most 16 engines per arena. */
int ntts_backbone_share_arena(ntts_backbone* e, ntts_backbone* donor);

/* How many decode chains run side by side on this GPU, this engine's included (an engine gang: several engines on one arena, each
 * on a lane stream, their step graphs replayed alternately).  The decode step's shape follows it: alone, an engine tiles its
 * GEMMs to fill the chip by itself (64-row m-blocks, split-K aimed at 224 workgroups, row-block XCD placement); in a gang the
 * other chains fill the chip and what counts is how many bytes each CU pulls, so o_proj / down_proj take the 256-row tile (a weight
 * tile passes through a CU's load path once per chain) and the XCD placement is off.  Same arithmetic, same
 * summation order per output element: ids and logits do not depend on it, bit for bit.
 * ABI 9: the engine COUNTS the chains itself -- at every ntts_backbone_decode call, the engines of its arena (ntts_backbone_share_arena)
 * whose latest decode call lies within the last 50 ms, itself included -- and keeps one captured step graph per shape, so a caller
 * that never heard of this function gets the right shape, and an engine that leaves its gang for a while (one utterance on engine 0)
 * runs the single-chain shape meanwhile.  This call PINS the count instead (chains > 0: tests, sweeps, profiling one engine in the
 * gang's shape); chains = 0 returns to counting.  ABI 8 had only the pinned form (and dropped the captured graph on every call).
 * Replaces nothing in the reference (ref:neutts/neutts.py runs one utterance at a time); the arena is reference-counted (atomically since
 * ABI 9): donor and readers may be destroyed in any order, from any thread. */
int ntts_backbone_set_gang(ntts_backbone* e, int32_t chains);

/* ABI 8, OPT-IN.  Restrict the lm_head to the token ids [lo, hi) plus eos_id: the decode step streams a compacted copy of those
 * rows (NeuTTS-Air: 65 537 x 896 bf16 = 118 MB instead of 390 MB) and every other id is treated as logit -inf.  The reference
 * only ever CONSUMES ids of that shape -- ref:neutts/neutts.py:276 keeps `<|speech_N|>` tokens, ref:neutts/neutts.py:336-341 stops
 * at `<|SPEECH_GENERATION_END|>` -- but hf:generation/utils.py:2894-2925 takes the argmax / top-k over the WHOLE vocabulary: the
 * GREEDY ids are the reference's whenever its argmax lies in the range, and differ otherwise; a SAMPLED draw (do_sample, top_k)
 * matches the reference's only when the WHOLE full-vocabulary top-k set lies in range + EOS (one candidate outside changes the
 * candidate set and the softmax normaliser) and eos_id > hi (the compacted columns are in column order = token-id order only
 * then; true for NeuTTS-Air).  Never the parity configuration.
 * lo < 0 restores the full head.  While a range is set, requests must use eos_token_id == eos_id; no slot may be in use when this
 * is called.  Per engine (twins of a gang each build their own 118 MB copy). */
int ntts_backbone_set_logits_range(ntts_backbone* e, int32_t lo, int32_t hi, int32_t eos_id);

/* ABI 8.  fp8 activation-scale calibration.  The reference's quantised builds are ready-made files (ref:README.md:59-64: Q4 / Q8 GGUF);
 * the fp8 model here takes static per-tensor `input_scale`s as a static-fp8 checkpoint ships them.  For a user who holds the bf16
 * checkpoint: calibration mode on a BF16 engine (enable = 1 resets the record, 0 ends the mode) makes every prompt pass record
 * max |x| of every GEMM's input rows; ntts_backbone_read_amax hands the n = 4 * num_layers + 1 values out, [layer][q_proj (QKV), o_proj,
 * gate_proj (gate/up), down_proj] and the lm_head's last.  input_scale = amax / 448 (tools/calibrate_fp8.py writes them). */
int ntts_backbone_calibrate(ntts_backbone* e, int32_t enable);
int ntts_backbone_read_amax(ntts_backbone* e, float* out, int32_t n);

/* ABI 9.  Move `n` parked requests into free decode slots: park_slots[i] (a parking row holding a prefilled request: its KV pages, position, first token and
 * sampling state) -> slots[i] (a FREE decode slot, < max_batch - park_slots).  Stream-ordered behind the prompt pass that filled the parking row and ahead of
 * the next decode step, which then treats the request like any other; the parking row is free again.  Ids cannot depend on it: a request's arithmetic does
 * not depend on the slot it runs in (tests/test_gpu_backbone.py slot-invariance tests). */
int ntts_backbone_activate(ntts_backbone* e, int32_t n, const int32_t* park_slots, const int32_t* slots);

/* Sampling contract of one request = the keyword arguments of the reference's generate() call
 * (ref:neutts/neutts.py:338-347). */
typedef struct ntts_sampling {
    int32_t max_length;      /* total length cap (prompt + new), <= max_context        [2048] */
    int32_t min_new_tokens;  /* EOS logit = -inf while fewer new tokens than this      [50]   */
    int32_t eos_token_id;    /* <|SPEECH_GENERATION_END|>                                      */
    int32_t do_sample;       /* 0 = greedy argmax (first max wins); 1 = temperature, top-k, multinomial [1] */
    int32_t top_k;           /* [50]  >= 1; logits below the k-th largest are dropped, ties at the k-th kept (<= 512 kept) */
    float temperature;       /* [1.0] > 0 */
    uint64_t seed;           /* Philox4x32-10 key for do_sample=1; the draw of step t depends on (seed, t) only: give each request its own seed */
} ntts_sampling;

/* Prefill `n` prompts (packed back to back in `ids`, prompt i has lens[i] tokens) into the decode
 * slots slots[i] (each must be free), run the model over them (causal attention, KV written to
 * freshly allocated pages) and choose each sequence's first new token.  `ids`, `lens`, `slots`,
 * `samp` are HOST pointers (samp: one entry per prompt).  sum(lens) <= max_prefill_tokens.
 * Non-blocking on the engine stream except for the small H2D copies. */
int ntts_backbone_prefill(ntts_backbone* e, int32_t n, const int32_t* ids, const int32_t* lens,
                          const int32_t* slots, const ntts_sampling* samp);

/* Prefix sharing.  The reference rebuilds, for every utterance of a speaker, a prompt that begins
 * with the same tokens -- chat header + phonemised reference text (ref:neutts/neutts.py:307,315-325) -- and runs the
 * whole prompt through the model again (ref:neutts/neutts.py:338-347 via generate()).  Here prompt i may name a donor:
 * a RUNNING slot, or a prompt given earlier in this same call (by its slot), whose prompt starts with the same
 * shared_len[i] tokens (checked; NTTS_EINVAL on a mismatch).  The KV pages holding the first
 * floor(min(shared_len, donor length, lens[i]-1) / NTTS_PAGE_TOKENS) * NTTS_PAGE_TOKENS tokens are then shared
 * (reference-counted, never written again) and only the remaining tokens of prompt i are computed.  `ids` / `lens`
 * still describe the FULL prompts; only the computed tokens count against max_prefill_tokens.  donor_slot[i] < 0 = no
 * sharing for prompt i.  Results are identical to ntts_backbone_prefill: every K/V row and every query sees the same
 * operands in the same order (tests/test_emu_prefix.py, tests/test_gpu_backbone.py). */
int ntts_backbone_prefill_shared(ntts_backbone* e, int32_t n, const int32_t* ids, const int32_t* lens,
                                 const int32_t* slots, const ntts_sampling* samp, const int32_t* donor_slot,
                                 const int32_t* shared_len);
/* KV pool occupancy and prompt-token accounting since create: free / total pages, prompt tokens pushed through the
 * layers and prompt tokens served from shared pages.  Any out pointer may be NULL.  Host state only (no sync). */
int ntts_backbone_kv_stats(ntts_backbone* e, int32_t* free_pages, int32_t* total_pages, int64_t* prompt_tokens_computed,
                           int64_t* prompt_tokens_shared);

/* Run `n_steps` decode steps over all `max_batch` rows (one hipGraph replay per step).  Rows whose
 * slot is free or finished are carried along masked: they attend to nothing and emit nothing.
 * KV pages for up to n_steps more tokens per running slot are reserved first (NTTS_ENOMEM if the
 * pool is exhausted; nothing is enqueued then). */
int ntts_backbone_decode(ntts_backbone* e, int32_t n_steps);

/* Blocking.  Copy out what slot `slot` has generated so far (prompt stripped, like
 * ref:neutts/neutts.py:348-351): up to `cap` ids into out_ids, the count into *n_out, and
 * *finished = 1 once EOS was emitted or max_length reached.  The EOS id itself is included, as in
 * generate()'s return value. */
int ntts_backbone_read(ntts_backbone* e, int32_t slot, int32_t* out_ids, int32_t cap, int32_t* n_out,
                       int32_t* finished);
/* Blocking.  ntts_backbone_read for every slot at once (three device-to-host copies in total): row s of
 * out_ids (row stride `cap` ints) gets slot s's new ids, n_out[s] / finished[s] as above; free slots report 0. */
int ntts_backbone_read_all(ntts_backbone* e, int32_t* out_ids, int32_t cap, int32_t* n_out, int32_t* finished);
/* Blocking.  One int32 per slot: 0 free, 1 running, 2 finished; new-token counts in n_new (may be NULL). */
int ntts_backbone_poll(ntts_backbone* e, int32_t* state, int32_t* n_new);
/* ABI 5.  The same poll WITHOUT stopping the stream, for a scheduler that keeps one burst of decode steps queued ahead of its own
 * bookkeeping (the stopping criteria of hf:generation/utils.py:2783-2973 are evaluated on the device every step; the host only
 * has to notice finished rows some steps later).  _begin enqueues a copy of every slot's state / new-token count into page-locked
 * memory behind the work issued so far and returns; _end waits for THAT copy only -- not for anything enqueued after _begin -- and
 * hands the values out (n_new may be NULL).  One snapshot may be open at a time (NTTS_ESTATE otherwise). */
int ntts_backbone_poll_begin(ntts_backbone* e);
int ntts_backbone_poll_end(ntts_backbone* e, int32_t* state, int32_t* n_new);
/* ABI 5.  ntts_backbone_read for a slot that the last completed snapshot (poll_begin / poll_end) showed FINISHED and that has not
 * been released since: its ids are final, so the copy runs on a side stream past the decode steps still queued on the engine's
 * stream (ntts_backbone_read would wait for all of them).  NTTS_ESTATE if the slot was not finished in that snapshot. */
int ntts_backbone_read_finished(ntts_backbone* e, int32_t slot, int32_t* out_ids, int32_t cap, int32_t* n_out);
/* The id -> code hand-off of ref:neutts/neutts.py:349 (tokenizer.decode) + :276 (regex over "<|speech_N|>") without leaving the
 * device: for each of the `n` decode slots `slots[i]` (HOST array), the new ids that are speech tokens -- speech_base <= id <
 * speech_base + n_codes -- are written in order as id - speech_base to codes_dev[i * stride ...] (DEVICE int32, at most `stride`
 * per slot) and their count to lens_dev[i] (DEVICE).  modulo != 0 (synthetic benchmark only): every id becomes
 * id mod n_codes.  Enqueued on the engine's stream (ntts_backbone_stream) behind the decode steps issued so far. */
int ntts_backbone_export_codes(ntts_backbone* e, int32_t n, const int32_t* slots, int32_t speech_base, int32_t n_codes,
                               int32_t modulo, int32_t* codes_dev, int32_t stride, int32_t* lens_dev);
/* ABI 6.  The same hand-off for STREAMS (ref:neutts/neutts.py:390-404 appends every generated "<|speech_N|>" token to the stream's token
 * cache as it arrives): for stream i (decode slot slots[i], HOST array) the ids generated since the previous call -- counted by seen_dev[i] --
 * are filtered / mapped like ntts_backbone_export_codes does and APPENDED to cache_dev[i * stride + clen_dev[i] ...]; clen_dev[i] and
 * seen_dev[i] advance, fin_dev[i] = 1 once the slot has finished (EOS / max_length), else 0.  All four are DEVICE int32 arrays owned by the
 * caller (ntts_streams_* below is the in-tree user).  Enqueued on the engine's stream behind the decode steps issued so far. */
int ntts_backbone_append_codes(ntts_backbone* e, int32_t n, const int32_t* slots, int32_t speech_base, int32_t n_codes, int32_t modulo,
                               int32_t* cache_dev, int32_t stride, int32_t* clen_dev, int32_t* seen_dev, int32_t* fin_dev);
/* The engine's HIP stream (a hipStream_t), for a consumer that must order its own work behind the engine's. */
int ntts_backbone_stream(ntts_backbone* e, void** stream);
/* ABI 6.  SURVEY.md 8b ("all work enqueued on a caller-provided hipStream_t"): the engine creates a stream of its own; a caller that wants
 * the engine's launches, copies and graph replays ordered inside ITS stream (e.g. torch's current stream) lends it here; NULL returns to the
 * engine's own stream.  Blocking (drains the stream in use); NTTS_ESTATE while a snapshot (poll_begin) is open. */
int ntts_backbone_set_stream(ntts_backbone* e, void* stream);
/* Serving-side scheduling knob (no reference counterpart: ref:neutts/neutts.py runs one utterance at a time): run this engine's
 * prompt passes (ntts_backbone_prefill*) on a side stream restricted to the compute units whose bits are set in `mask` (n_words
 * 32-bit words, bit i of word w = CU 32 w + i; hipExtStreamCreateWithCUMask), ordered behind and before the engine's own stream.
 * A prefill is compute-bound and its 1024-thread workgroups occupy whole CUs; confined to a subset it leaves the rest of the GPU
 * to another engine's decode steps, which are latency-bound (bench.py pipelines consecutive batches this way).  n_words = 0
 * restores the default (prompt pass on the engine's stream, all CUs).  Blocking (drains the engine's streams). */
int ntts_backbone_set_prefill_cu_mask(ntts_backbone* e, const uint32_t* mask, int32_t n_words);
/* (ABI 7) The same side-stream arrangement on a stream the CALLER owns (no CU mask): every later prompt pass is enqueued there,
 * ordered behind the work already on the engine's stream and before what follows on it by events -- nothing blocks per pass.
 * For several engines side by side: all matrix-core-bound passes in one hardware queue, each engine's decode chain in its own
 * (the HIP runtime multiplexes streams onto four hardware queues; a prompt pass in the queue of another engine's decode chain
 * stalls that chain).  NULL restores the default.  Replaces a CU-masked side stream and vice versa.  Blocking (drains the engine's
 * streams) at the switch itself. */
int ntts_backbone_set_prefill_stream(ntts_backbone* e, void* stream);
/* Return the slot's KV pages to the pool and mark it free. */
int ntts_backbone_release(ntts_backbone* e, int32_t slot);
/* ntts_backbone_release for `n` distinct slots with ONE stream operation (a server frees a whole batch at once); nothing is released if a
 * slot is invalid or repeated. */
int ntts_backbone_release_many(ntts_backbone* e, int32_t n, const int32_t* slots);
int ntts_backbone_sync(ntts_backbone* e);

/* Test / profiling taps (blocking).  Last-step fp32 logits are only materialised when
 * `keep_logits` was enabled; row = slot. */
int ntts_backbone_set_debug(ntts_backbone* e, int32_t keep_logits);
int ntts_backbone_read_logits(ntts_backbone* e, int32_t slot, float* out, int32_t n);
/* Teacher forcing for the margin-aware parity tests: replace the token slot `slot` emitted last
 * (and will feed to the next step) by `token`. */
int ntts_backbone_debug_force(ntts_backbone* e, int32_t slot, int32_t token);
/* Timing: elapsed GPU milliseconds (hipEvents on the engine stream) of the most recent
 * ntts_backbone_decode call (all its steps) and of the most recent prefill. */
int ntts_backbone_last_timing(ntts_backbone* e, float* prefill_ms, float* decode_ms);
/* Replay ONE kernel of the decode step `iters` times at the current slot state and report its average
 * duration (hipEvents on the engine stream), its algorithmic HBM bytes per launch and how many times a
 * decode step launches it.  which: 0 paged attention (+RoPE/KV append), 1 QKV GEMM, 2 o_proj GEMM,
 * 3 gate/up GEMM (+SiLU*mul), 4 down_proj GEMM, 5 lm_head GEMM (+argmax partials), 6 add+RMSNorm.
 * Replay k runs on layer (k mod num_layers): consecutive replays touch different weights / KV pools, as
 * consecutive launches inside a decode step do, so the operands come from HBM and not from the Infinity Cache. */
int ntts_backbone_time_kernel(ntts_backbone* e, int32_t which, int32_t iters, float* avg_ms, double* alg_bytes,
                              int32_t* launches_per_step);
/* Diagnostics: launch the decode attention kernel of `layer` once at the current slot state and return the phase
 * timestamps it records: out[((slot * num_kv_heads + kv_head) * 4 + wave) * 8 + phase], 100 MHz ticks; phases: 0 kernel
 * entry, 1 slot state known, 2 RoPE / KV-append prologue done, 3 first K page through the matrix core, 4 scores done,
 * 5 softmax statistics merged, 6 PV done, 7 exit.  cap = entries available in `out` (>= max_batch * kv_heads * 32). */
int ntts_backbone_attn_timeline(ntts_backbone* e, int32_t layer, uint64_t* out, int64_t cap);
/* Diagnostics (tools/gemv_timeline.py): phase timestamps (100 MHz ticks) of ONE small-batch GEMV launch at the current slot state --
 * which = 1 QKV (+ RoPE + K append; slot 5 = K slices met in LDS, slot 6 = stores done), 2 o_proj, 3 gate/up, 4 down_proj of `layer`.  out[workgroup][16]: slots 0..5 of a workgroup's first feature wave (entry,
 * weights requested, weights landed, X panel complete, matrix-core chain done, stores done), slots 8..10 of its first helper wave (entry,
 * panel written, past the barrier).  cap >= 4096 * 16; *n_wg = workgroups that reported.  On a large-batch engine the same call times the tile kernels
 * (gemm.h / qkv_rope.h): slots 0..5 of wave 0 = entry, first ring slots requested, first tile landed, k-loop done, epilogue issued (qkv: K slices
 * met in LDS), stores drained.
 * which = 1 rewrites the residual ping-pong buffer out of sequence: do not continue generating from this slot state. */
int ntts_backbone_gemv_timeline(ntts_backbone* e, int32_t which, int32_t layer, uint64_t* out, int64_t cap, int32_t* n_wg);
/* Algorithmic HBM bytes of one decode step at the current slot lengths (SURVEY.md 8d formula:
 * layer weights + lm_head + sum_b len_b * kv_bytes_per_token + new-token KV writes). */
int ntts_backbone_step_bytes(ntts_backbone* e, double* bytes);

/* ------------------------------------------------------------------------------------------ */
/* Codec engine: NeuCodec decoder, codes -> 24 kHz waveform.  Replaces NeuCodec.decode_code behind  */
/* ref:neutts/neutts.py:288-291 (FSQ de-index -> Linear -> Conv/ResNet -> 12 x transformer ->       */
/* ResNet -> LayerNorm -> ISTFT head; hf:models/xcodec2/modeling_xcodec2.py:746-862).                */
/* ------------------------------------------------------------------------------------------ */
typedef struct ntts_codec ntts_codec;

typedef struct ntts_codec_config {
    int32_t hidden_size;        /* 1024 */
    int32_t intermediate_size;  /* 4096 */
    int32_t num_layers;         /* 12 */
    int32_t num_heads;          /* 16 */
    int32_t head_dim;           /* 64 */
    int32_t quantization_dim;   /* 2048 */
    int32_t n_levels;           /* 8 */
    int32_t levels[8];          /* FSQ levels, 4 each: 65 536 codes */
    int32_t hop_length;         /* 480 (ref:neutts/neutts.py:86); n_fft = 4 * hop */
    float rms_eps;              /* 1e-6 */
    int32_t max_frames;         /* longest utterance, in codec frames */
    int32_t max_rows;           /* workspace rows: sum over a decode call of (max frames of the call + 6) */
    int32_t precision;          /* ABI 9 (the field is ABI 8's; the numbering changed so that a zeroed struct gets the setting that holds the parity bar).
                                 * 0 = fp16 GEMM operands (DEFAULT): activations and weights as IEEE halves on v_mfma_f32_16x16x32_f16, fp32 accumulate --
                                 *     waveform within ~1e-3 RELATIVE rms (measured 9.5e-4) of the fp32 reference decoder (ref:neutts/neutts.py:288-291 runs it in fp32), i.e.
                                 *     inside BASELINE's 1e-3 absolute at any amplitude a [-1, 1] waveform can have; same matrix-core rate and bytes as
                                 *     bf16.  Operands must fit fp16's range: weights are checked at finalize (NTTS_EINVAL beyond 65504), activations
                                 *     saturate at +-65504 (post-norm activations are O(1-10));
                                 * 1 = "high": every GEMM operand as a split bf16 pair (hi + lo, K-concatenated [xh | xl | xh] x [wh | wh | wl]): ~16
                                 *     significant bits per operand at 3x the matrix-core work, bf16's range;
                                 * 2 = bf16 operands (rounds 1-5's default): 7e-3 RELATIVE rms, i.e. inside 1e-3 absolute only up to signal rms ~0.14;
                                 *     for weights outside fp16's range */
} ntts_codec_config;

const char* ntts_codec_last_error(const ntts_codec* c);
int ntts_codec_create(const ntts_codec_config* cfg, int device, ntts_codec** out);
void ntts_codec_destroy(ntts_codec* c);
/* `name` = parameter name of transformers' Xcodec2Model: "quantizer.project_out.{weight,bias}",
 * "decoder.fc.*", "decoder.embed.*", "decoder.{prior_net,post_net}.{0,1}.{norm1,conv1,norm2,conv2}.*",
 * "decoder.layers.{i}.{input_layernorm,post_attention_layernorm}.weight",
 * "decoder.layers.{i}.self_attn.{q,k,v,o}_proj.weight", "decoder.layers.{i}.mlp.{fc1,fc2}.weight",
 * "decoder.norm.*", "decoder.head.linear.*".  Host or device pointer, fp32 or bf16.  Blocking. */
int ntts_codec_load_tensor(ntts_codec* c, const char* name, const void* data, int dtype, const int64_t* shape,
                           int ndim, int is_device);
int ntts_codec_finalize(ntts_codec* c);
/* Decode `n` utterances: codes packed back to back (HOST int32, utterance i has lens[i] frames) ->
 * wav_out (HOST float32) row i = hop_length * lens[i] samples at offset i * wav_stride (wav_stride >= hop_length *
 * max(lens); the tail of a shorter row up to that length is scratch).  One strided D2H copy.  Blocking. */
int ntts_codec_decode(ntts_codec* c, int32_t n, const int32_t* codes, const int32_t* lens, float* wav_out,
                      int64_t wav_stride);
/* The same pass over codes that are already ON THE DEVICE (ntts_backbone_export_codes): utterance i's codes at
 * codes_dev + i * codes_stride, lens[i] of them (`lens` is a HOST array: the launch geometry depends on it).  Asynchronous:
 * the pass is ordered behind `producer_stream` (a hipStream_t, e.g. the backbone's; may be NULL), the waveforms are copied to
 * wav_out -- a DEVICE buffer (wav_on_device != 0) or page-locked host memory -- on the codec's own stream, and the call
 * returns once everything is enqueued; ntts_codec_sync() waits for it.  SURVEY.md 8b: device pointers + caller stream. */
int ntts_codec_decode_dev(ntts_codec* c, int32_t n, const int32_t* codes_dev, int32_t codes_stride, const int32_t* lens,
                          float* wav_out, int64_t wav_stride, int32_t wav_on_device, void* producer_stream);
/* Waits for the engine's stream; on a stream lent with ntts_codec_set_stream: for the most recent asynchronous pass and its hand-over
 * only (an event behind them), not for what the caller enqueued on that stream afterwards. */
int ntts_codec_sync(ntts_codec* c);
/* ABI 6.  Test taps (blocking), like ntts_encoder_read_stage: with keep_stages != 0 every decode call keeps the fp32 residual stream after
 * the stages of hf:models/xcodec2/modeling_xcodec2.py:838-862 -- 0 embed (fc + k = 7 conv :839-841), 1 prior_net (:845), 2 the transformer
 * layers (:855-856), 3 post_net (:859); read_stage copies utterance `utt`'s rows of the most recent call: HOST float32 [rows = frames][cols = hidden]. */
int ntts_codec_set_debug(ntts_codec* c, int32_t keep_stages);
int ntts_codec_read_stage(ntts_codec* c, int32_t stage, int32_t utt, float* out, int64_t cap, int32_t* rows, int32_t* cols);
/* ABI 6.  The codec engine's HIP stream (a hipStream_t) and its workspace limits (config fields max_frames / max_rows as created), for a
 * consumer that puts its own kernels before / behind a decode pass (ntts_streams_*). */
int ntts_codec_stream(ntts_codec* c, void** stream);
/* The codec passes on a stream of the caller's (as ntts_backbone_set_stream); NULL returns to the engine's own.  Blocking. */
int ntts_codec_set_stream(ntts_codec* c, void* stream);
int ntts_codec_limits(ntts_codec* c, int32_t* max_frames, int64_t* max_rows);
/* The same knob for the codec engine: re-create its stream restricted to the CUs of `mask` (n_words = 0: unrestricted).  Blocking. */
int ntts_codec_set_cu_mask(ntts_codec* c, const uint32_t* mask, int32_t n_words);
/* Page-locked host memory for wav_out: a pinned destination lets the D2H copy run at PCIe speed (pageable memory is
 * staged by the runtime at a fraction of it).  Plain malloc/free semantics; not tied to an engine. */
int ntts_host_alloc(size_t bytes, void** out);
int ntts_host_free(void* p);
/* (ABI 7) A non-blocking HIP stream on `device` (hipStreamCreateWithFlags) to lend to ntts_backbone_set_stream /
 * ntts_backbone_set_prefill_stream / ntts_codec_set_stream, for hosts that have no HIP runtime binding of their own (the Python
 * host creates the lanes of an engine gang with it); destroy drains it first.  No reference counterpart (one utterance, one stream). */
int ntts_stream_create(int32_t device, void** stream);
int ntts_stream_destroy(int32_t device, void* stream);
/* GPU milliseconds (hipEvents) of the most recent decode call, H2D/D2H excluded. */
int ntts_codec_last_timing(ntts_codec* c, float* ms);

/* ------------------------------------------------------------------------------------------ */
/* Device-side streaming (ABI 6): the per-chunk post-process of infer_stream for `n` concurrent    */
/* utterances -- token cache, window assembly, the 27-frame slice and the triangular cross-fade    */
/* with the previous chunk (ref:neutts/neutts.py:46-70, :385-388, :401-465) -- without the codes   */
/* or the decoded windows visiting the host: per burst only a few integers per stream travel up    */
/* and each stream's NEW samples come back.  Watermarking (ref :422-425, host library) is not      */
/* part of it: a caller with a watermarker keeps the host path.                                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct ntts_streams ntts_streams;
typedef struct ntts_stream_params {
    int32_t chunk;          /* streaming_frames_per_chunk  25   ref:neutts/neutts.py:88 */
    int32_t lookforward;    /* streaming_lookforward        5   ref :89 */
    int32_t lookback;       /* streaming_lookback          50   ref :90 */
    int32_t overlap;        /* streaming_overlap_frames     1   ref :87 */
    int32_t hop_length;     /* 480                              ref :86 */
    int32_t speech_base;    /* token id of "<|speech_0|>" */
    int32_t n_codes;        /* 65 536 */
    int32_t modulo;         /* != 0: every id becomes id mod n_codes (synthetic benchmark only, as ntts_backbone_export_codes) */
} ntts_stream_params;
const char* ntts_streams_last_error(const ntts_streams* s);
/* Stream i runs in decode slot slots[i] of `e` (already prefilled); its token cache starts with its ref_lens[i] reference codes (packed
 * back to back in ref_codes, HOST) -- ref :385-387 -- and can take max_new_tokens generated ones.  Windows are decoded by `c`. */
int ntts_streams_create(ntts_backbone* e, ntts_codec* c, const ntts_stream_params* prm, int32_t device, int32_t n, const int32_t* slots,
                        const int32_t* ref_codes, const int32_t* ref_lens, int32_t max_new_tokens, ntts_streams** out);
void ntts_streams_destroy(ntts_streams* s);
/* After a burst of decode steps has been ENQUEUED on the backbone: append the new codes to the caches and snapshot every stream's cache
 * length / finished flag behind that burst (asynchronous; the caller may enqueue the NEXT burst right away -- it runs beside pump_end). */
int ntts_streams_pump_begin(ntts_streams* s);
/* Optional: wait for that snapshot alone and report how many streams were still generating in it -- what a caller needs to know before
 * it enqueues the next burst (which then runs beside pump_end's codec passes). */
int ntts_streams_pump_wait(ntts_streams* s, int32_t* n_running);
/* Wait for that snapshot, decode every window it completes (one regular window per stream and round; a finished stream's final window
 * once no regular one is left), cross-fade, and hand out the new samples: chunk k belongs to stream chunk_stream[k], has
 * chunk_samples[k] samples at (*chunks) + k * (*row_stride) (page-locked memory owned by the set, valid until the next pump_end) and is
 * the stream's last one if chunk_last[k].  cap >= 2 n entries in the three HOST arrays.  *n_running = streams still generating;
 * *more = 1: further windows are already complete -- call pump_end again (without pump_begin) to get them.  Blocking. */
int ntts_streams_pump_end(ntts_streams* s, int32_t cap, int32_t* n_chunks, int32_t* chunk_stream, int32_t* chunk_samples, int32_t* chunk_last,
                          const float** chunks, int64_t* row_stride, int32_t* n_running, int32_t* more);
/* Streams that have emitted their last chunk.  (After a failed pump the set's bookkeeping is ahead of what was emitted: destroy it.) */
int ntts_streams_done(ntts_streams* s, int32_t* n_done);

/* ------------------------------------------------------------------------------------------ */
/* NeuCodec ENCODER engine: reference enrolment.  Replaces                                       */
/*     codec.encode_code(audio_or_path=wav16k[1,1,L]) -> int codes [1,1,T]                       */
/* (ref:neutts/neutts.py:266-271; neucodec is un-vendored, the architecture is restated from      */
/* hf:models/xcodec2/modeling_xcodec2.py:974-1049 = zero-pad to a hop multiple -> [kaldi fbank -> */
/* w2v-BERT 2.0 conformer layers 1..16 -> semantic adapter] || [acoustic conv encoder] -> concat  */
/* -> Linear -> FSQ).  One-off per speaker, off the synthesis hot path; fp32 throughout.          */
/* ------------------------------------------------------------------------------------------ */
typedef struct ntts_encoder ntts_encoder;

typedef struct ntts_encoder_config {
    int32_t sem_hidden;         /* 1024  w2v-BERT hidden size */
    int32_t sem_layers;         /* 16    conformer layers run (neucodec reads hidden_states[16]) */
    int32_t sem_heads;          /* 16 */
    int32_t sem_ffn;            /* 4096 */
    int32_t sem_conv_kernel;    /* 31    causal depthwise kernel */
    int32_t sem_left;           /* 64    relative_key distance clamp */
    int32_t sem_right;          /* 8 */
    float sem_ln_eps;           /* 1e-5 */
    int32_t ac_hidden;          /* 48    acoustic encoder base width */
    int32_t n_ratios;           /* 5 */
    int32_t ratios[8];          /* 2,2,4,4,5: 16 kHz -> 50 Hz (hop 320) */
    int32_t codec_hidden;       /* 1024  acoustic encoder output channels */
    int32_t n_levels;           /* 8 */
    int32_t levels[8];          /* FSQ levels, 4 each */
    int32_t max_samples;        /* longest clip (16 kHz samples) one encode call accepts */
} ntts_encoder_config;

const char* ntts_encoder_last_error(const ntts_encoder* e);
int ntts_encoder_create(const ntts_encoder_config* cfg, int device, ntts_encoder** out);
void ntts_encoder_destroy(ntts_encoder* e);
/* `name` = parameter name of transformers' Xcodec2Model: "semantic_encoder.feature_projection.*",
 * "semantic_encoder.encoder.layers.{i}.*", "semantic_adapter.conv{1..4}.*", "acoustic_encoder.*", "fc_encoder.*",
 * "quantizer.project_in.*".  Host or device pointer, fp32 or bf16 (widened to fp32).  Blocking. */
int ntts_encoder_load_tensor(ntts_encoder* e, const char* name, const void* data, int dtype, const int64_t* shape,
                             int ndim, int is_device);
/* Lists every missing tensor in the error text; repacks Conv1d weights for the implicit GEMM. */
int ntts_encoder_finalize(ntts_encoder* e);
/* wav: HOST float32 mono at 16 kHz, n_samples of them -> codes_out (HOST int32, capacity `cap`), *n_codes =
 * n_samples / hop + 1 (the reference appends one zero, then pads to a hop multiple).  Blocking. */
int ntts_encoder_encode(ntts_encoder* e, const float* wav, int64_t n_samples, int32_t* codes_out, int32_t cap, int32_t* n_codes);
/* Stage outputs of the most recent encode call, for the per-stage parity tests: 0 = fbank features [T,160], 1 = semantic
 * adapter || acoustic encoder [T, sem_hidden + codec_hidden], 2 = fc_encoder output [T, same], 3 = twice-bounded FSQ
 * latents [T, n_levels].  HOST float32 out of capacity `cap` floats. */
int ntts_encoder_read_stage(ntts_encoder* e, int32_t stage, float* out, int64_t cap, int32_t* rows, int32_t* cols);
/* GPU milliseconds (hipEvents) of the most recent encode call, H2D/D2H excluded. */
int ntts_encoder_last_timing(ntts_encoder* e, float* ms);

/* ------------------------------------------------------------------------------------------ */
/* Kernel-level entry points (used by the parity tests and micro-benchmarks; all pointers are   */
/* DEVICE pointers, bf16 unless noted, row-major; run on the NULL stream and block).           */
/* ------------------------------------------------------------------------------------------ */
/* C[M,N] = A[M,K] (lda) * W[N,K]^T (+ bias[N]); fp32 accumulate on MFMA, one RNE rounding to bf16.
 * variant: 0 = auto, 1 = 128x128 tile, 2 = 64x64 tile, 3 = 64x64 split-K slabs + reduce, 4 = 256x256 tile. */
int ntts_k_gemm_bf16(const void* A, int64_t lda, const void* W, const void* bias, void* C, int64_t ldc,
                     int32_t M, int32_t N, int32_t K, int32_t variant);
/* fp8 probes (tests): out[i] = e4m3(clamp(in[i] * inv_scale, +-448)), round-to-nearest-even -- the conversion every fp8
 * producer on the path uses; and C[M,N] = bf16(fma(A[M,K] . W[N,K]^T, xscale * wscale[n], bias[n])) with e4m3 BYTE operands
 * (row-major, K % 128 == 0) on v_mfma_f32_16x16x32_fp8_fp8.  variant: 2 = 64x64 tile, 4 = 256x256, else 128x128. */
int ntts_k_fp8_quantize(const float* in_dev, void* out_dev, int64_t n, float inv_scale);
int ntts_k_gemm_fp8(const void* A, const void* W, const float* wscale, float xscale, const void* bias, void* C,
                    int32_t M, int32_t N, int32_t K, int32_t variant);
/* y = rmsnorm(x) * w with Qwen2RMSNorm's rounding (hf:models/qwen2/modeling_qwen2.py:247-252). */
int ntts_k_rmsnorm_bf16(const void* x, const void* w, void* y, int32_t rows, int32_t cols, float eps);
/* Micro-benchmark of one GEMM tile configuration on synthetic operands (tools/ubench_gemm.py); see csrc/kapi.cpp. */
int ntts_k_gemm_probe(int32_t M, int32_t N, int32_t K, int32_t config, int32_t abl, int32_t copies, int32_t iters,
                      double* us);
/* Achieved HBM copy bandwidth probe: copies `bytes` device->device `iters` times, returns GB/s. */
int ntts_k_membw(size_t bytes, int32_t iters, double* gbps);
/* Diagnostics: writes 3 x 64 x 4 floats describing the MFMA 16x16x32 lane layout (see csrc/kapi.cpp). */
int ntts_k_mfma_probe(float* out_dev_768);
/* Launch-chain floor (diagnostics): captures `n_kernels` dependent launches of a kernel that only reads 4 bytes per workgroup
 * (`grid` workgroups of 256 threads; block = 256) into one hipGraph, replays it `iters` times and returns the microseconds per
 * replay -- what a decode step of that many launches costs before any kernel moves a byte.  block = -256: every thread of every
 * launch also loads 16 HBM-cold bytes and stores 16 (the least a kernel of the step does: one dependent round trip + a store). */
int ntts_k_launch_chain_probe(int32_t n_kernels, int32_t grid, int32_t block, int32_t iters, double* us_per_chain);
/* SiLU as the GEMM epilogues compute it (hf:activations.py SiLUActivation), elementwise on DEVICE bf16 values: out[i] =
 * bf16(silu(in[i])).  variant 0 = x / (1 + expf(-x)), 1 = the epilogues' fast form (gemm.h silu_fast); the parity tests run
 * every bf16 bit pattern through both and compare with torch. */
int ntts_k_silu_probe(const void* in_bf16_dev, void* out_bf16_dev, int64_t n, int32_t variant);

#ifdef __cplusplus
}
#endif
#endif /* NEUTTS_HIP_H */
