// sample.h -- token choice + per-slot generation bookkeeping, on device (so a decode step is replayable
// as a hipGraph: nothing about a step depends on host-side state).
//
// Replaces the tail of GenerationMixin._sample (hf:generation/utils.py:2894-2941):
//   fp32 logits -> MinNewTokensLength (applied in the lm_head epilogue via mask_eos) -> argmax
//   (first max wins, torch.argmax) -> append -> EosTokenCriteria / MaxLengthCriteria.
#pragma once
#include <ntts/dev.h>

namespace ntts {

enum SlotState { SLOT_FREE = 0, SLOT_RUNNING = 1, SLOT_FINISHED = 2, SLOT_PREFILLED = 3 };

struct SlotArrays {      // device arrays, one entry per decode slot
    int* state;
    int* pos;            // tokens in the KV cache == position of cur_tok
    int* n_new;          // generated so far
    int* cur_tok;        // token fed to the next step
    int* prompt_len;
    int* min_new;
    int* max_len;
    int* eos;
    int* mask_eos;       // eos+1 while EOS is masked for the NEXT sampled token (n_new < min_new), else 0
    int* out_tokens;     // [slots][out_stride]
    int out_stride;
};

struct SampleArgs {
    const float* part_val;   // [rows][n_part] per-row partial maxima from the lm_head epilogue
    const int* part_idx;
    int n_part;
    SlotArrays sl;
    int phase;               // SLOT_RUNNING: decode step; SLOT_PREFILLED: first token after prefill
};

NTTS_KERNEL(256) void sample_greedy_kernel(SampleArgs p) {
    NTTS_SHARED float sv[4];
    NTTS_SHARED int si[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int i = tid; i < p.n_part; i += 256) {
        const float v = p.part_val[(long)b * p.n_part + i];
        const int ix = p.part_idx[(long)b * p.n_part + i];
        if (v > best || (v == best && ix < bidx)) { best = v; bidx = ix; }
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
        const float ov = shfl_xor(best, sh);
        const int oi = shfl_xor(bidx, sh);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lane_id() == 0) { sv[wave_id()] = best; si[wave_id()] = bidx; }
    sync();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bidx)) { best = sv[w]; bidx = si[w]; }
        SlotArrays& s = p.sl;
        if (s.state[b] == p.phase) {
            const int tok = bidx;
            const int n = (p.phase == SLOT_PREFILLED) ? 0 : s.n_new[b];
            s.out_tokens[(long)b * s.out_stride + n] = tok;
            s.n_new[b] = n + 1;
            if (p.phase == SLOT_RUNNING) s.pos[b] = s.pos[b] + 1;
            s.cur_tok[b] = tok;
            s.mask_eos[b] = (n + 1 < s.min_new[b]) ? s.eos[b] + 1 : 0;
            const bool fin = (tok == s.eos[b]) || (s.prompt_len[b] + n + 1 >= s.max_len[b]);
            s.state[b] = fin ? SLOT_FINISHED : SLOT_RUNNING;
        }
    }
}

// ---- small host->device state plumbing kernels -------------------------------------------------
struct PrefillInit {
    const int* slot;       // [n]
    const int* seq_len;
    const int* min_new;
    const int* max_len;
    const int* eos;
    const int* bt_rows;    // [n][max_pages]
    int* block_table;      // [slots][max_pages]
    int max_pages;
    int n;
    SlotArrays sl;
};
NTTS_KERNEL(64) void prefill_init_kernel(PrefillInit p) {
    const int i = blockIdx.x;
    const int s = p.slot[i];
    for (int k = threadIdx.x; k < p.max_pages; k += 64) p.block_table[(long)s * p.max_pages + k] = p.bt_rows[(long)i * p.max_pages + k];
    if (threadIdx.x == 0) {
        p.sl.state[s] = SLOT_PREFILLED;
        p.sl.pos[s] = p.seq_len[i];
        p.sl.n_new[s] = 0;
        p.sl.prompt_len[s] = p.seq_len[i];
        p.sl.min_new[s] = p.min_new[i];
        p.sl.max_len[s] = p.max_len[i];
        p.sl.eos[s] = p.eos[i];
        p.sl.mask_eos[s] = p.min_new[i] > 0 ? p.eos[i] + 1 : 0;
        p.sl.cur_tok[s] = 0;
    }
}
NTTS_KERNEL(64) void bt_update_kernel(const int* trip, int n, int* block_table, int max_pages) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) block_table[(long)trip[3 * i] * max_pages + trip[3 * i + 1]] = trip[3 * i + 2];
}

// dst[dst_row(r)][:] = bf16(src[r][:])  -- weight packing (fp32 -> bf16 RNE like `.to(bfloat16)`)
NTTS_KERNEL(256) void pack_rows_kernel(const void* src, int src_is_f32, bf16_t* dst, const int* dst_rows, long cols) {
    const long r = blockIdx.x;
    const long dr = dst_rows ? dst_rows[r] : r;
    for (long c = threadIdx.x; c < cols; c += 256)
        dst[dr * cols + c] = src_is_f32 ? f2bf(((const float*)src)[r * cols + c]) : ((const bf16_t*)src)[r * cols + c];
}

}  // namespace ntts
