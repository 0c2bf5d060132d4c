// sample.h -- token choice + per-slot generation bookkeeping, on device (so a decode step is replayable
// as a hipGraph: nothing about a step depends on host-side state).
//
// Replaces the tail of GenerationMixin._sample (hf:generation/utils.py:2894-2941):
//   fp32 logits -> MinNewTokensLength (applied in the lm_head epilogue via mask_eos) ->
//     do_sample=0: argmax (first max wins, torch.argmax)
//     do_sample=1: [TemperatureLogitsWarper: scores / T] -> TopKLogitsWarper (scores < k-th largest -> -inf, ties at the
//                  k-th value kept; hf:generation/logits_process.py:542-595) -> softmax -> multinomial(1)
//                  (the reference's own call: do_sample=True, temperature=1.0, top_k=50, ref:neutts/neutts.py:338-347)
//   -> append -> EosTokenCriteria / MaxLengthCriteria.
// Sampling reads the row of bf16 logits the lm_head epilogue leaves behind (HF's logits ARE bf16 values cast to fp32, so a
// 16-bit radix select finds the exact k-th largest), draws its uniform from Philox4x32-10 keyed by the request's seed with
// the step index as counter: reproducible for a given seed whatever slot / batch the request runs in (callers give each
// request its own seed); torch's own generator stream is not reproduced.
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
    int* top_k;          // 0 = greedy (do_sample=0); k >= 1 = sample among the k largest logits
    float* temperature;  // > 0
    unsigned int* seed;  // [slots][2] Philox key
    int* out_tokens;     // [slots][out_stride]
    int out_stride;
};

struct SampleArgs {
    const float* part_val;   // [rows][n_part] per-row partial maxima from the lm_head epilogue
    const int* part_idx;
    int n_part;
    int part_width;          // columns per partial (consecutive: partial i covers columns [i * part_width, (i + 1) * part_width)); 0 = unknown
    const bf16_t* logits;    // [rows][ld_logits] processed bf16 logits (EOS mask applied); null if no slot samples
    long ld_logits;
    int vocab;
    SlotArrays sl;
    int phase;               // SLOT_RUNNING: decode step; SLOT_PREFILLED: first token after prefill
    // compacted head (ntts_backbone_set_logits_range): column c of the lm_head is token id_base + c for c < n_range and the EOS id for
    // c == n_range; n_range = 0: column = token id
    int n_range, id_base, id_tail;
};

constexpr int kSampleCap = 512;   // candidates kept (k plus ties at the k-th value, capped)

// Philox4x32-10 (Salmon et al. 2011), one block: 128-bit counter, 64-bit key -> 4 x 32 random bits
NTTS_D void philox4x32(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned int)p1;
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned int)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// order-preserving 16-bit key of a bf16 value (larger value <-> larger key; -inf smallest of the non-NaN keys)
NTTS_D unsigned int bf16_key(bf16_t v) { return (v & 0x8000u) ? (~(unsigned int)v & 0xFFFFu) : ((unsigned int)v | 0x8000u); }

// top-k + multinomial for one row; all 256 threads of the block take part.  Returns the token in every thread.
// The lm_head epilogue leaves one maximum per group of `gw` consecutive columns (pv[0 .. n_part-1]; SampleArgs::part_val).  The k-th
// largest of those maxima, T, is a lower bound of the k-th largest logit (k groups hold an element >= T), and a group whose maximum is
// below T holds no element >= T: only the groups with maximum >= T -- k of them plus ties, ~50 x 96 columns of the 217 488 -- are
// scanned for the exact threshold and the survivors.  Same result as three sweeps over the whole row (the fallback when a row has more
// than kGroupCap such groups, e.g. constant logits, or when there are fewer groups than k), 1/45 of the bytes and LDS atomics.
constexpr int kGroupCap = 1024;
NTTS_D int sample_topk_row(const bf16_t* row, int V, int k, float temperature, unsigned int s0, unsigned int s1,
                           unsigned int step, const float* pv, int n_part, int gw) {
    NTTS_SHARED unsigned int hist[256];
    NTTS_SHARED unsigned int sel[6];          // [0] high byte, [1] elements above that bin, [2] threshold key, [3] list length, [4] group threshold key, [5] groups kept
    NTTS_SHARED int cidx[kSampleCap];
    NTTS_SHARED unsigned short cval[kSampleCap];
    NTTS_SHARED int sidx[kSampleCap];
    NTTS_SHARED unsigned short sval[kSampleCap];
    NTTS_SHARED int glist[kGroupCap];
    NTTS_SHARED unsigned int wsum[4];
    NTTS_SHARED float ev[kSampleCap];
    NTTS_SHARED float wmax[4];
    NTTS_SHARED int result;
    const int tid = threadIdx.x;
    if (k > V) k = V;
    if (k > kSampleCap) k = kSampleCap;
    // Walking the 256 bins from the top, the bin at which the running count (starting from `base`) first reaches k, and the count above
    // it -> sel[0], sel[1].  All threads call it; thread t owns bin t: suffix sums by wave shuffles instead of one thread's 256 dependent
    // LDS reads (the two serial walks of a select were ~8 us of a 62 us kernel).  Bin 0 takes the row if no higher bin reaches k.
    auto find_bin = [&](unsigned int base) {
        const int lane = lane_id(), w = wave_id();
        const unsigned int h = hist[tid];
        int incl = (int)h;                                 // bins tid .. end of this wave's 64
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = shfl(incl, (lane + d) & 63);
            if (lane + d < 64) incl += o;
        }
        if (lane == 0) wsum[w] = (unsigned int)incl;
        sync();
        unsigned int excl = base + (unsigned int)incl - h;  // count strictly above bin tid
        for (int ww = w + 1; ww < 4; ++ww) excl += wsum[ww];
        if (tid == 0 ? excl < (unsigned int)k : (excl < (unsigned int)k && excl + h >= (unsigned int)k)) { sel[0] = (unsigned int)tid; sel[1] = excl; }
        sync();
    };
    // the k-th largest of `count` keys (two histogram passes over 16-bit keys): key_at(i) = key of element i or 0x10000 to skip it;
    // lo_bound: only keys >= lo_bound are counted.  Leaves the key in sel[out]
    auto kth_key = [&](auto&& for_each_key, int out) {
        hist[tid] = 0;
        sync();
        for_each_key([&](unsigned int key) { atomic_add_lds(&hist[key >> 8], 1u); });
        sync();
        find_bin(0u);
        const unsigned int hb = sel[0], above = sel[1];
        hist[tid] = 0;
        sync();
        for_each_key([&](unsigned int key) { if ((key >> 8) == hb) atomic_add_lds(&hist[key & 255u], 1u); });
        sync();
        find_bin(above);
        if (tid == 0) sel[out] = (hb << 8) | sel[0];
        sync();
    };
    if (tid < 6) sel[tid] = 0;
    sync();
    // ---- which column groups can hold one of the k largest logits
    bool grouped = pv != nullptr && gw >= 8 && (gw & 7) == 0 && n_part >= k;
    if (grouped) {
        kth_key([&](auto&& f) { for (int i = tid; i < n_part; i += 256) f(bf16_key(f2bf(pv[i]))); }, 4);
        const unsigned int gthr = sel[4];
        for (int i = tid; i < n_part; i += 256)
            if (bf16_key(f2bf(pv[i])) >= gthr) {
                const unsigned int at = atomic_add_lds(&sel[5], 1u);
                if (at < (unsigned int)kGroupCap) glist[at] = i;
            }
        sync();
        if (sel[5] > (unsigned int)kGroupCap) grouped = false;      // (block-uniform)
    }
    const int ng = grouped ? (int)sel[5] : 0;
    const unsigned int floor_key = grouped ? sel[4] : 0u;            // elements below the group threshold cannot be among the k largest
    const int vpg = gw >> 3;                                         // 16-byte vectors per group
    const int nvec = V >> 3;                                         // full row: 16-byte vectors; the scalar tail is handled by the first threads
    // every element (index, value) that can matter: the listed groups, or the whole row
    auto for_each_elem = [&](auto&& f) {
        if (grouped) {
            for (int i = tid; i < ng * vpg; i += 256) {
                const int c0 = glist[i / vpg] * gw + (i % vpg) * 8;
                if (c0 + 8 <= V) {
                    const bf16x8 v = ld16<bf16x8>(row + c0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f(c0 + e, (bf16_t)v[e]);
                } else {
                    for (int e = 0; e < 8 && c0 + e < V; ++e) f(c0 + e, row[c0 + e]);
                }
            }
        } else {
            for (int i = tid; i < nvec; i += 256) {
                const bf16x8 v = ld16<bf16x8>(row + (long)i * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) f(i * 8 + e, (bf16_t)v[e]);
            }
            for (int i = nvec * 8 + tid; i < V; i += 256) f(i, row[i]);
        }
    };
    // ---- the exact k-th largest logit (16-bit radix select: HF's logits ARE bf16 values)
    kth_key([&](auto&& f) { for_each_elem([&](int, bf16_t v) { const unsigned int key = bf16_key(v); if (key >= floor_key) f(key); }); }, 2);
    const unsigned int thr = sel[2];
    // ---- gather every logit >= the k-th largest (ties kept, like scores < kth -> -inf)
    for_each_elem([&](int idx, bf16_t v) {
        if (bf16_key(v) >= thr) {
            const unsigned int at = atomic_add_lds(&sel[3], 1u);
            if (at < (unsigned int)kSampleCap) { cidx[at] = idx; cval[at] = v; }
        }
    });
    sync();
    int n = (int)sel[3];
    if (n > kSampleCap) n = kSampleCap;
    // ---- order the candidates by token id (the append order above is not deterministic): rank sort
    for (int a = tid; a < n; a += 256) {
        const int ia = cidx[a];
        int rank = 0;
        for (int b = 0; b < n; ++b) rank += cidx[b] < ia;
        sidx[rank] = ia;
        sval[rank] = cval[a];
    }
    sync();
    // ---- softmax over the survivors + inverse-CDF draw: maximum and exponentials in parallel, the two running sums by one thread in
    //      token order (the draw depends on the order of those additions: kept)
    float mx = -INFINITY;
    for (int a = tid; a < n; a += 256) mx = fmaxf(mx, bf2f(sval[a]));
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) mx = fmaxf(mx, shfl_xor(mx, sh));
    if (lane_id() == 0) wmax[wave_id()] = mx;
    sync();
    const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const float it = 1.0f / temperature;
    for (int a = tid; a < n; a += 256) ev[a] = fexp((bf2f(sval[a]) - m) * it);
    sync();
    if (tid == 0) {
        float total = 0.f;
        for (int a = 0; a < n; ++a) total += ev[a];
        unsigned int c[4] = {step, 0u, 0u, 0u};
        philox4x32(c, s0, s1);
        const float u = (float)(c[0] >> 8) * (1.0f / 16777216.0f);   // [0, 1)
        const float target = u * total;
        float acc = 0.f;
        int pick = sidx[n - 1];
        for (int a = 0; a < n; ++a) {
            acc += ev[a];
            if (acc > target) { pick = sidx[a]; break; }
        }
        result = pick;
    }
    sync();
    return result;
}

NTTS_KERNEL(256) void sample_greedy_kernel(SampleArgs p) {
    NTTS_SHARED float sv[4];
    NTTS_SHARED int si[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    // The lm_head leaves n_part (max, first index) pairs per row (2 268 - 3 400 at NeuTTS-Air's vocabulary): 9 - 14 per thread.  A
    // plain loop is a chain of that many dependent round trips (the compare carries `best`): 9.1 us per decode step.  Here
    // the requests of kU iterations are in flight together, branch-free (an index past the end re-reads the last pair, which
    // changes nothing: max value / lowest index is idempotent).
    constexpr int kU = 8;
    const float* pv = p.part_val + (long)b * p.n_part;
    const int* pi = p.part_idx + (long)b * p.n_part;
    for (int i0 = tid; i0 < p.n_part; i0 += 256 * kU) {
        float v[kU];
        int ix[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = i0 + u * 256 < p.n_part ? i0 + u * 256 : p.n_part - 1;
            v[u] = pv[i];
            ix[u] = pi[i];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (v[u] > best || (v[u] == best && ix[u] < bidx)) { best = v[u]; bidx = ix[u]; }
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
        const float ov = shfl_xor(best, sh);
        const int oi = shfl_xor(bidx, sh);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lane_id() == 0) { sv[wave_id()] = best; si[wave_id()] = bidx; }
    sync();
    const bool live = p.sl.state[b] == p.phase;                     // block-uniform
    const int k = (live && p.logits) ? p.sl.top_k[b] : 0;
    int sampled = -1;
    if (k > 0) {
        const int step = (p.phase == SLOT_PREFILLED) ? 0 : p.sl.n_new[b];
        sampled = sample_topk_row(p.logits + (long)b * p.ld_logits, p.vocab, k, p.sl.temperature[b], p.sl.seed[2 * b],
                                  p.sl.seed[2 * b + 1], (unsigned int)step, pv, p.n_part, p.part_width);
    }
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bidx)) { best = sv[w]; bidx = si[w]; }
        SlotArrays& s = p.sl;
        if (live) {
            int tok = k > 0 ? sampled : bidx;
            if (p.n_range > 0) tok = tok < p.n_range ? tok + p.id_base : p.id_tail;
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
    const int* top_k;      // 0 = greedy
    const int* temp_bits;  // float bits
    const int* seed;       // [n][2]
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
        p.sl.top_k[s] = p.top_k[i];
        p.sl.temperature[s] = __builtin_bit_cast(float, p.temp_bits[i]);
        p.sl.seed[2 * s] = (unsigned int)p.seed[2 * i];
        p.sl.seed[2 * s + 1] = (unsigned int)p.seed[2 * i + 1];
        p.sl.mask_eos[s] = p.min_new[i] > 0 ? p.eos[i] + 1 : 0;
        p.sl.cur_tok[s] = 0;
    }
}
NTTS_KERNEL(64) void zero_slots_kernel(const int* slots, int n, int* state) {   // ntts_backbone_release_many: state[slots[i]] = FREE
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) state[slots[i]] = 0;
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

// GEMM weight packing: logical row r of the source lands in row dr = row0 + (dst_rows ? dst_rows[r] : r) of the packed matrix
// [*, cols]; tile_major = 1 stores it "tile-major" (gemm.h GemmArgs::w_tile_major): groups of 64 rows, inside a group the
// 64 x 64 blocks of consecutive K tiles follow each other, so a workgroup's weight stream is one sequential run of addresses
NTTS_KERNEL(256) void pack_weight_kernel(const void* src, int src_is_f32, bf16_t* dst, const int* dst_rows, long row0, long cols,
                                         int tile_major) {
    const long r = blockIdx.x;
    const long dr = row0 + (dst_rows ? dst_rows[r] : r);
    for (long c = threadIdx.x; c < cols; c += 256) {
        const bf16_t v = src_is_f32 ? f2bf(((const float*)src)[r * cols + c]) : ((const bf16_t*)src)[r * cols + c];
        const long at = tile_major ? (dr >> 6) * 64 * cols + (c >> 6) * 4096 + (dr & 63) * 64 + (c & 63) : dr * cols + c;
        dst[at] = v;
    }
}

// Compacted lm_head (ntts_backbone_set_logits_range): row r of dst = row rows[r] of src, both tile-major in 64-row x 128-byte blocks
// (esz = 2: 64 bf16 per block row, esz = 1: 128 e4m3); one workgroup per destination row, 16 bytes per thread and k-tile; the fp8 head's
// per-row scales travel along
NTTS_KERNEL(64) void gather_head_rows_kernel(const unsigned char* src, unsigned char* dst, const int* rows, long K_bytes, const float* sc_src, float* sc_dst) {
    const long r = blockIdx.x, sr = rows[r];
    const long ktiles = K_bytes / 128;
    for (long i = threadIdx.x; i < ktiles * 8; i += 64) {
        const long kt = i >> 3, c = i & 7;
        *(u32x4*)(dst + (r >> 6) * 64 * K_bytes + kt * 8192 + (r & 63) * 128 + c * 16) =
            *(const u32x4*)(src + (sr >> 6) * 64 * K_bytes + kt * 8192 + (sr & 63) * 128 + c * 16);
    }
    if (sc_src && threadIdx.x == 0) sc_dst[r] = sc_src[sr];
}
// running max |x| over bf16 rows (fp8 calibration, ntts_backbone_calibrate): 8 values per thread and iteration; the bit pattern of a
// non-negative float orders like the value, so the combine is an integer atomic max
NTTS_KERNEL(256) void amax_bf16_kernel(const bf16_t* x, long nvec, float* out) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const bf16x8 v = ld16<bf16x8>(x + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = __builtin_fabsf(bf2f((bf16_t)v[e]));
            m = a > m ? a : m;             // (NaN never wins)
        }
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) { const float o = shfl_xor(m, sh); m = o > m ? o : m; }
    if (lane_id() == 0) atomic_max_global_u32((unsigned int*)out, __builtin_bit_cast(unsigned int, m));
}

// ---- parking (ntts_backbone_activate): the per-slot state of parked request src[i] becomes decode slot dst[i]'s; one workgroup per pair
struct ActivateArgs {
    const int* pairs;        // [n][2] = (parking row, decode slot)
    SlotArrays sl;
    int* block_table;        // [slots][max_pages]
    int max_pages;
};
NTTS_KERNEL(64) void activate_slots_kernel(ActivateArgs p) {
    const int src = p.pairs[2 * blockIdx.x], dst = p.pairs[2 * blockIdx.x + 1];
    const int lane = threadIdx.x;
    const int nn = p.sl.n_new[src];
    for (int k = lane; k < nn; k += 64) p.sl.out_tokens[(long)dst * p.sl.out_stride + k] = p.sl.out_tokens[(long)src * p.sl.out_stride + k];
    for (int k = lane; k < p.max_pages; k += 64) p.block_table[(long)dst * p.max_pages + k] = p.block_table[(long)src * p.max_pages + k];
    if (lane == 0) {
        p.sl.pos[dst] = p.sl.pos[src]; p.sl.n_new[dst] = nn; p.sl.cur_tok[dst] = p.sl.cur_tok[src]; p.sl.prompt_len[dst] = p.sl.prompt_len[src];
        p.sl.min_new[dst] = p.sl.min_new[src]; p.sl.max_len[dst] = p.sl.max_len[src]; p.sl.eos[dst] = p.sl.eos[src]; p.sl.mask_eos[dst] = p.sl.mask_eos[src];
        p.sl.top_k[dst] = p.sl.top_k[src]; p.sl.temperature[dst] = p.sl.temperature[src];
        p.sl.seed[2 * dst] = p.sl.seed[2 * src]; p.sl.seed[2 * dst + 1] = p.sl.seed[2 * src + 1];
        p.sl.state[dst] = p.sl.state[src];
        p.sl.state[src] = SLOT_FREE;
    }
}

// ids -> codec codes on the device (ref:neutts/neutts.py:349 tokenizer.decode + :276 regex, as one pass): of slot s's new ids
// keep those in [speech_base, speech_base + n_codes), as id - speech_base, in order; `modulo` (synthetic benchmark only: random
// weights do not stay in the speech range) maps every id to id mod n_codes instead.  One workgroup per utterance.
struct ExportCodesArgs {
    const int* slots;       // [n] decode slots
    SlotArrays sl;
    int speech_base, n_codes, modulo;
    int* codes;             // [n][stride]
    int stride;
    int* lens;              // [n] number of codes written
};
NTTS_KERNEL(256) void export_codes_kernel(ExportCodesArgs p) {
    NTTS_SHARED int wcount[4];
    NTTS_SHARED int base_s;
    const int u = blockIdx.x, s = p.slots[u], tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int n = p.sl.n_new[s];
    const int* ids = p.sl.out_tokens + (long)s * p.sl.out_stride;
    if (tid == 0) base_s = 0;
    sync();
    for (int t0 = 0; t0 < n; t0 += 256) {      // order-preserving compaction, 256 ids per round
        const int t = t0 + tid;
        int code = -1;
        if (t < n) {
            const int id = ids[t];
            if (p.modulo) code = id % p.n_codes;
            else if (id >= p.speech_base && id < p.speech_base + p.n_codes) code = id - p.speech_base;
        }
        const unsigned long long m = ballot(code >= 0);
        const int before = popc64(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[w] = popc64(m);
        sync();
        int off = base_s;
        for (int k = 0; k < w; ++k) off += wcount[k];
        if (code >= 0 && off + before < p.stride) p.codes[(long)u * p.stride + off + before] = code;
        sync();
        if (tid == 0) base_s += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        sync();
    }
    if (tid == 0) p.lens[u] = base_s < p.stride ? base_s : p.stride;
}

// The same hand-off for STREAMS (ref:neutts/neutts.py:390-404: every generated "<|speech_N|>" token is appended to the stream's token
// cache as it arrives): of slot s's ids generated since the previous call (seen[u]), the speech codes are appended to the stream's
// device-side code cache behind its clen[u] entries; seen / clen are advanced and the slot's state is reported.  One workgroup per stream.
struct AppendCodesArgs {
    const int* slots;       // [n] decode slots
    SlotArrays sl;
    int speech_base, n_codes, modulo;
    int* cache;             // [n][stride] reference codes + generated codes so far
    int stride;
    int* clen;              // [n] codes in cache[u]
    int* seen;              // [n] generated ids already looked at
    int* fin;               // [n] out: 1 once the slot has finished (EOS / max_length) -- with everything it generated appended
};
NTTS_KERNEL(256) void append_codes_kernel(AppendCodesArgs p) {
    NTTS_SHARED int wcount[4];
    NTTS_SHARED int base_s;
    const int u = blockIdx.x, s = p.slots[u], tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int st = p.sl.state[s];
    const int n = (st == SLOT_RUNNING || st == SLOT_FINISHED) ? p.sl.n_new[s] : 0;
    const int from = p.seen[u];
    const int* ids = p.sl.out_tokens + (long)s * p.sl.out_stride;
    if (tid == 0) base_s = p.clen[u];
    sync();
    for (int t0 = from; t0 < n; t0 += 256) {      // order-preserving compaction, 256 ids per round (export_codes_kernel)
        const int t = t0 + tid;
        int code = -1;
        if (t < n) {
            const int id = ids[t];
            if (p.modulo) code = id % p.n_codes;
            else if (id >= p.speech_base && id < p.speech_base + p.n_codes) code = id - p.speech_base;
        }
        const unsigned long long m = ballot(code >= 0);
        const int before = popc64(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[w] = popc64(m);
        sync();
        int off = base_s;
        for (int k = 0; k < w; ++k) off += wcount[k];
        if (code >= 0 && off + before < p.stride) p.cache[(long)u * p.stride + off + before] = code;
        sync();
        if (tid == 0) base_s += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        sync();
    }
    if (tid == 0) {
        p.clen[u] = base_s < p.stride ? base_s : p.stride;
        p.seen[u] = n > from ? n : from;
        p.fin[u] = st == SLOT_FINISHED ? 1 : 0;
    }
}

// fp8 model: one source row -> e4m3 bytes + its scale.  The row is first rounded to bf16 (what the bf16 checkpoint holds), then
// scale = max|w| / 448 (1 for an all-zero row), w_q = e4m3(w / scale) -- per output channel, as static-fp8 checkpoints store
// them.  Layout of the byte matrix: tile-major in 64-row x 128-byte blocks (gemm.h F8).
NTTS_KERNEL(256) void pack_weight_fp8_kernel(const void* src, int src_is_f32, unsigned char* dst, float* scales, const int* dst_rows,
                                             long row0, long cols) {
    NTTS_SHARED float red[4];
    const long r = blockIdx.x;
    const long dr = row0 + (dst_rows ? dst_rows[r] : r);
    auto val = [&](long c) { return src_is_f32 ? rbf(((const float*)src)[r * cols + c]) : bf2f(((const bf16_t*)src)[r * cols + c]); };
    float am = 0.f;
    for (long c = threadIdx.x; c < cols; c += 256) am = fmaxf(am, fabsf(val(c)));
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) am = fmaxf(am, shfl_xor(am, sh));
    if (lane_id() == 0) red[wave_id()] = am;
    sync();
    am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = am > 0.f ? am / kFp8Max : 1.0f;
    if (threadIdx.x == 0) scales[dr] = scale;
    for (long c = threadIdx.x; c < cols; c += 256)
        dst[(dr >> 6) * 64 * cols + (c >> 7) * 8192 + (dr & 63) * 128 + (c & 127)] = f2fp8c(val(c) / scale);
}

// pre-quantised fp8 checkpoints: the matrix arrives as e4m3 bytes [rows][cols] and is stored as it is (same layout as above);
// its per-output-channel scales arrive separately (one fp32 per row, or one for the whole matrix: `broadcast`)
NTTS_KERNEL(256) void pack_weight_fp8_raw_kernel(const unsigned char* src, unsigned char* dst, const int* dst_rows, long row0, long cols) {
    const long r = blockIdx.x;
    const long dr = row0 + (dst_rows ? dst_rows[r] : r);
    for (long c = threadIdx.x; c < cols; c += 256)
        dst[(dr >> 6) * 64 * cols + (c >> 7) * 8192 + (dr & 63) * 128 + (c & 127)] = src[r * cols + c];
}
NTTS_KERNEL(256) void scatter_scales_kernel(const float* src, int broadcast, float* dst, const int* dst_rows, long row0, long n) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r < n) dst[row0 + (dst_rows ? dst_rows[r] : r)] = src[broadcast ? 0 : r];
}

// number of elements of `src` ([rows][cols], fp32 or bf16) whose bf16 value differs from ref[rows][cols]: the tied-head check
NTTS_KERNEL(256) void rows_mismatch_kernel(const void* src, int src_is_f32, const bf16_t* ref, long cols, unsigned int* count) {
    const long r = blockIdx.x;
    unsigned int bad = 0;
    for (long c = threadIdx.x; c < cols; c += 256) {
        const bf16_t v = src_is_f32 ? f2bf(((const float*)src)[r * cols + c]) : ((const bf16_t*)src)[r * cols + c];
        bad += v != ref[r * cols + c];
    }
    if (bad) atomic_add_global(count, bad);
}

}  // namespace ntts
