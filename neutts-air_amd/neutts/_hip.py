"""ctypes binding of libneutts_hip.so (include/neutts_hip.h) + thin host-side engine wrappers.

This is the only place the product touches native code.  There is NO CPU fallback: if the
library is missing it must be built (`python neutts-air_amd/build.py`), and if no gfx950 device is
present `ntts_backbone_create` fails with NTTS_ENODEV and we raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(os.path.dirname(_HERE), "libneutts_hip.so")

NTTS_DT_F32, NTTS_DT_BF16, NTTS_DT_I32, NTTS_DT_FP8_E4M3 = 0, 1, 2, 3
NTTS_W_BF16, NTTS_W_FP8_E4M3 = 0, 1
ABI_VERSION = 9
NTTS_PAGE_TOKENS = 32            # include/neutts_hip.h
PAGE_TOKENS = 32
ERRORS = {-1: "EINVAL", -2: "ENODEV", -3: "ENOMEM", -4: "ESTATE", -5: "EHIP"}


class NeuTTSHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libneutts_hip: {ERRORS.get(code, code)}: {msg}")
        self.code = code


class BackboneConfigC(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("intermediate_size", C.c_int32),
                ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32),
                ("head_dim", C.c_int32), ("rms_eps", C.c_float), ("max_context", C.c_int32),
                ("max_batch", C.c_int32), ("num_pages", C.c_int32), ("max_prefill_tokens", C.c_int32),
                ("tie_word_embeddings", C.c_int32), ("attention_bias", C.c_int32), ("qk_norm", C.c_int32),
                ("weight_dtype", C.c_int32), ("park_slots", C.c_int32)]


class CodecConfigC(C.Structure):
    _fields_ = [("hidden_size", C.c_int32), ("intermediate_size", C.c_int32), ("num_layers", C.c_int32),
                ("num_heads", C.c_int32), ("head_dim", C.c_int32), ("quantization_dim", C.c_int32),
                ("n_levels", C.c_int32), ("levels", C.c_int32 * 8), ("hop_length", C.c_int32), ("rms_eps", C.c_float),
                ("max_frames", C.c_int32), ("max_rows", C.c_int32), ("precision", C.c_int32)]


class EncoderConfigC(C.Structure):
    _fields_ = [("sem_hidden", C.c_int32), ("sem_layers", C.c_int32), ("sem_heads", C.c_int32), ("sem_ffn", C.c_int32),
                ("sem_conv_kernel", C.c_int32), ("sem_left", C.c_int32), ("sem_right", C.c_int32), ("sem_ln_eps", C.c_float),
                ("ac_hidden", C.c_int32), ("n_ratios", C.c_int32), ("ratios", C.c_int32 * 8), ("codec_hidden", C.c_int32),
                ("n_levels", C.c_int32), ("levels", C.c_int32 * 8), ("max_samples", C.c_int32)]


class StreamParamsC(C.Structure):
    _fields_ = [("chunk", C.c_int32), ("lookforward", C.c_int32), ("lookback", C.c_int32), ("overlap", C.c_int32),
                ("hop_length", C.c_int32), ("speech_base", C.c_int32), ("n_codes", C.c_int32), ("modulo", C.c_int32)]


class SamplingC(C.Structure):
    _fields_ = [("max_length", C.c_int32), ("min_new_tokens", C.c_int32), ("eos_token_id", C.c_int32),
                ("do_sample", C.c_int32), ("top_k", C.c_int32), ("temperature", C.c_float), ("seed", C.c_uint64)]


_LIBS: Dict[str, C.CDLL] = {}


def load_library(path: Optional[str] = None) -> C.CDLL:
    path = os.path.abspath(path or os.environ.get("NEUTTS_HIP_LIB", DEFAULT_LIB))
    if path in _LIBS:
        return _LIBS[path]
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build it with `python neutts-air_amd/build.py` (hipcc, gfx950). "
            "There is no CPU fallback for the NeuTTS hot path.")
    # Load order matters: PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  If this library
    # were dlopen'ed first it would pull /opt/rocm's copy in, torch would later load its bundled one, and the process
    # would hold TWO HIP runtimes (the second one sees no devices).  Importing torch first makes both share one.
    try:
        import torch  # noqa: F401  (plumbing only)
    except ImportError:
        pass
    lib = C.CDLL(path)
    p, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    sig = {
        "ntts_abi_version": (C.c_int, []),
        "ntts_last_error": (C.c_char_p, [p]),
        "ntts_backbone_create": (C.c_int, [C.POINTER(BackboneConfigC), C.c_int, C.POINTER(p)]),
        "ntts_backbone_destroy": (None, [p]),
        "ntts_backbone_load_tensor": (C.c_int, [p, C.c_char_p, p, C.c_int, C.POINTER(i64), C.c_int, C.c_int]),
        "ntts_backbone_finalize": (C.c_int, [p]),
        "ntts_backbone_arena": (C.c_int, [p, C.POINTER(p), C.POINTER(C.c_size_t)]),
        "ntts_backbone_adopt_arena": (C.c_int, [p]),
        "ntts_backbone_arena_derived": (C.c_int, [p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "ntts_backbone_arena_copy": (C.c_int, [p, p, C.c_size_t, C.c_int]),
        "ntts_backbone_share_arena": (C.c_int, [p, p]),
        "ntts_backbone_set_prefill_stream": (C.c_int, [p, p]),
        "ntts_backbone_time_kernel": (C.c_int, [p, i32, i32, C.POINTER(f32), C.POINTER(C.c_double), C.POINTER(i32)]),
        "ntts_backbone_prefill": (C.c_int, [p, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(SamplingC)]),
        "ntts_backbone_prefill_shared": (C.c_int, [p, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(SamplingC),
                                                   C.POINTER(i32), C.POINTER(i32)]),
        "ntts_backbone_set_gang": (C.c_int, [p, i32]),
        "ntts_backbone_activate": (C.c_int, [p, i32, C.POINTER(i32), C.POINTER(i32)]),
        "ntts_backbone_set_logits_range": (C.c_int, [p, i32, i32, i32]),
        "ntts_backbone_calibrate": (C.c_int, [p, i32]),
        "ntts_backbone_read_amax": (C.c_int, [p, C.POINTER(f32), i32]),
        "ntts_backbone_kv_stats": (C.c_int, [p, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64), C.POINTER(i64)]),
        "ntts_backbone_decode": (C.c_int, [p, i32]),
        "ntts_backbone_read": (C.c_int, [p, i32, C.POINTER(i32), i32, C.POINTER(i32), C.POINTER(i32)]),
        "ntts_backbone_read_all": (C.c_int, [p, C.POINTER(i32), i32, C.POINTER(i32), C.POINTER(i32)]),
        "ntts_backbone_poll": (C.c_int, [p, C.POINTER(i32), C.POINTER(i32)]),
        "ntts_backbone_poll_begin": (C.c_int, [p]),
        "ntts_backbone_poll_end": (C.c_int, [p, C.POINTER(i32), C.POINTER(i32)]),
        "ntts_backbone_read_finished": (C.c_int, [p, i32, C.POINTER(i32), i32, C.POINTER(i32)]),
        "ntts_backbone_release": (C.c_int, [p, i32]),
        "ntts_backbone_release_many": (C.c_int, [p, i32, C.POINTER(i32)]),
        "ntts_backbone_export_codes": (C.c_int, [p, i32, C.POINTER(i32), i32, i32, i32, p, i32, p]),
        "ntts_backbone_stream": (C.c_int, [p, C.POINTER(p)]),
        "ntts_backbone_set_stream": (C.c_int, [p, p]),
        "ntts_codec_set_stream": (C.c_int, [p, p]),
        "ntts_backbone_append_codes": (C.c_int, [p, i32, C.POINTER(i32), i32, i32, i32, p, i32, p, p, p]),
        "ntts_codec_stream": (C.c_int, [p, C.POINTER(p)]),
        "ntts_codec_set_debug": (C.c_int, [p, i32]),
        "ntts_codec_read_stage": (C.c_int, [p, i32, i32, C.POINTER(f32), i64, C.POINTER(i32), C.POINTER(i32)]),
        "ntts_codec_limits": (C.c_int, [p, C.POINTER(i32), C.POINTER(i64)]),
        "ntts_streams_last_error": (C.c_char_p, [p]),
        "ntts_streams_create": (C.c_int, [p, p, C.POINTER(StreamParamsC), i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32, C.POINTER(p)]),
        "ntts_streams_destroy": (None, [p]),
        "ntts_streams_pump_begin": (C.c_int, [p]),
        "ntts_streams_pump_wait": (C.c_int, [p, C.POINTER(i32)]),
        "ntts_streams_pump_end": (C.c_int, [p, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(C.POINTER(f32)),
                                            C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]),
        "ntts_streams_done": (C.c_int, [p, C.POINTER(i32)]),
        "ntts_codec_decode_dev": (C.c_int, [p, i32, p, i32, C.POINTER(i32), p, i64, i32, p]),
        "ntts_codec_sync": (C.c_int, [p]),
        "ntts_codec_set_cu_mask": (C.c_int, [p, C.POINTER(C.c_uint32), i32]),
        "ntts_backbone_set_prefill_cu_mask": (C.c_int, [p, C.POINTER(C.c_uint32), i32]),
        "ntts_backbone_sync": (C.c_int, [p]),
        "ntts_backbone_set_debug": (C.c_int, [p, i32]),
        "ntts_backbone_read_logits": (C.c_int, [p, i32, C.POINTER(f32), i32]),
        "ntts_backbone_debug_force": (C.c_int, [p, i32, i32]),
        "ntts_backbone_last_timing": (C.c_int, [p, C.POINTER(f32), C.POINTER(f32)]),
        "ntts_backbone_step_bytes": (C.c_int, [p, C.POINTER(C.c_double)]),
        "ntts_backbone_attn_timeline": (C.c_int, [p, i32, C.POINTER(C.c_uint64), i64]),
        "ntts_backbone_gemv_timeline": (C.c_int, [p, i32, i32, C.POINTER(C.c_uint64), i64, C.POINTER(i32)]),
        "ntts_codec_last_error": (C.c_char_p, [p]),
        "ntts_codec_create": (C.c_int, [C.POINTER(CodecConfigC), C.c_int, C.POINTER(p)]),
        "ntts_codec_destroy": (None, [p]),
        "ntts_codec_load_tensor": (C.c_int, [p, C.c_char_p, p, C.c_int, C.POINTER(i64), C.c_int, C.c_int]),
        "ntts_codec_finalize": (C.c_int, [p]),
        "ntts_codec_decode": (C.c_int, [p, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(f32), i64]),
        "ntts_codec_last_timing": (C.c_int, [p, C.POINTER(f32)]),
        "ntts_encoder_last_error": (C.c_char_p, [p]),
        "ntts_encoder_create": (C.c_int, [C.POINTER(EncoderConfigC), C.c_int, C.POINTER(p)]),
        "ntts_encoder_destroy": (None, [p]),
        "ntts_encoder_load_tensor": (C.c_int, [p, C.c_char_p, p, C.c_int, C.POINTER(i64), C.c_int, C.c_int]),
        "ntts_encoder_finalize": (C.c_int, [p]),
        "ntts_encoder_encode": (C.c_int, [p, C.POINTER(f32), i64, C.POINTER(i32), i32, C.POINTER(i32)]),
        "ntts_encoder_read_stage": (C.c_int, [p, i32, C.POINTER(f32), i64, C.POINTER(i32), C.POINTER(i32)]),
        "ntts_encoder_last_timing": (C.c_int, [p, C.POINTER(f32)]),
        "ntts_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(p)]),
        "ntts_host_free": (C.c_int, [p]),
        "ntts_stream_create": (C.c_int, [i32, C.POINTER(p)]),
        "ntts_stream_destroy": (C.c_int, [i32, p]),
        "ntts_k_gemm_bf16": (C.c_int, [p, i64, p, p, p, i64, i32, i32, i32, i32]),
        "ntts_k_gemm_probe": (C.c_int, [i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_double)]),
        "ntts_k_rmsnorm_bf16": (C.c_int, [p, p, p, i32, i32, f32]),
        "ntts_k_fp8_quantize": (C.c_int, [p, p, i64, f32]),
        "ntts_k_gemm_fp8": (C.c_int, [p, p, p, f32, p, p, i32, i32, i32, i32]),
        "ntts_k_membw": (C.c_int, [C.c_size_t, i32, C.POINTER(C.c_double)]),
        "ntts_k_mfma_probe": (C.c_int, [p]),
        "ntts_k_silu_probe": (C.c_int, [p, p, i64, i32]),
        "ntts_k_launch_chain_probe": (C.c_int, [i32, i32, i32, i32, C.POINTER(C.c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here == header/library drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    _LIBS[path] = lib
    return lib


def _tensor_ptr(t):
    """(pointer, dtype code, shape, is_device, keepalive) for numpy fp32 / torch fp32|bf16 (cpu|cuda)."""
    if isinstance(t, np.ndarray):
        a = np.ascontiguousarray(t, dtype=np.float32)
        return a.ctypes.data, NTTS_DT_F32, a.shape, 0, a
    import torch  # plumbing only: tensor containers
    if isinstance(t, torch.Tensor):
        tt = t.detach().contiguous()
        if tt.dtype == getattr(torch, "float8_e4m3fn", None):      # a pre-quantised matrix of an fp8 checkpoint: its bytes
            return tt.data_ptr(), NTTS_DT_FP8_E4M3, tuple(tt.shape), int(tt.is_cuda), tt
        if tt.dtype == torch.bfloat16:
            code = NTTS_DT_BF16
        else:
            tt = tt.to(torch.float32)
            code = NTTS_DT_F32
        return tt.data_ptr(), code, tuple(tt.shape), int(tt.is_cuda), tt
    raise TypeError(f"unsupported tensor type {type(t)}")


@dataclass
class Sampling:
    """Keyword arguments of the reference's generate() call (ref:neutts/neutts.py:338-347)."""
    max_length: int = 2048
    min_new_tokens: int = 50
    eos_token_id: int = 0
    do_sample: bool = True
    top_k: int = 50
    temperature: float = 1.0
    seed: int = 0

    def to_c(self) -> SamplingC:
        return SamplingC(self.max_length, self.min_new_tokens, self.eos_token_id, int(self.do_sample), self.top_k,
                         self.temperature, self.seed)


class BackboneEngine:
    """One engine per GPU (per process): weights + paged KV + `max_batch` decode slots."""

    def __init__(self, cfg: dict, device: int = 0, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        if self.lib.ntts_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libneutts_hip ABI mismatch: library {self.lib.ntts_abi_version()}, binding {ABI_VERSION} (rebuild: python neutts-air_amd/build.py)")
        self.cfg = dict(cfg)
        self._device, self._lib_path = device, lib_path
        c = BackboneConfigC(cfg["vocab_size"], cfg["hidden_size"], cfg["intermediate_size"], cfg["num_layers"],
                            cfg["num_heads"], cfg["num_kv_heads"], cfg.get("head_dim", 64), cfg.get("rms_eps", 1e-6),
                            cfg.get("max_context", 2048), cfg.get("max_batch", 1) + int(cfg.get("park_slots", 0)), cfg.get("num_pages", 0),
                            cfg.get("max_prefill_tokens", 0), int(cfg.get("tie_word_embeddings", True)),
                            int(cfg.get("attention_bias", True)), int(cfg.get("qk_norm", False)),
                            {"bf16": NTTS_W_BF16, "fp8": NTTS_W_FP8_E4M3, "fp8_e4m3": NTTS_W_FP8_E4M3}[str(cfg.get("weight_dtype", "bf16"))],
                            int(cfg.get("park_slots", 0)))
        self.fp8 = c.weight_dtype == NTTS_W_FP8_E4M3
        h = C.c_void_p()
        rc = self.lib.ntts_backbone_create(C.byref(c), device, C.byref(h))
        if rc != 0:
            raise NeuTTSHipError(rc, (self.lib.ntts_last_error(None) or b"").decode())
        self.h = h
        # cfg["park_slots"] = P extra PARKING rows behind the cfg["max_batch"] decode slots (ABI 9 ntts_backbone_config.park_slots: the library counts
        # them into its max_batch): rows max_batch .. max_batch + P - 1 take prompt passes but never decode; activate() moves one into a decode slot
        self.park_slots = c.park_slots
        self.n_rows = c.max_batch                         # every slot, parking rows included (the library's max_batch)
        self.max_batch = c.max_batch - c.park_slots       # decode slots (rows of the decode step)
        self._free_park: List[int] = list(range(self.n_rows - 1, self.max_batch - 1, -1))
        self.max_context = c.max_context
        self.vocab_size = c.vocab_size
        self._free: List[int] = list(range(self.max_batch - 1, -1, -1))   # host-side pool of decode slots (pop -> slot 0 first)
        self.counters = {"decode_steps": 0, "prefill_calls": 0, "prefill_prompts": 0, "prefill_tokens": 0}   # diagnostics (bench.py)

    # -- decode-slot pool: every path that admits a request (generate, the streaming generators) draws from here, so an
    #    unfinished stream and a later call can never be handed the same slot
    def acquire_slot(self) -> int:
        if not self._free:
            raise NeuTTSHipError(-4, f"all {self.max_batch} decode slots are in use")
        return self._free.pop()

    def free_slots(self) -> int:
        return len(self._free)

    # -- plumbing
    def _chk(self, rc: int):
        if rc != 0:
            raise NeuTTSHipError(rc, (self.lib.ntts_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.ntts_backbone_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights
    def load_tensor(self, name: str, t):
        ptr, code, shape, is_dev, keep = _tensor_ptr(t)
        shp = (C.c_int64 * len(shape))(*shape)
        self._chk(self.lib.ntts_backbone_load_tensor(self.h, name.encode(), C.c_void_p(ptr), code, shp, len(shape), is_dev))
        del keep

    def load_state_dict(self, sd: Dict[str, object], inv_freq=None, input_scales: Optional[Dict[str, float]] = None):
        """HF Qwen2ForCausalLM state dict (+ rope.inv_freq; computed like hf:...modeling_qwen2.py:86 if absent).
        fp8 engines (weight_dtype="fp8"): bf16 / fp32 matrices are quantised on upload; matrices that are ALREADY torch.float8_e4m3fn
        (a pre-quantised checkpoint) are stored as they are and need their `<module>.weight_scale` entry; the static activation scales
        come as `*.input_scale` entries of `sd` (static-fp8 checkpoints) or as the `input_scales` dict {tensor name: float}."""
        for k, v in sd.items():
            if k.endswith("rotary_emb.inv_freq"):
                continue
            if k.endswith(".input_scale"):
                v = np.asarray(v.float().cpu() if hasattr(v, "cpu") else v, dtype=np.float32).reshape(1)
            elif k.endswith(".weight_scale"):       # pre-quantised fp8 checkpoint: per-output-channel (or per-matrix) scales, fp32 on the way in
                v = np.ascontiguousarray(np.asarray(v.float().cpu() if hasattr(v, "cpu") else v, dtype=np.float32))
            self.load_tensor(k, v)
        for k, v in (input_scales or {}).items():
            self.load_tensor(k, np.asarray([v], dtype=np.float32))
        if inv_freq is None:
            raise ValueError("inv_freq (fp32 [head_dim/2]) must be supplied: it is a model buffer, not a constant")
        self.load_tensor("rope.inv_freq", np.asarray(inv_freq, dtype=np.float32))
        self._chk(self.lib.ntts_backbone_finalize(self.h))

    def arena(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.ntts_backbone_arena(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def arena_derived(self):
        """(offset, bytes) of the arena range a broadcast may skip (rebuilt by adopt_arena on the receiver)."""
        o, n = C.c_size_t(), C.c_size_t()
        self._chk(self.lib.ntts_backbone_arena_derived(self.h, C.byref(o), C.byref(n)))
        return o.value, n.value

    def arena_copy(self, buf_ptr: int, nbytes: int, to_arena: bool):
        self._chk(self.lib.ntts_backbone_arena_copy(self.h, C.c_void_p(buf_ptr), nbytes, int(to_arena)))

    KERNELS = ["attn_decode_kernel", "gemm_qkv", "gemm_o_proj_splitk", "gemm_gate_up_silu", "gemm_down_splitk",
               "gemm_lm_head_argmax", "add_rmsnorm_kernel"]

    def time_kernel(self, which: int, iters: int = 20):
        ms, nb, nl = C.c_float(), C.c_double(), C.c_int32()
        self._chk(self.lib.ntts_backbone_time_kernel(self.h, which, iters, C.byref(ms), C.byref(nb), C.byref(nl)))
        return ms.value, nb.value, nl.value

    def adopt_arena(self):
        self._chk(self.lib.ntts_backbone_adopt_arena(self.h))

    def twin(self, share: bool = True) -> "BackboneEngine":
        """A second engine with this one's configuration and weights -- its own KV pool, slot state, stream and step graph.  What
        running several batches at once needs: the twins' step graphs replayed alternately (each chain fills the other's launch gaps),
        one twin's prompt pass beside another's decode steps (bench.py static mode, EngineGang).  share=True: the twin READS THIS
        ENGINE'S ARENA (ntts_backbone_share_arena: no second copy of the weights; this engine is kept alive by the twin);
        share=False: its own arena, filled by one device-to-device copy."""
        self.sync()
        t = BackboneEngine(self.cfg, self._device, self._lib_path)
        if share:
            t._chk(self.lib.ntts_backbone_share_arena(t.h, self.h))
            t._donor = self
        else:
            ptr, nbytes = self.arena()
            t.arena_copy(ptr, nbytes, True)
            t.adopt_arena()
        if getattr(self, "logits_range", None):
            t.set_logits_range(*self.logits_range)          # (an opt-in restricted lm_head goes along: its own compacted copy)
        return t

    # -- requests
    def prefill(self, prompts: Sequence[Sequence[int]], slots: Sequence[int], sampling: Sequence[Sampling],
                donors: Optional[Sequence[Optional[tuple]]] = None):
        """donors (optional): per prompt None or (donor_slot, shared_len) -- re-use the KV pages of the first
        `shared_len` tokens (rounded down to whole pages) of a running slot / an earlier prompt of this call whose
        prompt starts with the same tokens (include/neutts_hip.h: ntts_backbone_prefill_shared)."""
        n = len(prompts)
        lens = np.array([len(p) for p in prompts], dtype=np.int32)
        ids = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.int32) for p in prompts]))
        sl = np.asarray(slots, dtype=np.int32)
        try:
            self._prefill_call(n, ids, lens, sl, sampling, donors)
        except NeuTTSHipError:
            raise                      # the engine admitted nothing: slots drawn from the pool stay with the caller to release
        self._mark_busy(slots)
        self.counters["prefill_calls"] += 1
        self.counters["prefill_prompts"] += n
        self.counters["prefill_tokens"] += int(lens.sum())

    def _prefill_call(self, n, ids, lens, sl, sampling, donors):
        sc = (SamplingC * n)(*[s.to_c() for s in sampling])
        i32p = C.POINTER(C.c_int32)
        if donors is None or all(d is None for d in donors):
            self._chk(self.lib.ntts_backbone_prefill(self.h, n, ids.ctypes.data_as(i32p), lens.ctypes.data_as(i32p),
                                                     sl.ctypes.data_as(i32p), sc))
            return
        ds = np.array([-1 if d is None else d[0] for d in donors], dtype=np.int32)
        dl = np.array([0 if d is None else d[1] for d in donors], dtype=np.int32)
        self._chk(self.lib.ntts_backbone_prefill_shared(self.h, n, ids.ctypes.data_as(i32p), lens.ctypes.data_as(i32p),
                                                        sl.ctypes.data_as(i32p), sc, ds.ctypes.data_as(i32p),
                                                        dl.ctypes.data_as(i32p)))

    def kv_stats(self) -> dict:
        fp, tp, tc, ts = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        self._chk(self.lib.ntts_backbone_kv_stats(self.h, C.byref(fp), C.byref(tp), C.byref(tc), C.byref(ts)))
        return {"free_pages": fp.value, "total_pages": tp.value, "prompt_tokens_computed": tc.value,
                "prompt_tokens_shared": ts.value}

    def decode(self, n_steps: int = 1):
        self._chk(self.lib.ntts_backbone_decode(self.h, n_steps))
        self.counters["decode_steps"] += n_steps

    def read(self, slot: int):
        out = np.empty(self.max_context, dtype=np.int32)
        n, fin = C.c_int32(), C.c_int32()
        self._chk(self.lib.ntts_backbone_read(self.h, slot, out.ctypes.data_as(C.POINTER(C.c_int32)), len(out),
                                              C.byref(n), C.byref(fin)))
        return out[: n.value].tolist(), bool(fin.value)

    def poll_begin(self):
        """Asynchronous poll: snapshot every slot's state / new-token count behind the work enqueued so far (no host wait)."""
        self._chk(self.lib.ntts_backbone_poll_begin(self.h))

    def poll_end(self):
        """(state, n_new) of the snapshot opened by poll_begin; waits for that copy only, not for work enqueued after it."""
        st = np.empty(self.n_rows, dtype=np.int32)
        nn = np.empty(self.n_rows, dtype=np.int32)
        i32p = C.POINTER(C.c_int32)
        self._chk(self.lib.ntts_backbone_poll_end(self.h, st.ctypes.data_as(i32p), nn.ctypes.data_as(i32p)))
        return st, nn

    def read_finished(self, slot: int) -> List[int]:
        """The ids of a slot the last completed snapshot showed finished -- copied past the decode steps still queued."""
        out = np.empty(self.max_context, dtype=np.int32)
        n = C.c_int32()
        self._chk(self.lib.ntts_backbone_read_finished(self.h, slot, out.ctypes.data_as(C.POINTER(C.c_int32)), len(out), C.byref(n)))
        return out[: n.value].tolist()

    def read_all_array(self):
        """Every slot's new ids in one call, as arrays: (ids [max_batch, max_context] int32 -- row s valid up to n[s]),
        n [max_batch], finished [max_batch] bool).  The batch hand-off to the codec stays in numpy (no Python lists)."""
        out = np.empty((self.n_rows, self.max_context), dtype=np.int32)
        n = np.empty(self.n_rows, dtype=np.int32)
        fin = np.empty(self.n_rows, dtype=np.int32)
        i32p = C.POINTER(C.c_int32)
        self._chk(self.lib.ntts_backbone_read_all(self.h, out.ctypes.data_as(i32p), self.max_context,
                                                  n.ctypes.data_as(i32p), fin.ctypes.data_as(i32p)))
        return out, n, fin.astype(bool)

    def read_all(self):
        """Every slot's new ids in one call: ([ids per slot], [finished per slot])."""
        out, n, fin = self.read_all_array()
        return [out[s, : n[s]].tolist() for s in range(self.max_batch)], [bool(f) for f in fin]

    def poll(self):
        st = np.empty(self.n_rows, dtype=np.int32)
        nn = np.empty(self.n_rows, dtype=np.int32)
        i32p = C.POINTER(C.c_int32)
        self._chk(self.lib.ntts_backbone_poll(self.h, st.ctypes.data_as(i32p), nn.ctypes.data_as(i32p)))
        return st, nn

    def warm_up(self, decode_steps: int = 2):
        """One throw-away request through slot 0 (a 1-token prompt, `decode_steps` greedy steps): captures and instantiates the
        decode-step hipGraph and lets the HIP runtime size its command pools, so that the first real request does not pay for
        either.  Serving start-up hygiene; the engine must be finalised and slot 0 free."""
        eos = self.logits_range[2] if getattr(self, "logits_range", None) else 0      # (a restricted lm_head accepts its own EOS id only)
        sp = Sampling(max_length=2 + decode_steps, min_new_tokens=1 + decode_steps, eos_token_id=eos, do_sample=False)
        self.prefill([[1 % self.vocab_size]], [0], [sp])
        self.decode(decode_steps)
        self.sync()
        self.release(0)
        self.sync()
        # The runtime sizes its pools by the number of graph replays that were ever queued before a synchronisation: the first
        # call AFTER the first long decode call pays for it once (43 ms on the host, measured in bench.py's step walls).  With
        # `decode_steps` as long as a real decode call, a second throw-away prompt pass absorbs that too.
        if decode_steps > 8:
            self.prefill([[1 % self.vocab_size]], [0], [Sampling(max_length=3, min_new_tokens=1, eos_token_id=eos, do_sample=False)])
            self.sync()
            self.release(0)
            self.sync()

    def set_prefill_cu_mask(self, mask_words: Optional[Sequence[int]]):
        """Run this engine's prompt passes on a side stream restricted to the CUs of `mask_words` (32-bit words, bit i of word w =
        CU 32 w + i); None / empty = default.  See include/neutts_hip.h: ntts_backbone_set_prefill_cu_mask."""
        words = list(mask_words or [])
        arr = (C.c_uint32 * max(1, len(words)))(*words)
        self._chk(self.lib.ntts_backbone_set_prefill_cu_mask(self.h, arr, len(words)))

    def stream(self) -> int:
        st = C.c_void_p()
        self._chk(self.lib.ntts_backbone_stream(self.h, C.byref(st)))
        return st.value or 0

    def export_codes(self, slots: Sequence[int], speech_base: int, n_codes: int, codes_dev_ptr: int, stride: int,
                     lens_dev_ptr: int, modulo: bool = False):
        """Device-side id -> code hand-off (ntts_backbone_export_codes): codes land in a caller-owned DEVICE int32 buffer
        [len(slots), stride], their counts in a DEVICE int32 buffer [len(slots)]; asynchronous on the engine's stream."""
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        self._chk(self.lib.ntts_backbone_export_codes(self.h, len(sl), sl.ctypes.data_as(C.POINTER(C.c_int32)), speech_base,
                                                      n_codes, int(modulo), C.c_void_p(codes_dev_ptr), stride,
                                                      C.c_void_p(lens_dev_ptr)))

    def release(self, slot: int):
        self._chk(self.lib.ntts_backbone_release(self.h, slot))
        pool = self._free if slot < self.max_batch else self._free_park
        if slot not in pool:
            pool.append(slot)

    def acquire_park(self) -> int:
        if not self._free_park:
            raise NeuTTSHipError(-4, f"all {self.park_slots} parking rows are in use")
        return self._free_park.pop()

    def activate(self, park_rows: Sequence[int], slots: Sequence[int]):
        """Parked requests -> free decode slots (ntts_backbone_activate): stream-ordered behind the prompt pass that filled the parking rows and
        ahead of the next decode step.  The caller holds the slots (acquire_slot) and gets the parking rows back."""
        a = np.ascontiguousarray(park_rows, dtype=np.int32)
        b = np.ascontiguousarray(slots, dtype=np.int32)
        self._chk(self.lib.ntts_backbone_activate(self.h, len(a), a.ctypes.data_as(C.POINTER(C.c_int32)), b.ctypes.data_as(C.POINTER(C.c_int32))))
        for r in a.tolist():
            if r not in self._free_park:
                self._free_park.append(r)

    def release_many(self, slots: Sequence[int]):
        """release() for a whole set of slots with one stream operation."""
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        self._chk(self.lib.ntts_backbone_release_many(self.h, len(sl), sl.ctypes.data_as(C.POINTER(C.c_int32))))
        for s in sl.tolist():
            pool = self._free if s < self.max_batch else self._free_park
            if s not in pool:
                pool.append(s)

    def _mark_busy(self, slots: Sequence[int]):
        """Callers that choose slot numbers themselves (tests, bench): keep the pool consistent."""
        for s in slots:
            if s in self._free:
                self._free.remove(s)

    def sync(self):
        self._chk(self.lib.ntts_backbone_sync(self.h))

    def set_gang(self, chains: int):
        """PIN how many decode chains (engines of an EngineGang, itself included) the engine assumes side by side on the GPU: the decode
        step's GEMM tiles and XCD placement are chosen for that (ntts_backbone_set_gang).  0 = count them (the default since ABI 9: engines
        of the same arena that decoded within the last 50 ms); both shapes keep their captured step graph.  Tests, sweeps, profiling."""
        self._chk(self.lib.ntts_backbone_set_gang(self.h, int(chains)))

    def calibrate(self, enable: bool = True):
        """fp8 calibration mode of a BF16 engine (ntts_backbone_calibrate): the prompt passes that follow record max |x| of every GEMM input."""
        self._chk(self.lib.ntts_backbone_calibrate(self.h, int(bool(enable))))

    def fp8_input_scales(self, margin: float = 1.0) -> Dict[str, float]:
        """The static `input_scale`s of the fp8 model from the record so far: amax / 448 x margin, under the tensor names a static-fp8
        checkpoint uses (what load_state_dict(..., input_scales=...) of a weight_dtype='fp8' engine takes)."""
        L = int(self.cfg["num_layers"])
        a = np.zeros(4 * L + 1, dtype=np.float32)
        self._chk(self.lib.ntts_backbone_read_amax(self.h, a.ctypes.data_as(C.POINTER(C.c_float)), len(a)))
        if not (a > 0).all():
            raise NeuTTSHipError(-4, "calibration saw no data for some GEMM inputs: run prompt passes between calibrate() and fp8_input_scales()")
        out = {"lm_head.input_scale": float(a[4 * L]) / 448.0 * margin}
        for i in range(L):
            for j, t in enumerate(("self_attn.q_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.down_proj")):
                out[f"model.layers.{i}.{t}.input_scale"] = float(a[4 * i + j]) / 448.0 * margin
        return out

    def set_logits_range(self, lo: Optional[int], hi: int = 0, eos_id: int = 0):
        """OPT-IN: lm_head over the token ids [lo, hi) + eos_id only (ntts_backbone_set_logits_range; lo=None restores the full head).
        The reference takes its argmax / top-k over the whole vocabulary: identical ids only while its choice lies in the range."""
        if lo is None:
            self._chk(self.lib.ntts_backbone_set_logits_range(self.h, -1, 0, 0))
        else:
            self._chk(self.lib.ntts_backbone_set_logits_range(self.h, int(lo), int(hi), int(eos_id)))
        self.logits_range = None if lo is None else (int(lo), int(hi), int(eos_id))

    def set_stream(self, stream: Optional[int]):
        """Run the engine's work on the caller's HIP stream (a hipStream_t as an integer, e.g. torch.cuda.Stream().cuda_stream); None = its own."""
        self._chk(self.lib.ntts_backbone_set_stream(self.h, C.c_void_p(stream or None)))

    def set_prefill_stream(self, stream: Optional[int]):
        """Prompt passes on the caller's HIP stream, event-ordered with the engine's stream (ntts_backbone_set_prefill_stream);
        None = back on the engine's stream."""
        self._chk(self.lib.ntts_backbone_set_prefill_stream(self.h, C.c_void_p(stream or None)))

    def set_debug(self, keep_logits: bool):
        self._chk(self.lib.ntts_backbone_set_debug(self.h, int(keep_logits)))

    def read_logits(self, slot: int) -> np.ndarray:
        out = np.empty(self.vocab_size, dtype=np.float32)
        self._chk(self.lib.ntts_backbone_read_logits(self.h, slot, out.ctypes.data_as(C.POINTER(C.c_float)), len(out)))
        return out

    def debug_force(self, slot: int, token: int):
        self._chk(self.lib.ntts_backbone_debug_force(self.h, slot, token))

    def last_timing(self):
        a, b = C.c_float(), C.c_float()
        self._chk(self.lib.ntts_backbone_last_timing(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def gemv_timeline(self, which: int, layer: int = 0) -> np.ndarray:
        """[workgroups, 16] phase timestamps (100 MHz ticks) of one small-batch GEMV launch: which = 2 o_proj, 3 gate/up, 4 down_proj."""
        out = np.zeros((4096, 16), dtype=np.uint64)
        n = C.c_int32()
        self._chk(self.lib.ntts_backbone_gemv_timeline(self.h, which, layer, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size, C.byref(n)))
        return out[: n.value]

    def attn_timeline(self, layer: int = 0) -> np.ndarray:
        """Phase timestamps of one decode-attention launch: [max_batch, kv_heads, 4 waves, 8 phases], 100 MHz ticks."""
        nkv = self.cfg["num_kv_heads"]
        out = np.zeros((self.max_batch, nkv, 4, 8), dtype=np.uint64)
        self._chk(self.lib.ntts_backbone_attn_timeline(self.h, layer, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size))
        return out

    def step_bytes(self) -> float:
        d = C.c_double()
        self._chk(self.lib.ntts_backbone_step_bytes(self.h, C.byref(d)))
        return d.value

    # -- continuous batching (host scheduler): keep every slot busy until all prompts are done
    def generate(self, prompts: Sequence[Sequence[int]], sampling, steps_per_poll: int = 16,
                 prefill_token_budget: Optional[int] = None, share_prefix: bool = False, min_admit: int = 1,
                 on_finished=None, run_ahead: bool = True, admit_gate=None, on_admit=None) -> List[List[int]]:
        """Batched equivalent of calling ref:neutts/neutts.py:338-351 once per prompt.
        Returns the NEW ids of each prompt (prompt stripped), in order.
        share_prefix=True: a prompt that starts like one already in flight (same speaker: chat header + reference
        text, ref:neutts/neutts.py:307,315-325) re-uses that slot's KV pages for the common whole pages.
        min_admit: waiting prompts are admitted only once that many slots are free (or nothing is running): a prompt pass over
        a handful of prompts runs the big GEMM tiles nearly empty, so under load it pays to let a few slots idle for some steps
        and prefill them together.
        run_ahead (default): the scheduler keeps ONE burst of `steps_per_poll` decode steps queued ahead of its own bookkeeping --
        it enqueues burst k + 1 and only then reads the slot states as they were after burst k (ntts_backbone_poll_begin / _end),
        so the GPU never idles while the host reads finished rows, releases slots and packs the next prompt pass.  The price is
        that a finished row is noticed up to one burst later (it is carried along masked meanwhile: stopping is decided on the
        device, the ids cannot differ).
        on_finished(request_index, slot, n_new): called for every finished request BEFORE its slot is released, INSTEAD of
        copying its ids to the host (that request's entry of the result is then []): the hook of a device-side hand-off
        (ntts_backbone_export_codes enqueued from it is ordered before the slot's re-use).
        admit_gate(n_free, n_wanted) -> bool replaces the `min_admit` test while requests are running (n_wanted = min(min_admit,
        requests waiting)); on_admit(n_prompts) is called after every prompt pass: the two hooks by which an EngineGang coordinates
        its engines' admissions (a prompt pass takes an engine's lane for ~ 0.8 us per prompt token; with nothing running the gate
        is not asked)."""
        it = self.generate_iter(prompts, sampling, steps_per_poll, prefill_token_budget, share_prefix, min_admit, on_finished, run_ahead,
                                admit_gate, on_admit)
        while True:
            try:
                next(it)
            except StopIteration as done:
                return done.value

    def generate_iter(self, prompts: Sequence[Sequence[int]], sampling, steps_per_poll: int = 16,
                      prefill_token_budget: Optional[int] = None, share_prefix: bool = False, min_admit: int = 1,
                      on_finished=None, run_ahead: bool = True, admit_gate=None, on_admit=None):
        """generate() as a generator: yields (None) once per scheduler iteration -- after this engine's next burst and snapshot
        are enqueued -- and returns generate()'s result as the StopIteration value.  What EngineGang alternates between: while one
        engine's scheduler waits for its previous snapshot, the other engines' bursts are already queued on their own streams."""
        if isinstance(sampling, Sampling):
            sampling = [sampling] * len(prompts)
        budget = prefill_token_budget or self.cfg.get("max_prefill_tokens", 0) or 16384
        # validate up front: a request that cannot run must not strand the ones admitted before it
        for i, (p, sp) in enumerate(zip(prompts, sampling)):
            if len(p) < 1:
                raise ValueError(f"prompt {i} is empty")
            if not (len(p) < sp.max_length <= self.max_context):
                raise ValueError(f"prompt {i}: need len(prompt) < max_length <= max_context "
                                 f"({len(p)}, {sp.max_length}, {self.max_context})")
            if len(p) > budget:
                raise ValueError(f"prompt {i}: {len(p)} tokens exceed max_prefill_tokens {budget}")
        # KV admission control: a request is admitted only if the pool can hold it up to ITS max_length next to everything
        # already running (pages are allocated as the sequence grows; without the reservation two admitted requests could
        # starve each other in the middle of decoding)
        # Pages held OUTSIDE this call (a suspended infer_stream generator, slots another caller prefilled) are not this scheduler's
        # to hand out: the reservation budget is what was free when the call started (comparing with the pool's size let
        # the scheduler admit requests the pool could never hold, and the run-ahead loop then span on a burst that could not succeed).
        st0 = self.kv_stats()
        pool_pages = st0["total_pages"]
        total_pages = st0["free_pages"]
        need_pages = [(sp.max_length - 1 + NTTS_PAGE_TOKENS - 1) // NTTS_PAGE_TOKENS for sp in sampling]
        for i, n in enumerate(need_pages):
            if n > total_pages:
                raise NeuTTSHipError(-3, f"prompt {i}: max_length {sampling[i].max_length} needs {n} KV pages, the pool has {total_pages} free of {pool_pages}")
        committed: Dict[int, int] = {}                      # slot -> pages reserved for it
        results: List[Optional[List[int]]] = [None] * len(prompts)
        owner: Dict[int, int] = {}
        anchors: List[tuple] = []       # (slot, prompt as int32 array) of live slots that later prompts are compared with
        arrs = [np.asarray(p, dtype=np.int32) for p in prompts] if share_prefix else None
        # run-ahead bookkeeping: snapshots are numbered in stream order; a slot's entry in snapshot q describes the request that
        # owns it now only if the slot was (re)filled before snapshot q was enqueued
        snap_seq = 0                    # number of the next snapshot to be enqueued
        valid_from: Dict[int, int] = {}  # slot -> first snapshot number that describes its current request
        steps_left: Dict[int, int] = {}  # slot -> decode steps its request can still take before max_length stops it (run-ahead: upper bound)
        open_snap: Optional[int] = None  # number of the snapshot opened by poll_begin and not read yet

        def find_donor(i):
            best = None
            for slot, arr in anchors:
                m = min(len(arr), len(arrs[i]) - 1)
                if m < NTTS_PAGE_TOKENS:
                    continue
                neq = np.flatnonzero(arr[:m] != arrs[i][:m])
                lcp = int(neq[0]) if len(neq) else m
                if lcp >= NTTS_PAGE_TOKENS and (best is None or lcp > best[1]):
                    best = (slot, lcp)
            return best

        nxt = 0

        # Parking (cfg["park_slots"] > 0): a prompt pass may also fill PARKING rows -- the requests wait there, prefilled, and move into a decode slot
        # the moment one is released.  Admission then counts the free parking rows as well: the waves of `min_admit` prompts keep their
        # efficient size while no decode row idles waiting for the next wave (slot occupancy 0.92 -> ~0.99: bench.py --mode continuous).
        from collections import deque
        parked: "deque[int]" = deque()   # parking rows holding a prefilled request, oldest first

        def n_free():
            return len(self._free) + len(self._free_park)

        def may_admit():
            if not owner:
                return True
            want = min(min_admit, len(prompts) - nxt)
            return admit_gate(n_free(), want) if admit_gate is not None else n_free() >= want

        def activate_parked():
            """parked requests into the decode slots that are free right now; their bookkeeping moves along"""
            nonlocal anchors
            rows, slots_ = [], []
            while parked and self._free:
                rows.append(parked.popleft())
                slots_.append(self.acquire_slot())
            if not rows:
                return
            self.activate(rows, slots_)
            for r, s in zip(rows, slots_):
                owner[s] = owner.pop(r)
                committed[s] = committed.pop(r)
                steps_left[s] = steps_left.pop(r)
                valid_from.pop(r, None)
                valid_from[s] = snap_seq                  # the snapshots enqueued from here on describe it in its decode slot
                anchors = [(s if a[0] == r else a[0], a[1]) for a in anchors]

        try:
            while nxt < len(prompts) or owner:
                # admit as many waiting prompts as slots / prefill workspace allow
                while nxt < len(prompts) and n_free() and may_admit():
                    batch, used, donors = [], 0, []
                    while nxt < len(prompts) and n_free():
                        d = find_donor(nxt) if share_prefix else None
                        cost = len(prompts[nxt]) - (d[1] // NTTS_PAGE_TOKENS * NTTS_PAGE_TOKENS if d else 0)
                        if used + cost > budget or sum(committed.values()) + need_pages[nxt] > total_pages:
                            break
                        s = self.acquire_slot() if self._free else self.acquire_park()    # decode slots first, then parking rows
                        if s >= self.max_batch:
                            parked.append(s)
                        committed[s] = need_pages[nxt]
                        batch.append((nxt, s))
                        donors.append(d)
                        used += cost
                        owner[s] = nxt
                        valid_from[s] = snap_seq
                        steps_left[s] = sampling[nxt].max_length - len(prompts[nxt]) - 1
                        if share_prefix and d is None and len(anchors) < 16:
                            anchors.append((s, arrs[nxt]))      # a new beginning: later prompts may share it
                        nxt += 1
                    if not batch:
                        break
                    try:
                        self.prefill([prompts[i] for i, _ in batch], [s for _, s in batch], [sampling[i] for i, _ in batch],
                                     donors if share_prefix else None)
                    except NeuTTSHipError as ex:
                        running = [s for s in owner if s not in [b[1] for b in batch]]
                        if ex.code != -3 or not running:
                            raise
                        # KV page pool exhausted while other requests still run: hand this batch back, let the running
                        # ones finish and free their pages, then try again
                        for i, s in reversed(batch):
                            owner.pop(s)
                            steps_left.pop(s, None)
                            committed.pop(s, None)
                            anchors = [a for a in anchors if a[0] != s]
                            if s >= self.max_batch:
                                parked.remove(s)
                                self._free_park.append(s)
                            else:
                                self._free.append(s)
                        nxt = batch[0][0]
                        break
                    if on_admit is not None:
                        on_admit(len(batch))
                if not owner:
                    raise NeuTTSHipError(-4, "no decode slot is free (held by an unfinished stream?)")
                activate_parked()                                 # (decode slots freed in the previous iteration)
                if run_ahead:
                    # keep the GPU fed: this burst goes in BEFORE the host looks at the previous burst's outcome.  Stream order per
                    # iteration j: [prompt pass j] [burst j] [exports / releases j] [snapshot j]; the host waits for snapshot j - 1
                    # only, with burst j queued behind it.  No burst when every owner has certainly stopped already (each has been
                    # given the decode steps its max_length allows: the snapshots in flight will show them finished).
                    burst_failed = None
                    if any(steps_left.get(s, 1) > 0 for s in owner if s < self.max_batch):
                        try:
                            self.decode(steps_per_poll)
                            for s in owner:
                                if s < self.max_batch:            # (parked requests take no decode steps)
                                    steps_left[s] = steps_left.get(s, 0) - steps_per_poll
                        except NeuTTSHipError as ex:
                            # Running one burst ahead, rows that have finished on the device still count as running on the host and
                            # ntts_backbone_decode reserves KV pages for them (up to one page per slot): with a tightly sized pool that
                            # can fail where the blocking scheduler succeeds.  Nothing was enqueued: drain the snapshot in flight,
                            # release what has finished, and try again on the next iteration.
                            if ex.code != -3:
                                raise
                            burst_failed = ex
                    if open_snap is None:
                        st = nn = None
                        if burst_failed is not None:
                            st, nn = self.poll()                      # blocking: everything enqueued so far
                            q = snap_seq
                            if not any(st[s] == 2 for s in owner if s < self.max_batch):
                                raise burst_failed                    # nothing to drain: the pool really is too small
                    else:
                        st, nn = self.poll_end()
                        q, open_snap = open_snap, None
                        if burst_failed is not None and not any(valid_from.get(s, 0) <= q and st[s] == 2 for s in owner if s < self.max_batch):
                            # the burst could not reserve its pages and the snapshot in flight frees nothing: look at everything
                            # enqueued so far; if no owner has finished there either, no later iteration can cure it (ADVICE r4:
                            # this used to spin on poll_end / poll_begin forever, and an EngineGang with it)
                            st, nn = self.poll()
                            q = snap_seq
                            if not any(st[s] == 2 for s in owner if s < self.max_batch):
                                raise burst_failed
                else:
                    st, nn = self.poll()
                    q = snap_seq                                  # a blocking poll describes everything enqueued so far
                if st is not None:
                    for s in [s for s in list(owner) if s < self.max_batch and valid_from.get(s, 0) <= q and st[s] == 2]:
                        i = owner.pop(s)
                        if on_finished is not None:
                            on_finished(i, s, int(nn[s]))
                            results[i] = []
                        else:
                            results[i] = self.read_finished(s) if run_ahead else self.read(s)[0]
                        committed.pop(s, None)
                        self.release(s)                          # shared pages live on until their last user is released
                        valid_from.pop(s, None)
                        steps_left.pop(s, None)
                        anchors = [a for a in anchors if a[0] != s]
                if run_ahead:
                    self.poll_begin()
                    open_snap = snap_seq
                    snap_seq += 1
                elif owner and (parked or any(st[s] == 1 for s in owner if s < self.max_batch)):
                    activate_parked()
                    self.decode(steps_per_poll)
                yield None
        finally:
            # an exception (KV pool exhausted, a failed launch, ...) must not leave admitted slots RUNNING with their
            # pages held: the next call would find them busy
            if open_snap is not None:
                try:
                    self.poll_end()
                except NeuTTSHipError:
                    pass
            if owner:
                try:
                    self.sync()
                except NeuTTSHipError:
                    pass
                for s in list(owner):
                    try:
                        self.release(s)
                    except NeuTTSHipError:
                        pass
        return [r if r is not None else [] for r in results]


class EngineGang:
    """Several BackboneEngines on ONE set of weights, run side by side on one GPU.  A decode step is a chain of 171 dependent
    launches of 5-20 us each, latency-bound at a quarter of the HBM peak; several such chains on streams of their own fill each
    other's launch gaps and first round trips (MI355X, 256 rows at context 625: 1.57 ms per step alone, 1.11 / 1.02 / 0.96 ms per
    256-row step with two / three / four chains).  The gang creates n - 1 twins of `engine` that read its arena
    (ntts_backbone_share_arena: KV pool, slots, workspaces and step graph per engine, the weights once), gives every engine a lane
    stream of its own for all of its work -- the HIP runtime multiplexes streams onto FOUR hardware queues (more make it slower), and
    two chains in one queue run one after the other, so n <= 4 and the lanes are created back to back -- and alternates between the
    engines' schedulers from the calling thread.  Each request's arithmetic is that of a single engine (same kernels, same
    slots per engine): ids are identical to BackboneEngine.generate's."""

    MAX_ENGINES = 4        # hardware queues the HIP runtime multiplexes streams onto: a fifth chain shares a queue and the gang gets SLOWER

    def __init__(self, engine: "BackboneEngine", n: int = 4, lanes: bool = True):
        if not 1 <= n <= self.MAX_ENGINES:
            raise ValueError(f"an engine gang has 1..{self.MAX_ENGINES} engines (one hardware queue each), not {n}")
        self.lib, self._device = engine.lib, engine._device
        self._streams: List[int] = []
        self.engines: List[BackboneEngine] = [engine]
        try:
            for _ in range(n - 1):
                self.engines.append(engine.twin())
            if lanes and n > 1:
                for _ in range(n):                       # created back to back: up to four streams land in distinct hardware queues
                    st = C.c_void_p()
                    rc = self.lib.ntts_stream_create(self._device, C.byref(st))
                    if rc != 0:
                        raise NeuTTSHipError(rc, "ntts_stream_create failed")
                    self._streams.append(st.value)
                for e, st in zip(self.engines, self._streams):
                    e.set_stream(st)
            # (the decode step's tile family depends on how many chains share the chip: since ABI 9 every engine counts them itself at each
            #  decode call -- ntts_backbone_set_gang only pins the count for tests and sweeps -- so an engine of the gang that decodes alone
            #  for a while, e.g. NeuTTS.infer on engine 0, runs the single-chain shape meanwhile)
        except Exception:
            self.close()                                 # twins and lanes made so far must not leak
            raise

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lane(self, k: int) -> Optional[int]:
        """Engine k's lane stream (lend it to that engine's codec engine too: CodecEngine.set_stream), None without lanes."""
        return self._streams[k] if self._streams else None

    @property
    def max_batch(self) -> int:
        return sum(e.max_batch for e in self.engines)

    def warm_up(self, decode_steps: int = 2):
        for e in self.engines:
            e.warm_up(decode_steps)

    def generate(self, prompts: Sequence[Sequence[int]], sampling, on_finished=None, steps_per_poll: int = 1, admit: str = "wave",
                 on_admit=None, **kw) -> List[List[int]]:
        """BackboneEngine.generate over the gang: request i goes to engine i % n (prompts of one speaker that follow each other n apart
        still share their prefix pages inside an engine), the engines' schedulers advance in turn.  Bursts of `steps_per_poll` = 1 step
        by default: the engines' bursts are enqueued in turn, and the shorter the turn the closer their chains run side by side (8192
        ragged requests, round 4: 144.5 k codec-tokens/s at 1-2 steps, 132.0 k at 4, 106.7 k at 8, profiles/r04s_sweep_continuous_gang_sched.txt;
        round 5 with admission waves: 149.8 k at 1, 146.3-147.0 k at 2).  on_finished(request index, slot, n_new, engine) -- the engine is
        passed along for the device-side hand-off.  Returns the new ids in request order.
        admit: how the engines' prompt passes are placed against each other once `min_admit` > 1 (a pass holds its engine's lane for
        ~ 0.8 us per prompt token while the other chains go on).  "wave[:F[:W]]" (default) -- when one engine admits `min_admit`
        prompts, the others admit within W rounds (default 1) with min_admit / F (default a third) of their slots free: all prompt
        passes at once, then four decode chains side by side again, the static schedule's pattern -- 149.8 k against 145.6 k for
        "independent" (each engine admits when `min_admit` of ITS slots are free; steady state 156.3 k against 149.7 k), flat over
        min_admit 20-28, F 2-3, W 1-2; "spaced:R" -- not within R rounds of ANY engine's last pass (at most one engine out of the
        decode gang at a time): 142.0 k at R = 3, 121.5 k at 6 (profiles/r05m_sweep_continuous_admission_*.txt).  Scheduling only:
        ids cannot differ.  on_admit(engine, n_prompts): called after each prompt pass of an engine is enqueued (the moment to put
        other lane-holding work of that engine -- a codec pass over what it has finished -- next to the wave)."""
        n = len(self.engines)
        sched = {"round": 0, "last": -(1 << 30)}
        space, follow, window = 0, 3, 1
        try:
            kind, *num = admit.split(":")
            num = [int(x) for x in num]
            if kind == "spaced" and len(num) == 1 and num[0] > 0:
                space = num[0]
            elif kind == "wave" and len(num) <= 2 and all(x > 0 for x in num):
                follow, window = (num + [3, 1][len(num):])
            elif kind != "independent" or num:
                raise ValueError
        except ValueError:
            raise ValueError(f"admit = {admit!r}: 'independent', 'spaced:<rounds>' or 'wave[:<follower divisor>[:<rounds>]]'") from None

        def gate(n_free, want):
            if space:
                return n_free >= want and sched["round"] - sched["last"] >= space
            return n_free >= want or (sched["round"] - sched["last"] <= window and n_free >= max(1, want // follow))

        def admitted(n_prompts):
            if space or n_prompts >= kw.get("min_admit", 1):     # (a follower's small pass does not re-arm the wave)
                sched["last"] = sched["round"]
        if admit != "independent":
            kw = dict(kw, admit_gate=gate)
        if isinstance(sampling, Sampling):
            sampling = [sampling] * len(prompts)
        parts = [list(range(k, len(prompts), n)) for k in range(n)]
        its = []
        for e, idx in zip(self.engines, parts):
            if not idx:
                continue
            hook = None
            if on_finished is not None:
                hook = (lambda i, slot, n_new, _e=e, _idx=idx: on_finished(_idx[i], slot, n_new, _e))
            note = None
            if admit != "independent" or on_admit is not None:
                def note(n_prompts, _e=e):
                    if admit != "independent":
                        admitted(n_prompts)
                    if on_admit is not None:
                        on_admit(_e, n_prompts)
            its.append((idx, e.generate_iter([prompts[i] for i in idx], [sampling[i] for i in idx], steps_per_poll=steps_per_poll, on_finished=hook,
                                             on_admit=note, **kw)))
        results: List[List[int]] = [[] for _ in prompts]
        try:
            while its:
                sched["round"] += 1
                for item in list(its):
                    idx, it = item
                    try:
                        next(it)
                    except StopIteration as done:
                        for i, r in zip(idx, done.value):
                            results[i] = r
                        its.remove(item)
        finally:
            for _, it in its:
                it.close()                                # an exception in one engine: the others release their slots (generate_iter's finally)
        return results

    def sync(self):
        for e in self.engines:
            e.sync()

    def close(self):
        """Twins destroyed, engine 0 back on its own stream and its single-chain tiles, lanes destroyed.  Idempotent."""
        engines, self.engines = getattr(self, "engines", []), getattr(self, "engines", [])[:1]
        for e in reversed(engines[1:]):
            e.close()
        if engines and getattr(engines[0], "h", None):
            if self._streams:
                engines[0].set_stream(None)
            engines[0].set_gang(0)
        for st in self._streams:
            self.lib.ntts_stream_destroy(self._device, C.c_void_p(st))
        self._streams = []


class CodecEngine:
    """NeuCodec decoder on one GPU: codes -> 24 kHz waveform (replaces codec.decode_code, ref:neutts/neutts.py:288-291)."""

    def __init__(self, cfg: dict, device: int = 0, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        lv = list(cfg.get("levels", [4] * 8))
        c = CodecConfigC(cfg.get("hidden_size", 1024), cfg.get("intermediate_size", 4096), cfg.get("num_layers", 12),
                         cfg.get("num_heads", 16), cfg.get("head_dim", 64), cfg.get("quantization_dim", 2048), len(lv),
                         (C.c_int32 * 8)(*(lv + [1] * (8 - len(lv)))), cfg.get("hop_length", 480), cfg.get("rms_eps", 1e-6),
                         cfg.get("max_frames", 2048), cfg.get("max_rows", 4096),
                         {"fp16": 0, "high": 1, "bf16": 2, 0: 0, 1: 1, 2: 2}[cfg.get("precision", "fp16")])
        h = C.c_void_p()
        rc = self.lib.ntts_codec_create(C.byref(c), device, C.byref(h))
        if rc != 0:
            raise NeuTTSHipError(rc, (self.lib.ntts_codec_last_error(None) or b"").decode())
        self.h = h
        self.hop_length = c.hop_length
        self.max_frames, self.max_rows = c.max_frames, c.max_rows
        self.device = device

    def _chk(self, rc: int):
        if rc != 0:
            raise NeuTTSHipError(rc, (self.lib.ntts_codec_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "_pin_ptr", None):
            self.lib.ntts_host_free(self._pin_ptr)
            self._pin_ptr, self._pin_cap = None, 0
        if getattr(self, "h", None):
            self.lib.ntts_codec_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd: Dict[str, object]):
        for k, v in sd.items():
            ptr, code, shape, is_dev, keep = _tensor_ptr(v)
            shp = (C.c_int64 * len(shape))(*shape)
            self._chk(self.lib.ntts_codec_load_tensor(self.h, k.encode(), C.c_void_p(ptr), code, shp, len(shape), is_dev))
            del keep
        self._chk(self.lib.ntts_codec_finalize(self.h))

    def _pinned(self, n_floats: int) -> np.ndarray:
        """Engine-owned page-locked staging buffer (grown on demand), viewed as a float32 numpy array."""
        if getattr(self, "_pin_cap", 0) < n_floats:
            if getattr(self, "_pin_ptr", None):
                self.lib.ntts_host_free(self._pin_ptr)
            ptr = C.c_void_p()
            if self.lib.ntts_host_alloc(n_floats * 4, C.byref(ptr)) != 0:
                raise MemoryError("pinned host allocation failed")
            self._pin_ptr, self._pin_cap = ptr, n_floats
        return np.ctypeslib.as_array((C.c_float * n_floats).from_address(self._pin_ptr.value))

    def decode(self, codes: Sequence[Sequence[int]], reuse_output: bool = False) -> List[np.ndarray]:
        """codes: one int sequence per utterance -> list of float32 waveforms (hop_length * len each).
        reuse_output=True returns views into an engine-owned pinned buffer (fast D2H, no copy): they are only valid
        until the next decode() call on this engine."""
        out: List[Optional[np.ndarray]] = [None] * len(codes)
        order = sorted(range(len(codes)), key=lambda i: -len(codes[i]))   # batch similar lengths together
        i = 0
        i32p = C.POINTER(C.c_int32)
        while i < len(order):
            tmax = len(codes[order[i]])
            nb = max(1, min(len(order) - i, self.max_rows // (tmax + 6)))
            grp = order[i:i + nb]
            lens = np.array([len(codes[j]) for j in grp], dtype=np.int32)
            flat = np.ascontiguousarray(np.concatenate([np.asarray(codes[j], dtype=np.int32) for j in grp]))
            stride = int(self.hop_length * tmax)
            if reuse_output and i == 0 and nb == len(order):      # single call covers the batch: pinned fast path
                wav = self._pinned(len(grp) * stride).reshape(len(grp), stride)
            else:
                wav = np.empty((len(grp), stride), dtype=np.float32)
            self._chk(self.lib.ntts_codec_decode(self.h, len(grp), flat.ctypes.data_as(i32p), lens.ctypes.data_as(i32p),
                                                 wav.ctypes.data_as(C.POINTER(C.c_float)), stride))
            for r, j in enumerate(grp):
                out[j] = wav[r, : self.hop_length * len(codes[j])]     # view into this call's buffer: no second copy
            i += nb
        return out  # type: ignore[return-value]

    def decode_array(self, codes: np.ndarray, lens: Optional[np.ndarray] = None, reuse_output: bool = False) -> np.ndarray:
        """Batch fast path: codes [n, T] int (row i valid up to lens[i]; all T when lens is None) -> waveforms
        [n, hop_length * T] float32 in ONE engine call (n * (T + 6) must fit max_rows).  reuse_output=True returns a
        view of the engine's pinned staging buffer, valid until the next decode on this engine."""
        codes = np.asarray(codes)
        n, T = codes.shape
        if lens is None:
            lens = np.full(n, T, dtype=np.int32)
            flat = np.ascontiguousarray(codes, dtype=np.int32).reshape(-1)
        else:
            lens = np.ascontiguousarray(lens, dtype=np.int32)
            flat = np.ascontiguousarray(np.concatenate([codes[i, : lens[i]] for i in range(n)]), dtype=np.int32)
        stride = int(self.hop_length * int(lens.max()))
        wav = (self._pinned(n * stride) if reuse_output else np.empty(n * stride, dtype=np.float32)).reshape(n, stride)
        i32p = C.POINTER(C.c_int32)
        self._chk(self.lib.ntts_codec_decode(self.h, n, flat.ctypes.data_as(i32p), lens.ctypes.data_as(i32p),
                                             wav.ctypes.data_as(C.POINTER(C.c_float)), stride))
        return wav

    def set_debug(self, keep_stages: bool):
        self._chk(self.lib.ntts_codec_set_debug(self.h, int(keep_stages)))

    STAGES = ["embed", "prior", "layers", "post"]            # the tap names of oracle/codec_ref.py decode_code(taps=...)

    def read_stage(self, stage: int, utt: int = 0) -> np.ndarray:
        """fp32 residual stream [frames, hidden] of utterance `utt` after stage `stage` of the most recent decode call (set_debug(True) first)."""
        buf = np.empty(self.max_frames * 4096, dtype=np.float32)
        r, c_ = C.c_int32(), C.c_int32()
        self._chk(self.lib.ntts_codec_read_stage(self.h, stage, utt, buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size, C.byref(r), C.byref(c_)))
        return buf[: r.value * c_.value].reshape(r.value, c_.value).copy()

    def set_cu_mask(self, mask_words: Optional[Sequence[int]]):
        """Restrict the codec engine's stream to the CUs of `mask_words` (None / empty = all); ntts_codec_set_cu_mask."""
        words = list(mask_words or [])
        arr = (C.c_uint32 * max(1, len(words)))(*words)
        self._chk(self.lib.ntts_codec_set_cu_mask(self.h, arr, len(words)))

    def decode_device(self, codes_dev_ptr: int, codes_stride: int, lens: np.ndarray, producer_stream: int = 0,
                      wav_dev_ptr: Optional[int] = None, wav_stride: Optional[int] = None, reuse_output: bool = True):
        """Codes already on the device (BackboneEngine.export_codes) -> waveforms, asynchronously: returns the [n, stride]
        float32 destination (a view of the engine's pinned host buffer, or None when `wav_dev_ptr` names a device buffer);
        call sync() before reading it."""
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        n = len(lens)
        stride = int(wav_stride or self.hop_length * int(lens.max()))
        wav = None
        if wav_dev_ptr is None:
            wav = self._pinned(n * stride).reshape(n, stride)
            dst, on_dev = wav.ctypes.data, 0
        else:
            dst, on_dev = wav_dev_ptr, 1
        self._chk(self.lib.ntts_codec_decode_dev(self.h, n, C.c_void_p(codes_dev_ptr), codes_stride,
                                                 lens.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(dst), stride, on_dev,
                                                 C.c_void_p(producer_stream or None)))
        return wav

    def sync(self):
        self._chk(self.lib.ntts_codec_sync(self.h))

    def set_stream(self, stream: Optional[int]):
        """Run the codec passes on the caller's HIP stream (a hipStream_t as an integer); None = the engine's own."""
        self._chk(self.lib.ntts_codec_set_stream(self.h, C.c_void_p(stream or None)))

    def last_timing(self) -> float:
        ms = C.c_float()
        self._chk(self.lib.ntts_codec_last_timing(self.h, C.byref(ms)))
        return ms.value


class StreamSet:
    """Device-side streaming state of `n` concurrent infer_stream utterances (ntts_streams_*, ABI 6): token caches, window assembly,
    the 27-frame slice and the cross-fade with the previous chunk (ref:neutts/neutts.py:385-388, :401-465) run on the device; per burst
    the host sees a few integers per stream and receives each stream's new samples."""

    def __init__(self, backbone: "BackboneEngine", codec: "CodecEngine", slots: Sequence[int], ref_codes: Sequence[Sequence[int]],
                 max_new_tokens: int, chunk: int, lookforward: int, lookback: int, overlap: int, hop_length: int, speech_base: int,
                 n_codes: int, modulo: bool = False):
        self.lib = backbone.lib
        self.n = len(slots)
        prm = StreamParamsC(chunk, lookforward, lookback, overlap, hop_length, speech_base, n_codes, int(bool(modulo)))
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        lens = np.array([len(r) for r in ref_codes], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.int32) for r in ref_codes]) if lens.sum() else np.zeros(1, np.int32))
        i32p = C.POINTER(C.c_int32)
        h = C.c_void_p()
        rc = self.lib.ntts_streams_create(backbone.h, codec.h, C.byref(prm), backbone._device, self.n, sl.ctypes.data_as(i32p),
                                          flat.ctypes.data_as(i32p), lens.ctypes.data_as(i32p), int(max_new_tokens), C.byref(h))
        if rc != 0:
            raise NeuTTSHipError(rc, (self.lib.ntts_streams_last_error(None) or b"").decode())
        self.h = h
        self._keep = (backbone, codec)                                   # the set borrows both engines
        cap = 2 * self.n
        self._cs, self._cn, self._cl = (np.zeros(cap, dtype=np.int32) for _ in range(3))

    def _chk(self, rc: int):
        if rc != 0:
            raise NeuTTSHipError(rc, (self.lib.ntts_streams_last_error(self.h) or b"").decode())

    def pump_begin(self):
        """Append the ids generated so far to the caches and snapshot the streams behind the decode steps enqueued so far (async)."""
        self._chk(self.lib.ntts_streams_pump_begin(self.h))

    def pump_wait(self) -> int:
        """Wait for the snapshot of pump_begin alone; returns the number of streams still generating in it."""
        run = C.c_int32()
        self._chk(self.lib.ntts_streams_pump_wait(self.h, C.byref(run)))
        return run.value

    def pump_end(self):
        """-> (chunks, n_running, more): chunks = [(stream, samples (view of pinned memory, valid until the next pump_end), last)]."""
        i32p = C.POINTER(C.c_int32)
        n, run, more = C.c_int32(), C.c_int32(), C.c_int32()
        ptr, stride = C.POINTER(C.c_float)(), C.c_int64()
        self._chk(self.lib.ntts_streams_pump_end(self.h, len(self._cs), C.byref(n), self._cs.ctypes.data_as(i32p), self._cn.ctypes.data_as(i32p),
                                                 self._cl.ctypes.data_as(i32p), C.byref(ptr), C.byref(stride), C.byref(run), C.byref(more)))
        out = []
        if n.value:
            buf = np.ctypeslib.as_array(ptr, shape=(n.value, stride.value))
            out = [(int(self._cs[k]), buf[k, : int(self._cn[k])], bool(self._cl[k])) for k in range(n.value)]
        return out, run.value, bool(more.value)

    def done(self) -> int:
        d = C.c_int32()
        self._chk(self.lib.ntts_streams_done(self.h, C.byref(d)))
        return d.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.ntts_streams_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EncoderEngine:
    """NeuCodec encoder on one GPU: 16 kHz mono waveform -> FSQ codes at 50 Hz (replaces codec.encode_code,
    ref:neutts/neutts.py:266-271).  One clip per call; fp32."""

    STAGES = {"features": 0, "concat": 1, "fc": 2, "latents": 3}

    def __init__(self, cfg: dict, device: int = 0, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        ratios = list(cfg.get("ratios", [2, 2, 4, 4, 5]))
        lv = list(cfg.get("levels", [4] * 8))
        self.hop = int(np.prod(ratios))
        self.max_samples = int(cfg.get("max_samples", 30 * 16000))
        c = EncoderConfigC(cfg.get("sem_hidden", 1024), cfg.get("sem_layers", 16), cfg.get("sem_heads", 16),
                           cfg.get("sem_ffn", 4096), cfg.get("sem_conv_kernel", 31), cfg.get("sem_left", 64),
                           cfg.get("sem_right", 8), cfg.get("sem_ln_eps", 1e-5), cfg.get("ac_hidden", 48), len(ratios),
                           (C.c_int32 * 8)(*(ratios + [1] * (8 - len(ratios)))), cfg.get("codec_hidden", 1024), len(lv),
                           (C.c_int32 * 8)(*(lv + [1] * (8 - len(lv)))), self.max_samples)
        h = C.c_void_p()
        rc = self.lib.ntts_encoder_create(C.byref(c), device, C.byref(h))
        if rc != 0:
            raise NeuTTSHipError(rc, (self.lib.ntts_encoder_last_error(None) or b"").decode())
        self.h = h
        self.device = device

    def _chk(self, rc: int):
        if rc != 0:
            raise NeuTTSHipError(rc, (self.lib.ntts_encoder_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.ntts_encoder_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd: Dict[str, object]):
        """Encoder tensors of an Xcodec2Model-named state dict; decoder / project_out entries are skipped here (they belong
        to CodecEngine), anything else unknown is an error at finalize only if a needed tensor is missing."""
        for k, v in sd.items():
            if k.startswith(("acoustic_decoder.", "decoder.", "quantizer.project_out.")) or k.endswith("masked_spec_embed"):
                continue
            ptr, code, shape, is_dev, keep = _tensor_ptr(v)
            shp = (C.c_int64 * len(shape))(*shape)
            self._chk(self.lib.ntts_encoder_load_tensor(self.h, k.encode(), C.c_void_p(ptr), code, shp, len(shape), is_dev))
            del keep
        self._chk(self.lib.ntts_encoder_finalize(self.h))

    def encode(self, wav: np.ndarray) -> np.ndarray:
        """wav: float32 mono at 16 kHz, [L] -> int32 codes [L // hop + 1]."""
        w = np.ascontiguousarray(np.asarray(wav, dtype=np.float32).reshape(-1))
        if w.size < 1:
            raise ValueError("empty waveform")
        if w.size > self.max_samples:
            raise ValueError(f"clip of {w.size} samples exceeds max_samples {self.max_samples}")
        cap = w.size // self.hop + 2
        out = np.empty(cap, dtype=np.int32)
        n = C.c_int32()
        self._chk(self.lib.ntts_encoder_encode(self.h, w.ctypes.data_as(C.POINTER(C.c_float)), w.size,
                                               out.ctypes.data_as(C.POINTER(C.c_int32)), cap, C.byref(n)))
        return out[: n.value].copy()

    def read_stage(self, name: str) -> np.ndarray:
        """Intermediate of the last encode() call ('features' | 'concat' | 'fc' | 'latents') as [T, C] float32."""
        cap = (self.max_samples // self.hop + 2) * 4096
        buf = np.empty(cap, dtype=np.float32)
        r, c = C.c_int32(), C.c_int32()
        self._chk(self.lib.ntts_encoder_read_stage(self.h, self.STAGES[name], buf.ctypes.data_as(C.POINTER(C.c_float)), cap,
                                                   C.byref(r), C.byref(c)))
        return buf[: r.value * c.value].reshape(r.value, c.value).copy()

    def last_timing(self) -> float:
        ms = C.c_float()
        self._chk(self.lib.ntts_encoder_last_timing(self.h, C.byref(ms)))
        return ms.value

