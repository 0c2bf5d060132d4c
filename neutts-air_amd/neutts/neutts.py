"""NeuTTS -- the reference's class surface (ref:neutts/neutts.py:73-465) over the MI355X-native hot path.

Same constructor arguments, public attributes, `infer` / `infer_stream` / `encode_reference` signatures, return
types and error messages as the reference; the two third-party calls on its hot path are replaced:

    self.backbone.generate(...)      ref:neutts/neutts.py:338-347   ->  _hip.BackboneEngine (HIP, paged KV, batched)
    self.codec.decode_code(codes)    ref:neutts/neutts.py:288-291   ->  _hip.CodecEngine    (HIP)

and the id -> string -> regex -> id round trip between them (ref :349 -> :276) becomes `code = id - id(<|speech_0|>)`
with a range mask (ids outside the speech range are dropped, exactly what the regex does).

Differences, all deliberate:
  * devices: this implementation runs on MI355X only -- "cpu" raises (there is no CPU fallback); defaults are "cuda".
  * `infer_stream` is implemented for this backend (the reference raises NotImplementedError for torch, :264) with
    the GGUF path's window / overlap-add semantics (:401-465).
  * batched entry points (`infer_batch`, `infer_stream_batch`, `generate_codes`, `decode_codes`) expose what the engine is
    built for.
  * GGUF (llama.cpp) backbones and the ONNX codec are other runtimes of the same model and are not provided.
Optional front/back-end dependencies (phonemizer, librosa, neucodec, perth) are imported lazily, where used.
"""
from __future__ import annotations

import re
import warnings
from pathlib import Path
from typing import Dict, Generator, List, Optional, Sequence

import numpy as np

from . import _hip

_SPEECH_RE = re.compile(r"<\|speech_(\d+)\|>")


def _linear_overlap_add(frames: List[np.ndarray], stride: int) -> np.ndarray:
    """Triangular-weight cross-fade of overlapping chunks; behaviour of ref:neutts/neutts.py:46-70."""
    assert len(frames)
    dtype = frames[0].dtype
    total = max(stride * i + f.shape[-1] for i, f in enumerate(frames))
    weight_sum = np.zeros(total, dtype=dtype)
    mixed = np.zeros(total, dtype=dtype)
    for i, f in enumerate(frames):
        n = f.shape[-1]
        t = np.linspace(0, 1, n + 2, dtype=dtype)[1:-1]
        tri = np.abs(0.5 - (t - 0.5))
        mixed[stride * i: stride * i + n] += tri * f
        weight_sum[stride * i: stride * i + n] += tri
    assert weight_sum.min() > 0
    return mixed / weight_sum


class _StreamBlender:
    """Incremental form of the reference's streaming post-process (ref:neutts/neutts.py:441-448, :461-465): the
    reference re-blends its whole audio cache for every chunk (O(n^2)); frames are `chunk + 2*overlap` hops long at a
    stride of `chunk` hops, so only the previous frame can touch the samples a new frame releases.  `push` returns
    exactly `_linear_overlap_add(all frames)[n_decoded_samples:new_end]` (bit-identical: same operands, same order)."""

    def __init__(self, stride: int):
        self.stride = stride
        self.prev: Optional[np.ndarray] = None

    def push(self, frame: np.ndarray, last: bool = False) -> np.ndarray:
        st = self.stride
        if self.prev is None:
            mixed = _linear_overlap_add([frame], stride=st)
            out = mixed if last else mixed[:st]
        else:
            assert self.prev.shape[-1] < 2 * st, "frame longer than two strides: more than one frame overlaps"
            mixed = _linear_overlap_add([self.prev, frame], stride=st)
            out = mixed[st:] if last else mixed[st:2 * st]
        self.prev = frame
        return out


def _device_index(device, what: str) -> int:
    s = str(device)
    if s in ("gpu", "cuda", "hip"):
        return 0
    if s.startswith("cuda:") or s.startswith("hip:"):
        return int(s.split(":", 1)[1])
    raise RuntimeError(
        f"{what}={device!r}: this NeuTTS implementation runs on AMD MI355X (gfx950) only and has no CPU fallback; "
        "pass 'cuda' / 'cuda:N'.")


class _CodecFacade:
    """What `self.codec` looks like to callers of the reference: `.decode_code(codes[B,1,T]) -> wav[B,1,480*T]`,
    `.encode_code(audio_or_path=...) -> codes[B,1,T]` and `.device`.  `enc_engine` = the on-device encoder
    (_hip.EncoderEngine), the only encoder there is: no torch module is kept beside it."""

    def __init__(self, engine: _hip.CodecEngine, enc_engine: Optional[_hip.EncoderEngine] = None):
        self.engine = engine
        self.enc_engine = enc_engine
        self.device = f"cuda:{engine.device}"

    def decode_code(self, codes):
        arr = np.asarray(codes.cpu() if hasattr(codes, "cpu") else codes)
        if arr.ndim != 3 or arr.shape[1] != 1:
            raise ValueError("codes must have shape [B, 1, T]")
        wavs = self.engine.decode([arr[b, 0].tolist() for b in range(arr.shape[0])])
        import torch  # tensor container for API compatibility
        return torch.from_numpy(np.stack(wavs)[:, None, :])

    def encode_code(self, audio_or_path):
        """neucodec `encode_code`: a path, or a 16 kHz waveform [B, 1, L] (tensor / array) -> int codes [B, 1, T]."""
        if self.enc_engine is None:
            raise RuntimeError("Reference encoding needs encoder weights on the encoder engine: a codec spec with an 'encoder' entry "
                               "or a NeuCodec checkpoint; or pre-encode references (ref:examples/encode_reference.py).")
        import torch  # tensor container for API compatibility
        if isinstance(audio_or_path, (str, Path)):
            wavs = [load_audio_16k(audio_or_path)]
        else:
            arr = np.asarray(audio_or_path.detach().cpu() if hasattr(audio_or_path, "detach") else audio_or_path, dtype=np.float32)
            if arr.ndim == 1:
                arr = arr[None, None, :]
            elif arr.ndim == 2:
                arr = arr[:, None, :]
            if arr.ndim != 3 or arr.shape[1] != 1:
                raise ValueError("audio must have shape [B, 1, L] (mono, 16 kHz)")
            wavs = [arr[b, 0] for b in range(arr.shape[0])]
        return torch.from_numpy(np.stack([self.enc_engine.encode(w) for w in wavs])[:, None, :].astype(np.int64))


def load_audio_16k(path) -> np.ndarray:
    """`librosa.load(path, sr=16000, mono=True)` (ref:neutts/neutts.py:268) -> float32 [L].  Without librosa, PCM / float WAV
    files are read with scipy and resampled with a polyphase filter (a front-end detail: the two resamplers are not
    sample-identical); anything else needs librosa, as in the reference."""
    try:
        import librosa
    except ImportError:
        librosa = None
    if librosa is not None:
        wav, _ = librosa.load(str(path), sr=16000, mono=True)
        return np.asarray(wav, dtype=np.float32)
    if not str(path).lower().endswith(".wav"):
        raise ImportError("Reading this audio format needs librosa (pip install librosa), as in the reference.")
    from math import gcd
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    sr, x = wavfile.read(str(path))
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(2 ** (8 * x.dtype.itemsize - 1))
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = x.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)
    if sr != 16000:
        g = gcd(int(sr), 16000)
        x = resample_poly(x, 16000 // g, int(sr) // g).astype(np.float32)
    return x


class NeuTTS:

    def __init__(
        self,
        backbone_repo="neuphonic/neutts-nano",
        backbone_device="cuda",
        codec_repo="neuphonic/neucodec",
        codec_device="cuda",
        *,
        max_batch: int = 1,
        engines: int = 1,
        lib_path: Optional[str] = None,
        do_sample: bool = True,
        seed: int = 0,
        speech_range_head: bool = False,
        codec_precision: str = "fp16",
    ):
        # Consts (ref:neutts/neutts.py:84-91)
        self.sample_rate = 24_000
        self.max_context = 2048
        self.hop_length = 480
        self.streaming_overlap_frames = 1
        self.streaming_frames_per_chunk = 25
        self.streaming_lookforward = 5
        self.streaming_lookback = 50
        self.streaming_stride_samples = self.streaming_frames_per_chunk * self.hop_length
        self.streaming_overlap_compute = True    # backbone decode of chunk k+1 runs beside the codec pass of chunk k

        self._is_quantized_model = False
        self._is_onnx_codec = False
        self._lib_path = lib_path
        self._max_batch = max_batch
        # sampling contract of the reference call (ref:neutts/neutts.py:338-347); do_sample=False = greedy
        self.do_sample = do_sample
        self.top_k = 50
        self.temperature = 1.0
        self.min_new_tokens = 50
        self._seed = seed

        # NeuCodec's GEMM operand format (the reference runs this decoder in fp32, ref:neutts/neutts.py:288-291): "fp16" (default) = IEEE-half
        # operands, ~9.5e-4 RELATIVE rms of the fp32 decoder (8.1e-4 from the GEMM operands, the rest from the single-term fp16 ISTFT) -- inside BASELINE's 1e-3 absolute at any amplitude; "high" = split bf16 operands
        # (hi + lo), ~7e-4 at 3x the matrix-core work and bf16's range; "bf16" = rounds 1-5's default, 7e-3 relative
        if codec_precision not in ("fp16", "bf16", "high"):
            raise ValueError("codec_precision must be 'fp16', 'bf16' or 'high'")
        self._codec_precision = codec_precision
        self.tokenizer = None
        self.phonemizer = None       # created on first use: text front-end is off the hot path
        self._load_backbone(backbone_repo, backbone_device)
        self._load_codec(codec_repo, codec_device)
        # engines > 1 (serving; the reference runs one utterance at a time): that many backbone engines of `max_batch` slots each on
        # ONE copy of the weights, their decode chains side by side on the GPU (neutts._hip.EngineGang) -- infer_batch deals its
        # utterances out over them.  infer / infer_stream stay on the first engine.
        # speech_range_head=True (OPT-IN, off by default; SURVEY 7 "hard parts"): the lm_head over `<|speech_0|>` ... `<|speech_65535|>` +
        # `<|SPEECH_GENERATION_END|>` only -- the ids ref:neutts/neutts.py:276,336-341 can do anything with -- a third of the head's
        # bytes per decode step.  The reference samples over the whole vocabulary: the ids are its ids only while its choice lies in
        # that range (a trained checkpoint after a well-formed prompt), so this is a serving option, not the parity configuration.
        if speech_range_head:
            if self._speech_base is None or self._eos_id is None:
                raise RuntimeError("speech_range_head needs the ids of <|speech_0|> and <|SPEECH_GENERATION_END|> (tokenizer, or 'speech_base' / 'eos_token_id' with in-memory weights)")
            self.backbone.set_logits_range(int(self._speech_base), int(self._speech_base) + 65536, int(self._eos_id))
        self.gang = _hip.EngineGang(self.backbone, engines) if engines > 1 else None

        try:  # optional watermarker, as the reference (ref:neutts/neutts.py:110-121)
            import perth
            self.watermarker = perth.PerthImplicitWatermarker()
        except (ImportError, AttributeError) as e:
            warnings.warn(f"Perth watermarking unavailable: {e}. Audio will not be watermarked. "
                          "Install with: pip install perth>=0.2.0")
            self.watermarker = None

    # ------------------------------------------------------------------------------------------ loading
    def close(self):
        """Release the GPU engines (not part of the reference's surface: its torch modules go with the garbage collector).  The gang
        goes first -- its twins read the first engine's arena and its lane streams carry engine 0's work."""
        gang, self.gang = getattr(self, "gang", None), None
        codecs, self._gang_codecs = getattr(self, "_gang_codecs", None), None
        for k, c in enumerate(codecs or []):
            c.set_stream(None)                       # (they ran on the gang's lanes)
            if k > 0:
                c.close()
        if gang is not None:
            gang.close()
        for name in ("backbone",):
            eng = getattr(self, name, None)
            if eng is not None and hasattr(eng, "close"):
                eng.close()
        codec = getattr(self, "codec", None)             # the class's own codec engine (gang codec 0) and the encoder engine: their GPU buffers and
        for eng in (getattr(codec, "engine", None), getattr(codec, "enc_engine", None)):   # page-locked staging rings do not wait for the collector
            if eng is not None and hasattr(eng, "close"):
                eng.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _load_backbone(self, backbone_repo, backbone_device):
        print(f"Loading backbone from: {backbone_repo if isinstance(backbone_repo, str) else '<in-memory weights>'}"
              f" on {backbone_device} ...")
        dev = _device_index(backbone_device, "backbone_device")
        if isinstance(backbone_repo, dict):
            # in-memory weights (tests, synthetic benchmarks): {"config", "state_dict", "inv_freq", "tokenizer",
            # "speech_base", "eos_token_id"}
            spec = backbone_repo
            cfg, sd, inv_freq = dict(spec["config"]), spec["state_dict"], spec["inv_freq"]
            input_scales = spec.get("input_scales")          # fp8 model (config["weight_dtype"] = "fp8"): static activation scales
            self.tokenizer = spec.get("tokenizer")
            self._speech_base = spec.get("speech_base")
            self._eos_id = spec.get("eos_token_id")
        else:
            if str(backbone_repo).endswith("gguf"):
                raise NotImplementedError(
                    "GGUF backbones run on llama.cpp (another runtime of the same model, quantised numerics); this "
                    "implementation provides the transformers-path backbone on MI355X only.")
            from transformers import AutoConfig, AutoModelForCausalLM, AutoTokenizer  # checkpoint readers only
            self.tokenizer = AutoTokenizer.from_pretrained(backbone_repo)
            hc = AutoConfig.from_pretrained(backbone_repo)
            cfg = _engine_config_from_hf(hc)
            input_scales = None
            sd = inv_freq = None
            shards = _safetensors_shards(backbone_repo)
            theta = _default_rope_theta(hc)
            if shards and theta is not None:
                # stream the checkpoint tensor by tensor into the engine's arena: no fp32 nn.Module copy of the model
                # on the host (the reference builds one, ref:neutts/neutts.py:164; only its weights matter here)
                import torch
                d = cfg["head_dim"]
                inv_freq = (1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.int64).to(dtype=torch.float) / d))).numpy()
                sd = _SafetensorsStateDict(shards)        # hf:modeling_rope_utils.py _compute_default_rope_parameters
                self._backbone_loader = "safetensors"
            else:
                model = AutoModelForCausalLM.from_pretrained(backbone_repo)
                sd = {k: v for k, v in model.state_dict().items() if not k.endswith("inv_freq")}
                inv_freq = model.model.rotary_emb.inv_freq.float().cpu().numpy()
                del model
                self._backbone_loader = "transformers"
            self._speech_base = self.tokenizer.convert_tokens_to_ids("<|speech_0|>")
            self._eos_id = self.tokenizer.convert_tokens_to_ids("<|SPEECH_GENERATION_END|>")
        cfg.setdefault("max_context", self.max_context)
        cfg["max_batch"] = self._max_batch
        cfg.setdefault("max_prefill_tokens", max(2 * self.max_context, 8192))
        self.backbone = _hip.BackboneEngine(cfg, dev, self._lib_path)
        self.backbone.load_state_dict(sd, inv_freq=inv_freq, input_scales=input_scales)
        self._vocab_size = cfg["vocab_size"]

    def _load_codec(self, codec_repo, codec_device):
        print(f"Loading codec from: {codec_repo if isinstance(codec_repo, str) else '<in-memory weights>'} on {codec_device} ...")
        dev = _device_index(codec_device, "codec_device")
        enc_spec = None
        if isinstance(codec_repo, dict):
            cfg, sd = dict(codec_repo["config"]), codec_repo["state_dict"]
            enc_spec = codec_repo.get("encoder")       # {"config": EncoderConfig fields, "state_dict": Xcodec2Model names}
        else:
            match codec_repo:
                case "neuphonic/neucodec" | "neuphonic/distill-neucodec":
                    try:
                        from neucodec import DistillNeuCodec, NeuCodec
                    except ImportError as e:
                        raise ImportError(
                            "Loading the NeuCodec checkpoint needs the `neucodec` package (weights + reference "
                            "encoder): pip install neucodec") from e
                    cls = NeuCodec if codec_repo == "neuphonic/neucodec" else DistillNeuCodec
                    # the package is the checkpoint READER only (its state dict is re-keyed and uploaded; the module is dropped):
                    # a tensor that does not map onto the engines is an error that lists the keys -- there is no torch fallback
                    full_sd = cls.from_pretrained(codec_repo).state_dict()
                    sd = neucodec_to_xcodec2_names(full_sd)
                    cfg = {}
                    enc_spec = {"config": {}, "state_dict": neucodec_encoder_to_xcodec2_names(full_sd)}
                case "neuphonic/neucodec-onnx-decoder":
                    raise NotImplementedError("The ONNX decoder is a CPU runtime of the same decoder; use "
                                              "'neuphonic/neucodec' for the MI355X path.")
                case _:
                    raise ValueError(
                        "Invalid codec repo! Must be one of:"
                        " 'neuphonic/neucodec', 'neuphonic/distill-neucodec',"
                        " 'neuphonic/neucodec-onnx-decoder'."
                    )
        cfg.setdefault("hop_length", self.hop_length)
        cfg.setdefault("precision", self._codec_precision)
        cfg.setdefault("max_frames", self.max_context)
        cfg.setdefault("max_rows", max(2 * (self.max_context + 6), self._max_batch * 512))
        self.hop_length = cfg["hop_length"]
        self.streaming_stride_samples = self.streaming_frames_per_chunk * self.hop_length
        engine = _hip.CodecEngine(cfg, dev, self._lib_path)
        engine.load_state_dict(sd)
        self._codec_spec = (dict(cfg), sd, dev)          # (a gang of stream sets gives every backbone engine a codec engine of its own)
        self._gang_codecs = None
        enc_engine = None
        if enc_spec is not None:
            ecfg = dict(enc_spec.get("config", {}))
            ecfg.setdefault("max_samples", 30 * 16000)       # the context holds ~30 s of audio (ref:README.md:35)
            enc_engine = _hip.EncoderEngine(ecfg, dev, self._lib_path)
            enc_engine.load_state_dict(enc_spec["state_dict"])
        self.codec = _CodecFacade(engine, enc_engine)

    # ------------------------------------------------------------------------------------------ public API
    def infer(self, text: str, ref_codes, ref_text: str) -> np.ndarray:
        """Generate speech for `text` in the voice of the encoded reference (ref:neutts/neutts.py:216-243)."""
        prompt_ids = self._apply_chat_template(ref_codes, ref_text, text)
        new_ids = self.generate_codes([prompt_ids])[0]
        wav = self._decode_ids(new_ids)
        return wav if self.watermarker is None else self.watermarker.apply_watermark(wav, sample_rate=24_000)

    def infer_batch(self, texts: Sequence[str], ref_codes, ref_texts) -> List[np.ndarray]:
        """Many utterances at once: continuous batching over the engine's decode slots (over every engine's, with engines > 1),
        one codec pass."""
        if not isinstance(ref_texts, (list, tuple)):
            ref_texts = [ref_texts] * len(texts)
            ref_codes = [ref_codes] * len(texts)
        prompts = [self._apply_chat_template(rc, rt, t) for rc, rt, t in zip(ref_codes, ref_texts, texts)]
        codes = [self._ids_to_codes(ids) for ids in self.generate_codes(prompts)]
        if any(len(c) == 0 for c in codes):
            raise ValueError("No valid speech tokens found in the output.")
        wavs = self.codec.engine.decode(codes)
        if self.watermarker is not None:
            wavs = [self.watermarker.apply_watermark(w, sample_rate=24_000) for w in wavs]
        return wavs

    def infer_stream(self, text: str, ref_codes, ref_text: str) -> Generator[np.ndarray, None, None]:
        """Streaming synthesis with the reference's window / cross-fade semantics (ref:neutts/neutts.py:373-465)."""
        prompt_ids = self._apply_chat_template(ref_codes, ref_text, text)
        ref = [int(c) for c in _to_list(ref_codes)]
        if self._stream_on_device([ref]):       # token cache, windows, codec pass and cross-fade on the device (csrc/stream.cpp): a set of one stream
            return (chunk for _, chunk in self._infer_stream_batch_hip([prompt_ids], [ref]))
        return self._infer_stream_hip(prompt_ids, ref)        # host loop: a watermarker (a host library) sits between the codec and the slice

    def infer_stream_batch(self, texts: Sequence[str], ref_codes, ref_texts) -> Generator[tuple, None, None]:
        """Many utterances streamed at once (BASELINE config 5's shape: a decode batch with the codec on its own stream):
        yields `(utterance index, chunk)` pairs; the chunks of one utterance, in order, are exactly what `infer_stream`
        yields for it.  One decode burst serves every running utterance, and all windows that became decodable in a burst
        go through the codec in ONE batched call, enqueued behind the next burst (the two engines' streams overlap).
        At most `max_batch` utterances (the engine's decode slots) per call."""
        if not isinstance(ref_texts, (list, tuple)):
            ref_texts = [ref_texts] * len(texts)
            ref_codes = [ref_codes] * len(texts)
        cap = self.gang.max_batch if (self.gang is not None and getattr(self, "stream_on_gang", False)) else self.backbone.max_batch
        if len(texts) > cap:
            raise ValueError(f"{len(texts)} utterances exceed the engine's {cap} decode slots")
        prompts = [self._apply_chat_template(rc, rt, t) for rc, rt, t in zip(ref_codes, ref_texts, texts)]
        return self._infer_stream_batch_hip(prompts, [[int(c) for c in _to_list(rc)] for rc in ref_codes])

    def encode_reference(self, ref_audio_path: str | Path):
        """ref:neutts/neutts.py:266-271: 16 kHz mono -> `codec.encode_code` -> 1-D int codes at 50 Hz.  On the encoder engine
        (kernels/enc.h) when encoder weights are loaded; one-off per speaker, off the synthesis hot path."""
        import torch
        wav = load_audio_16k(ref_audio_path)
        wav_tensor = torch.from_numpy(wav).float().unsqueeze(0).unsqueeze(0)  # [1, 1, T]
        with torch.no_grad():
            ref_codes = self.codec.encode_code(audio_or_path=wav_tensor).squeeze(0).squeeze(0)
        return ref_codes

    # ------------------------------------------------------------------------------------------ id-level hot path
    def _sampling(self, prompt_len: int, index: int = 0) -> _hip.Sampling:
        if self._eos_id is None:
            raise RuntimeError("eos token id unknown: supply 'eos_token_id' with in-memory weights")
        # one Philox key per request: (call counter, index within the call)
        seed = (self._seed * 0x9E3779B97F4A7C15 + index * 0xD1B54A32D192ED03 + 1) & 0xFFFFFFFFFFFFFFFF
        return _hip.Sampling(max_length=self.max_context, min_new_tokens=self.min_new_tokens, eos_token_id=self._eos_id,
                             do_sample=self.do_sample, top_k=self.top_k, temperature=self.temperature, seed=seed)

    def generate_codes(self, prompts: Sequence[Sequence[int]]) -> List[List[int]]:
        """Batched equivalent of `_infer_torch` (ref:neutts/neutts.py:334-352): new token ids per prompt."""
        self._seed += 1
        # utterances of one speaker start with the same tokens (chat header + reference-text phones, ref :307,:315-325):
        # the engine keeps one copy of those KV pages and computes only what differs
        sampling = [self._sampling(len(p), i) for i, p in enumerate(prompts)]
        # more utterances than decode slots: freed slots are refilled about a tenth of an engine's slots at a time (24 of 256) -- a prompt
        # pass over a handful of prompts costs 1.5-2 us per token against 0.84 for 32 and more, an idle slot a step each
        # (profiles/r05m_probe_prefill_size.txt); scheduling only, the ids do not depend on it
        eng_slots = self.backbone.max_batch
        slots = eng_slots * (len(self.gang.engines) if self.gang is not None else 1)
        min_admit = max(1, min(24, eng_slots // 10)) if len(prompts) > slots else 1
        if self.gang is not None and len(prompts) > 1:
            return self.gang.generate(prompts, sampling, share_prefix=True, min_admit=min_admit)
        return self.backbone.generate(prompts, sampling, share_prefix=len(prompts) > 1, min_admit=min_admit)

    def _ids_to_codes(self, ids: Sequence[int]) -> List[int]:
        """ref :349 (tokenizer.decode) + :276 (regex): keep `<|speech_N|>` tokens, N = id - id(<|speech_0|>)."""
        if self._speech_base is not None:
            n_codes = 65536
            return [i - self._speech_base for i in ids if self._speech_base <= i < self._speech_base + n_codes]
        text = self.tokenizer.decode(list(ids), add_special_tokens=False)
        return [int(n) for n in _SPEECH_RE.findall(text)]

    def _ids_to_codes_array(self, ids: np.ndarray) -> np.ndarray:
        """`_ids_to_codes` on an id array (the batched stream's per-burst hand-off): the range test of the regex, vectorised, when
        the speech-token base is known and `_ids_to_codes` has not been replaced on the instance."""
        if self._speech_base is not None and "_ids_to_codes" not in self.__dict__:
            x = np.asarray(ids, dtype=np.int64) - self._speech_base
            return x[(x >= 0) & (x < 65536)].astype(np.int32)
        return np.asarray(self._ids_to_codes(np.asarray(ids).tolist()), dtype=np.int32)

    def decode_codes(self, codes: Sequence[Sequence[int]]) -> List[np.ndarray]:
        return self.codec.engine.decode(codes)

    def _decode_ids(self, ids: Sequence[int]) -> np.ndarray:
        speech_ids = self._ids_to_codes(ids)
        if len(speech_ids) > 0:
            return self.codec.engine.decode([speech_ids])[0]
        raise ValueError("No valid speech tokens found in the output.")

    def _decode(self, codes: str) -> np.ndarray:
        """String form kept for callers of the reference's private helper (ref:neutts/neutts.py:273-295)."""
        speech_ids = [int(num) for num in _SPEECH_RE.findall(codes)]
        if len(speech_ids) > 0:
            return self.codec.engine.decode([speech_ids])[0]
        raise ValueError("No valid speech tokens found in the output.")

    # ------------------------------------------------------------------------------------------ text front-end
    def _to_phones(self, text: str) -> str:
        if self.phonemizer is None:
            try:
                from phonemizer.backend import EspeakBackend
            except ImportError as e:
                raise ImportError("Text input needs `phonemizer` + espeak-ng (ref:requirements.txt:4); id-level entry "
                                  "points (generate_codes / decode_codes) do not.") from e
            print("Loading phonemizer...")
            self.phonemizer = EspeakBackend(language="en-us", preserve_punctuation=True, with_stress=True)
        phones = self.phonemizer.phonemize([text])
        return " ".join(phones[0].split())

    def _apply_chat_template(self, ref_codes, ref_text: str, input_text: str) -> List[int]:
        """Prompt ids exactly as ref:neutts/neutts.py:303-332 builds them."""
        tk = self.tokenizer
        input_text = self._to_phones(ref_text) + " " + self._to_phones(input_text)
        speech_replace = tk.convert_tokens_to_ids("<|SPEECH_REPLACE|>")
        speech_gen_start = tk.convert_tokens_to_ids("<|SPEECH_GENERATION_START|>")
        text_replace = tk.convert_tokens_to_ids("<|TEXT_REPLACE|>")
        text_prompt_start = tk.convert_tokens_to_ids("<|TEXT_PROMPT_START|>")
        text_prompt_end = tk.convert_tokens_to_ids("<|TEXT_PROMPT_END|>")
        input_ids = tk.encode(input_text, add_special_tokens=False)
        chat = """user: Convert the text to speech:<|TEXT_REPLACE|>\nassistant:<|SPEECH_REPLACE|>"""
        ids = tk.encode(chat)
        i = ids.index(text_replace)
        ids = ids[:i] + [text_prompt_start] + input_ids + [text_prompt_end] + ids[i + 1:]
        j = ids.index(speech_replace)
        codes_str = "".join(f"<|speech_{c}|>" for c in _to_list(ref_codes))
        codes = tk.encode(codes_str, add_special_tokens=False)
        return ids[:j] + [speech_gen_start] + list(codes)

    # ------------------------------------------------------------------------------------------ streaming
    def _infer_stream_hip(self, prompt_ids: List[int], ref_codes: List[int]) -> Generator[np.ndarray, None, None]:
        """Window / cross-fade semantics of ref:neutts/neutts.py:401-465 (Appendix C of SURVEY.md), token by token.
        The backbone and the codec are separate engines on separate HIP streams: the next `chunk` decode steps are
        enqueued (asynchronously) BEFORE the codec pass of the chunk that just became decodable, so the codec runs
        beside the backbone instead of between its bursts; the host only blocks in `read()`."""
        eng = self.backbone
        self._seed += 1
        slot = eng.acquire_slot()         # from the engine's pool: a stream that is still open never shares its slot
        try:
            eng.prefill([prompt_ids], [slot], [self._sampling(len(prompt_ids))])
        except Exception:
            eng.release(slot)
            raise
        token_cache: List[int] = list(ref_codes)          # codec codes (the reference caches "<|speech_N|>" strings)
        n_decoded_tokens = len(ref_codes)
        n_seen = 0
        hop = self.hop_length
        stride = self.streaming_stride_samples
        chunk, look_f, look_b, ovl = (self.streaming_frames_per_chunk, self.streaming_lookforward,
                                      self.streaming_lookback, self.streaming_overlap_frames)

        blend = _StreamBlender(stride)

        try:
            finished = False
            while not finished:
                ids, finished = eng.read(slot)            # blocks until the steps enqueued so far are done
                if not finished and self.streaming_overlap_compute:
                    eng.decode(chunk)                     # async: runs while the codec below decodes the last chunk
                new = self._ids_to_codes(ids[n_seen:])
                n_seen = len(ids)
                for c in new:
                    token_cache.append(c)
                    if len(token_cache) - n_decoded_tokens >= chunk + look_f:
                        t0 = max(n_decoded_tokens - look_b - ovl, 0)
                        t1 = n_decoded_tokens + chunk + look_f + ovl
                        s0 = (n_decoded_tokens - t0) * hop
                        s1 = s0 + (chunk + 2 * ovl) * hop
                        recon = self.codec.engine.decode([token_cache[t0:t1]])[0]
                        if self.watermarker is not None:
                            recon = self.watermarker.apply_watermark(recon, sample_rate=24_000)
                        n_decoded_tokens += chunk
                        yield blend.push(np.array(recon[s0:s1]))
                if not finished and not self.streaming_overlap_compute:
                    eng.decode(chunk)                     # serial variant (A/B aid): backbone waits for the codec
            remaining = len(token_cache) - n_decoded_tokens
            if remaining > 0:
                t0 = max(len(token_cache) - (look_b + ovl + remaining), 0)
                s0 = (len(token_cache) - t0 - remaining - ovl) * hop
                recon = self.codec.engine.decode([token_cache[t0:]])[0]
                if self.watermarker is not None:
                    recon = self.watermarker.apply_watermark(recon, sample_rate=24_000)
                yield blend.push(np.array(recon[s0:]), last=True)
        finally:
            eng.sync()
            eng.release(slot)


    # ---- device-side streaming (csrc/stream.cpp, ABI 6)
    def _stream_on_device(self, ref_codes) -> bool:
        # the device path covers what the class does by default: ids -> codes by the speech-token range (or the synthetic benchmark's
        # modulo rule), no watermarker (Perth runs on the host, ref:neutts/neutts.py:422-425), every stream opened by >= overlap codes
        if self.watermarker is not None or not getattr(self, "stream_on_device", True):
            return False
        if not getattr(self, "_stream_modulo", 0) and (self._speech_base is None or "_ids_to_codes" in self.__dict__):
            return False
        if self.streaming_lookforward < 2 * self.streaming_overlap_frames or 2 * self.streaming_overlap_frames >= self.streaming_frames_per_chunk:
            return False                   # (ntts_streams_create refuses these; the host loop slices like the reference does)
        return min(len(rc) for rc in ref_codes) >= self.streaming_overlap_frames

    def _stream_batch_device(self, slots, ref_codes):
        """The loop of _infer_stream_batch_hip with the token caches, the window assembly, the 27-frame slice and the cross-fade on the
        device (ntts_streams_*): per burst the host waits for a snapshot of 2 n integers, enqueues the next burst, and receives each
        stream's new samples.  Same windows, same arithmetic (kernels/stream.h restates numpy's operations one by one): the chunks are
        bit-identical to the host path's."""
        eng = self.backbone
        chunk, look_f = self.streaming_frames_per_chunk, self.streaming_lookforward
        mod = int(getattr(self, "_stream_modulo", 0))
        ss = _hip.StreamSet(eng, self.codec.engine, slots, ref_codes, int(eng.cfg["max_context"]), chunk, look_f, self.streaming_lookback,
                            self.streaming_overlap_frames, self.hop_length, 0 if mod else int(self._speech_base), mod or 65536, bool(mod))
        try:
            # the first window is complete once chunk + lookforward tokens exist (ref :401-404) and the prompt pass produced one of them
            steps = chunk + look_f - 1
            while True:
                ss.pump_begin()                           # behind everything enqueued so far
                running = ss.pump_wait()
                if running and self.streaming_overlap_compute:
                    eng.decode(steps)                     # async: runs beside the codec passes of pump_end
                more = True
                while more:
                    chunks, _, more = ss.pump_end()
                    for i, samples, _last in chunks:
                        yield i, np.array(samples)        # (the pinned buffer is reused by the next pump)
                if running and not self.streaming_overlap_compute:
                    eng.decode(steps)
                if not running:
                    break
                steps = chunk
        finally:
            ss.close()

    def _gang_codec_engines(self):
        """One codec engine per engine of the gang, each on that engine's lane stream (engine k's codec passes in engine k's hardware
        queue: four queues, four lanes -- _hip.EngineGang); the first one is the class's own."""
        if self._gang_codecs is None:
            cfg, sd, dev = self._codec_spec
            engs = [self.codec.engine]
            for _ in self.gang.engines[1:]:
                c = _hip.CodecEngine(cfg, dev, self._lib_path)
                c.load_state_dict(sd)
                engs.append(c)
            for k, c in enumerate(engs):
                if self.gang.lane(k):
                    c.set_stream(self.gang.lane(k))
            self._gang_codecs = engs
            self._gang_codec0_lent = True
        elif not getattr(self, "_gang_codec0_lent", False) and self.gang.lane(0):
            self._gang_codecs[0].set_stream(self.gang.lane(0))     # (handed back at the end of the previous gang call)
            self._gang_codec0_lent = True
        return self._gang_codecs

    def _infer_stream_batch_gang(self, prompts: List[List[int]], ref_codes: List[List[int]]):
        """infer_stream_batch over an ENGINE GANG (BASELINE configs[4]'s literal shape: ref:neutts/neutts.py:373-465
        for many utterances at once).  The utterances are dealt out in admission groups over the gang's engines; every group is a
        device-side stream set (ntts_streams_*) on its engine, the engine's codec engine on the same lane.  Per turn of an engine:
        [snapshot + windows of its sets -> codec pass -> chunks out] [admit its next group: prompt pass] [next decode burst] -- the other
        engines' bursts run meanwhile.  Opt-in (`stream_on_gang = True`: it pays for the fp8 Nano-sized model, not for NeuTTS-Air bf16, see
        _infer_stream_batch_hip).  Group size `stream_admit`: by default ONE group per engine (the streams split evenly, at least 64
        per engine), all prompt passes at once and the chains side by side from then on -- 512 streams of the fp8 Nano-sized model over
        four 128-slot engines: 150.2 k codec-tokens/s, first audio 152 / 168 / 186 ms (first / median / last stream) against 144.2 k and
        182 ms on one 512-slot engine; smaller groups stagger the admissions (a group's first audio waits for ITS prompt pass only) but a
        decode chain next to another engine's prompt pass crawls: groups of 64 144.1 k, 157 / 177 / 203 ms; of 32 125.3 k
        (profiles/r05o_sweep_stream_gang.txt).  A window is defined by token COUNTS (30 undecoded tokens, csrc/stream.cpp pump_end), not
        by when it is looked for: the chunks are those of the single stream, bit for bit, whatever the burst boundaries."""
        gang = self.gang
        G = len(gang.engines)
        codecs = self._gang_codec_engines()
        n = len(prompts)
        self._seed += 1
        admit = getattr(self, "stream_admit", None)
        if not admit:
            groups = min(G, max(1, n // 64))
            admit = (n + groups - 1) // groups
        admit = max(1, int(admit))
        chunk, look_f = self.streaming_frames_per_chunk, self.streaming_lookforward
        mod = int(getattr(self, "_stream_modulo", 0))
        room = [e.free_slots() for e in gang.engines]
        if n > sum(room):
            raise ValueError(f"{n} utterances exceed the gang's free decode slots ({sum(room)})")
        # groups of at most `admit` streams, each to the engine with the most room left (ties: round-robin), a group no larger than that room
        pending: List[List[List[int]]] = [[] for _ in range(G)]
        i0, turn = 0, 0
        while i0 < n:
            k = max(range(G), key=lambda j: (room[j], -((j - turn) % G)))
            m = min(admit, room[k], n - i0)
            pending[k].append(list(range(i0, i0 + m)))
            room[k] -= m
            i0 += m
            turn += 1
        active: List[list] = [[] for _ in range(G)]           # per engine: [stream set, utterance indices, slots]
        try:
            while any(pending) or any(active):
                for k, eng in enumerate(gang.engines):
                    if not pending[k] and not active[k]:
                        continue
                    running = 0
                    for item in active[k]:
                        item[0].pump_begin()                  # behind the burst enqueued in this engine's previous turn
                    for item in active[k]:
                        running += item[0].pump_wait()
                    for item in list(active[k]):
                        ss, idx, slots = item[0], item[1], item[2]
                        more = True
                        while more:
                            chunks, _, more = ss.pump_end()   # codec pass + cross-fade on the engine's lane, ahead of its next burst
                            for i, samples, _last in chunks:
                                yield idx[i], np.array(samples)
                        if ss.done() == ss.n:
                            ss.close()
                            eng.release_many(slots)
                            item[2] = []
                            active[k].remove(item)
                    admitted = False
                    if pending[k]:
                        idx = pending[k].pop(0)
                        slots = [eng.acquire_slot() for _ in idx]
                        item = [None, idx, slots]
                        active[k].append(item)                # (registered first: its slots are released on any exit path)
                        budget = eng.cfg.get("max_prefill_tokens", 0) or 16384
                        i0 = 0
                        while i0 < len(idx):
                            i1, used = i0, 0
                            while i1 < len(idx) and (i1 == i0 or used + len(prompts[idx[i1]]) <= budget):
                                used += len(prompts[idx[i1]])
                                i1 += 1
                            eng.prefill([prompts[i] for i in idx[i0:i1]], slots[i0:i1], [self._sampling(len(prompts[i]), i) for i in idx[i0:i1]])
                            i0 = i1
                        item[0] = _hip.StreamSet(eng, codecs[k], slots, [ref_codes[i] for i in idx], int(eng.cfg["max_context"]), chunk, look_f,
                                                 self.streaming_lookback, self.streaming_overlap_frames, self.hop_length,
                                                 0 if mod else int(self._speech_base), mod or 65536, bool(mod))
                        admitted = True
                        running += len(idx)
                    if running:
                        # a new group holds one token per stream: lookforward - 1 steps now and whole chunks from then on complete its
                        # first window (chunk + lookforward tokens) exactly at a burst boundary; the older groups just run ahead.
                        # (Measured and not kept: bursts sized to whatever the set nearest to its next window needs -- more, shorter
                        #  bursts and pump rounds: 127.0 k instead of 143.5 k codec-tokens/s at 512 streams, first audio 182 instead of
                        #  157 ms; profiles/r05g_bench_nano-fp8_stream_*.json)
                        eng.decode(look_f - 1 if (admitted and look_f > 1) else chunk)
        finally:
            try:
                codecs[0].set_stream(None)               # the class's own codec engine goes back to its stream for the non-gang calls that follow
            except _hip.NeuTTSHipError:                  # (infer, infer_stream, infer_stream_batch on engine 0); re-lent by the next gang call
                pass
            self._gang_codec0_lent = False
            for k, eng in enumerate(gang.engines):
                for item in active[k]:
                    if item[0] is not None:
                        item[0].close()
                if any(item[2] for item in active[k]):
                    try:
                        eng.sync()
                    except _hip.NeuTTSHipError:
                        pass
                    for item in active[k]:
                        for sl in item[2]:
                            try:
                                eng.release(sl)
                            except _hip.NeuTTSHipError:
                                pass

    def _infer_stream_batch_hip(self, prompts: List[List[int]], ref_codes: List[List[int]]):
        # (stream_on_gang: off unless the caller sets it -- it pays for one of the two models measured.  512 streams of the fp8 Nano-sized model
        #  over four 128-slot engines, one admission group per engine: 150.2 k codec-tokens/s and first audio 152 / 168 / 186 ms against 144.2 k /
        #  182 ms on ONE 512-slot engine; NeuTTS-Air bf16: 256 streams over 4 x 64 slots 80.3 k against 100.5 k on one engine, 512 streams over
        #  4 x 128 111.9 k against 114.9 k -- four chains stream Air's larger weights four times; profiles/r05o_sweep_stream_gang*.txt)
        if (self.gang is not None and len(prompts) > 1 and getattr(self, "stream_on_gang", False) and self._stream_on_device(ref_codes)):
            yield from self._infer_stream_batch_gang(prompts, ref_codes)
            return
        eng = self.backbone
        n = len(prompts)
        self._seed += 1
        if eng.free_slots() < n:
            raise ValueError(f"{n} utterances exceed the engine's {eng.free_slots()} free decode slots")
        slots = [eng.acquire_slot() for _ in range(n)]
        budget = eng.cfg.get("max_prefill_tokens", 0) or 16384       # the prefill workspace bounds one call
        try:
            i0 = 0
            while i0 < n:
                i1, used = i0, 0
                while i1 < n and (i1 == i0 or used + len(prompts[i1]) <= budget):
                    used += len(prompts[i1])
                    i1 += 1
                eng.prefill(prompts[i0:i1], slots[i0:i1], [self._sampling(len(prompts[i]), i) for i in range(i0, i1)])
                i0 = i1
        except Exception:                  # a later group failed: the earlier groups' slots must not stay RUNNING
            eng.sync()
            for s in slots:
                eng.release(s)
            raise
        hop, stride = self.hop_length, self.streaming_stride_samples
        chunk, look_f, look_b, ovl = (self.streaming_frames_per_chunk, self.streaming_lookforward,
                                      self.streaming_lookback, self.streaming_overlap_frames)
        if self._stream_on_device(ref_codes):
            try:
                yield from self._stream_batch_device(slots, ref_codes)
            finally:
                eng.sync()
                for s in slots:
                    eng.release(s)
            return
        # ---- host path (a watermarker is a host library; an id -> code rule other than the speech-token range): per utterance the
        #      reference codes + generated codes in one int32 row (the reference caches "<|speech_N|>" strings)
        cap = max(len(rc) for rc in ref_codes) + int(eng.cfg["max_context"]) + 1
        cache = np.zeros((n, cap), dtype=np.int32)
        clen = np.zeros(n, dtype=np.int64)                # tokens in cache[i]
        for i, rc in enumerate(ref_codes):
            cache[i, :len(rc)] = rc
            clen[i] = len(rc)
        n_dec = clen.copy()                               # tokens already turned into audio
        n_seen = [0] * n
        blend = [_StreamBlender(stride) for _ in range(n)]
        done = [False] * n                                # final chunk emitted
        need = chunk + look_f                             # undecoded tokens that make a window decodable (ref :401-404)
        try:
            while not all(done):
                ids_all, n_new_all, fin_all = eng.read_all_array()    # blocks until the bursts enqueued so far are done
                ids, n_new, fin = ids_all[slots], n_new_all[slots], fin_all[slots]     # row i = utterance i
                running = [i for i in range(n) if not done[i] and not fin[i]]
                if running and self.streaming_overlap_compute:
                    eng.decode(chunk)                     # async: runs beside the codec pass below
                jobs = []                                 # (utterance, window codes, s0, s1 or None, last)
                for i in range(n):
                    if done[i]:
                        continue
                    if n_new[i] > n_seen[i]:
                        new = self._ids_to_codes_array(ids[i, n_seen[i]:n_new[i]])
                        cache[i, clen[i]:clen[i] + len(new)] = new
                        clen[i] += len(new)
                        n_seen[i] = int(n_new[i])
                    # The reference checks after EVERY appended token and slices `token_cache[start : n_dec + 31]` at the moment the
                    # 30th undecoded token arrives -- the slice then ends at n_dec + 30, whatever arrives later.  Same windows here
                    # without the per-token loop: as many as the tokens of this burst complete, each ending at its own n_dec + 30.
                    while clen[i] - n_dec[i] >= need:
                        t0 = max(int(n_dec[i]) - look_b - ovl, 0)
                        t1 = int(n_dec[i]) + need
                        s0 = (int(n_dec[i]) - t0) * hop
                        jobs.append((i, cache[i, t0:t1], s0, s0 + (chunk + 2 * ovl) * hop, False))
                        n_dec[i] += chunk
                    if fin[i]:
                        remaining = int(clen[i] - n_dec[i])
                        if remaining > 0:
                            t0 = max(int(clen[i]) - (look_b + ovl + remaining), 0)
                            s0 = (int(clen[i]) - t0 - remaining - ovl) * hop
                            jobs.append((i, cache[i, t0:clen[i]], s0, None, True))
                        done[i] = True
                if jobs:
                    wavs = self.codec.engine.decode([j[1] for j in jobs], reuse_output=True)     # one batched codec pass; views of the
                    pieces = []                                                                   # engine's pinned buffer: EVERY job's slice is
                    for (i, _, s0, s1, last), recon in zip(jobs, wavs):                           # copied out before the first yield -- while the
                        if self.watermarker is not None:                                         # generator is suspended another decode on this
                            recon = self.watermarker.apply_watermark(recon, sample_rate=24_000)   # codec engine may overwrite that buffer
                        pieces.append((i, np.array(recon[s0:s1]), last))
                    del wavs
                    for i, piece, last in pieces:
                        yield i, blend[i].push(piece, last=last)
                if running and not self.streaming_overlap_compute:
                    eng.decode(chunk)
        finally:
            eng.sync()
            for s in slots:
                eng.release(s)


def _engine_config_from_hf(hc) -> Dict[str, object]:
    """AutoModelForCausalLM's dispatch (ref:neutts/neutts.py:164), restated for the decoder family the engine implements:
    the pre-norm RoPE / GQA / SwiGLU decoder of Qwen2 (NeuTTS-Air), Llama-style checkpoints (no q/k/v bias, possibly an
    untied head) and Qwen3 (per-head q/k RMSNorm, head_dim 128: the engine's general attention path, round 6).  Everything is read from config.json; what the kernels cannot do fails HERE with the reason, not later
    with wrong audio."""
    mt = getattr(hc, "model_type", "")
    if mt not in ("qwen2", "llama", "mistral", "qwen3"):
        raise NotImplementedError(f"backbone model_type {mt!r}: the MI355X engine implements the Qwen2 / Llama decoder "
                                  "family (NeuTTS-Air is 'qwen2')")
    head_dim = getattr(hc, "head_dim", None) or hc.hidden_size // hc.num_attention_heads
    if head_dim not in (64, 128):
        raise NotImplementedError(f"backbone head_dim {head_dim}: the attention kernels are built for head_dim 64 and 128")
    if getattr(hc, "sliding_window", None) and getattr(hc, "use_sliding_window", False):
        raise NotImplementedError("sliding-window attention is not implemented")
    if getattr(hc, "mlp_bias", False):
        raise NotImplementedError("MLP biases are not implemented")
    return dict(vocab_size=hc.vocab_size, hidden_size=hc.hidden_size, intermediate_size=hc.intermediate_size,
                num_layers=hc.num_hidden_layers, num_heads=hc.num_attention_heads,
                num_kv_heads=getattr(hc, "num_key_value_heads", None) or hc.num_attention_heads, rms_eps=hc.rms_norm_eps,
                head_dim=head_dim, tie_word_embeddings=bool(getattr(hc, "tie_word_embeddings", True)),
                attention_bias=bool(getattr(hc, "attention_bias", mt == "qwen2")),
                qk_norm=(mt == "qwen3"))       # Qwen3Attention: q_norm / k_norm (RMSNorm over head_dim) on every head before RoPE


def _to_list(codes) -> List[int]:
    if hasattr(codes, "tolist"):
        return [int(c) for c in np.asarray(codes.cpu() if hasattr(codes, "cpu") else codes).reshape(-1).tolist()]
    return [int(c) for c in codes]


def _safetensors_shards(repo) -> List[str]:
    """*.safetensors files of a checkpoint: a local directory, or a hub repo id resolved through the local HF cache /
    a download of exactly those files."""
    import glob
    import os
    repo = str(repo)
    if os.path.isdir(repo):
        return sorted(glob.glob(os.path.join(repo, "*.safetensors")))
    try:
        from huggingface_hub import snapshot_download
        d = snapshot_download(repo, allow_patterns=["*.safetensors", "*.json"])
        return sorted(glob.glob(os.path.join(d, "*.safetensors")))
    except Exception:
        return []


def _default_rope_theta(hc) -> Optional[float]:
    """rope_theta when the checkpoint uses the default RoPE parametrisation (what NeuTTS-Air / Qwen2.5 do), else None
    (the caller then lets transformers build the rotary embedding and reads its inv_freq buffer)."""
    rp = getattr(hc, "rope_parameters", None) or getattr(hc, "rope_scaling", None)
    if isinstance(rp, dict):
        if rp.get("rope_type", rp.get("type", "default")) != "default":
            return None
        if "rope_theta" in rp:
            return float(rp["rope_theta"])
    theta = getattr(hc, "rope_theta", None)
    return float(theta) if theta is not None else None


class _SafetensorsStateDict:
    """Minimal `.items()` view over safetensors shards: one tensor in host memory at a time."""

    def __init__(self, shards: Sequence[str]):
        self.shards = list(shards)

    def items(self):
        from safetensors import safe_open
        for path in self.shards:
            with safe_open(path, framework="pt", device="cpu") as f:
                for k in f.keys():
                    if not k.endswith("inv_freq"):
                        yield k, f.get_tensor(k)


def neucodec_to_xcodec2_names(sd: Dict[str, object], strict: bool = True) -> Dict[str, object]:
    """Original `neucodec` state-dict keys -> the xcodec2 parameter names the codec engine loads.
    Fused `att.c_attn.weight [3H, H]` rows are split q | k | v.

    The key layout is the survey's reading of the neucodec package, which cannot be installed offline, so this function
    does not trust it: with `strict` (the default) EVERY decoder-side tensor the engine needs must be found under its
    expected source key, and every decoder-side source key (`generator.backbone.*`, `generator.head.*`,
    `generator.quantizer.project_out.*`, `fc_post_a.*`) must be consumed -- otherwise a ValueError lists the missing and
    the unrecognised keys, instead of a late 'tensor X not loaded' from the engine or, worse, silently skipped weights."""
    out: Dict[str, object] = {}
    missing: List[str] = []
    used = set()

    def put(dst, src):
        if src in sd:
            out[dst] = sd[src]
            used.add(src)
        else:
            missing.append(f"{src} (-> {dst})")

    put("quantizer.project_out.weight", "generator.quantizer.project_out.weight")
    put("quantizer.project_out.bias", "generator.quantizer.project_out.bias")
    put("decoder.fc.weight", "fc_post_a.weight")
    put("decoder.fc.bias", "fc_post_a.bias")
    put("decoder.embed.weight", "generator.backbone.embed.weight")
    put("decoder.embed.bias", "generator.backbone.embed.bias")
    for net in ("prior_net", "post_net"):
        for b in range(2):
            for part in ("norm1", "conv1", "norm2", "conv2"):
                for wb in ("weight", "bias"):
                    put(f"decoder.{net}.{b}.{part}.{wb}", f"generator.backbone.{net}.{b}.{part}.{wb}")
    i = 0
    while f"generator.backbone.transformers.{i}.att.c_attn.weight" in sd:
        p, q = f"generator.backbone.transformers.{i}.", f"decoder.layers.{i}."
        w = sd[p + "att.c_attn.weight"]
        used.add(p + "att.c_attn.weight")
        if w.shape[0] % 3:
            raise ValueError(f"{p}att.c_attn.weight: {tuple(w.shape)} is not a fused q|k|v matrix")
        h = w.shape[0] // 3
        out[q + "self_attn.q_proj.weight"], out[q + "self_attn.k_proj.weight"], out[q + "self_attn.v_proj.weight"] = (
            w[:h], w[h:2 * h], w[2 * h:])
        put(q + "self_attn.o_proj.weight", p + "att.c_proj.weight")
        put(q + "input_layernorm.weight", p + "att_norm.weight")
        put(q + "post_attention_layernorm.weight", p + "ffn_norm.weight")
        put(q + "mlp.fc1.weight", p + "mlp.fc1.weight")
        put(q + "mlp.fc2.weight", p + "mlp.fc2.weight")
        i += 1
    if i == 0:
        missing.append("generator.backbone.transformers.0.att.c_attn.weight (-> decoder.layers.0.self_attn.{q,k,v}_proj.weight)")
    put("decoder.norm.weight", "generator.backbone.final_layer_norm.weight")
    put("decoder.norm.bias", "generator.backbone.final_layer_norm.bias")
    put("decoder.head.linear.weight", "generator.head.out.weight")
    put("decoder.head.linear.bias", "generator.head.out.bias")
    if not out:
        raise ValueError("no NeuCodec decoder tensors found in the state dict")
    # decoder-side source keys nobody asked for: buffers that carry no parameters are fine, anything else is a layout
    # this mapping does not know
    benign = ("generator.head.istft.window", "rotary", "inv_freq", "freqs_cis", "num_batches_tracked", "codebook",
              "implicit_codebook", "_levels", "_basis", "scales", "project_in")
    decoder_side = ("generator.backbone.", "generator.head.", "generator.quantizer.project_out", "fc_post_a.")
    unknown = sorted(k for k in sd if k.startswith(decoder_side) and k not in used and not any(b in k for b in benign))
    if strict and (missing or unknown):
        raise ValueError(
            "NeuCodec checkpoint does not match the expected decoder layout (SURVEY.md B.4):\n"
            + (f"  missing source keys ({len(missing)}): " + ", ".join(missing[:12]) + (" ..." if len(missing) > 12 else "") + "\n"
               if missing else "")
            + (f"  unrecognised decoder-side keys ({len(unknown)}): " + ", ".join(unknown[:12]) + (" ..." if len(unknown) > 12 else "")
               if unknown else ""))
    return out


def neucodec_encoder_to_xcodec2_names(sd: Dict[str, object], n_layers: int = 16) -> Dict[str, object]:
    """Original `neucodec` ENCODER-side state-dict keys -> the xcodec2 parameter names the encoder engine loads.

    Like the decoder mapping above, the key layout is a reading of the neucodec / xcodec2 sources that cannot be checked
    offline, so nothing is trusted: the w2v-BERT, adapter, fc and project_in tensors are looked up by name and every one
    must exist; the acoustic encoder (`CodecEnc.*`, weight-normalised convolutions + SnakeBeta activations inside nested
    nn.Sequential containers) is mapped by MODULE ORDER and checked by SHAPE -- 1 input conv, per ratio 3 x (snake, conv 7,
    snake, conv 1) + snake + strided conv, then snake + conv 3 -- with weight norm folded (w = g * v / ||v||).  Any mismatch
    raises ValueError that lists the keys (there is no other encoder to fall back to)."""
    import torch
    out: Dict[str, object] = {}
    missing: List[str] = []

    def put(dst, src):
        if src in sd:
            out[dst] = sd[src]
        else:
            missing.append(f"{src} (-> {dst})")

    for k in sd:
        if k.startswith("semantic_model.") and "masked_spec_embed" not in k:
            parts = k.split(".")
            if parts[1:3] == ["encoder", "layers"] and int(parts[3]) >= n_layers:
                continue                                   # neucodec reads hidden_states[16]: layers 17..24 never run
            if parts[1] in ("feature_projection", "encoder"):
                out["semantic_encoder." + k[len("semantic_model."):]] = sd[k]
    if not any(k.startswith("semantic_encoder.encoder.layers.") for k in out):
        missing.append("semantic_model.encoder.layers.* (w2v-BERT 2.0)")
    put("semantic_adapter.conv1.weight", "SemanticEncoder_module.initial_conv.weight")
    put("semantic_adapter.conv2.weight", "SemanticEncoder_module.residual_blocks.1.weight")
    put("semantic_adapter.conv2.bias", "SemanticEncoder_module.residual_blocks.1.bias")
    put("semantic_adapter.conv3.weight", "SemanticEncoder_module.residual_blocks.3.weight")
    put("semantic_adapter.conv3.bias", "SemanticEncoder_module.residual_blocks.3.bias")
    put("semantic_adapter.conv4.weight", "SemanticEncoder_module.final_conv.weight")
    put("fc_encoder.weight", "fc_prior.weight")
    put("fc_encoder.bias", "fc_prior.bias")
    put("quantizer.project_in.weight", "generator.quantizer.project_in.weight")
    put("quantizer.project_in.bias", "generator.quantizer.project_in.bias")
    # acoustic encoder: modules in registration order
    mods: Dict[str, Dict[str, object]] = {}
    for k, v in sd.items():
        if not k.startswith("CodecEnc.") or "filter" in k:
            continue
        for suf in ("parametrizations.weight.original0", "parametrizations.weight.original1", "weight_g", "weight_v", "weight",
                    "bias", "alpha", "beta"):
            if k.endswith("." + suf):
                mods.setdefault(k[: -len(suf) - 1], {})[suf] = v
                break
        else:
            missing.append(f"{k} (unrecognised acoustic-encoder tensor)")
    convs, snakes = [], []
    for name, t in mods.items():
        if "alpha" in t and "beta" in t:
            snakes.append((name, t["alpha"].reshape(-1), t["beta"].reshape(-1)))
            continue
        g = t.get("weight_g", t.get("parametrizations.weight.original0"))
        v = t.get("weight_v", t.get("parametrizations.weight.original1"))
        if g is not None and v is not None:
            w = v * (g.reshape(-1, 1, 1) / torch.linalg.vector_norm(v.float(), dim=(1, 2), keepdim=True).to(v.dtype))
        elif "weight" in t:
            w = t["weight"]
        else:
            missing.append(f"{name} (neither a convolution nor a SnakeBeta)")
            continue
        convs.append((name, w, t.get("bias")))
    n_ratios = (len(convs) - 2) // 7
    if missing or len(convs) != 2 + 7 * n_ratios or len(snakes) != 1 + 7 * n_ratios or n_ratios < 1:
        raise ValueError("NeuCodec checkpoint does not match the expected ENCODER layout: "
                         + (", ".join(missing[:12]) + (" ..." if len(missing) > 12 else "") if missing
                            else f"{len(convs)} convolutions / {len(snakes)} SnakeBeta activations under CodecEnc.*"))
    ci, si = iter(convs), iter(snakes)

    def conv(dst, k=None, cin=None):
        name, w, b = next(ci)
        if w.dim() != 3 or (k is not None and w.shape[2] != k) or (cin is not None and w.shape[1] != cin) or b is None:
            raise ValueError(f"acoustic encoder: {name} {tuple(w.shape)} is not the convolution expected at {dst}")
        out[dst + ".weight"], out[dst + ".bias"] = w, b
        return w.shape[0]

    def snake(dst, c):
        name, a, b = next(si)
        if a.numel() != c or b.numel() != c:
            raise ValueError(f"acoustic encoder: {name} has {a.numel()} channels, {c} expected at {dst}")
        out[dst + ".act.alpha"], out[dst + ".act.beta"] = a, b

    ch = conv("acoustic_encoder.conv1", 7, 1)
    for bi in range(n_ratios):
        b = f"acoustic_encoder.block.{bi}."
        for u in (1, 2, 3):
            snake(f"{b}res_unit{u}.snake1", ch)
            conv(f"{b}res_unit{u}.conv1", 7, ch)
            snake(f"{b}res_unit{u}.snake2", ch)
            conv(f"{b}res_unit{u}.conv2", 1, ch)
        snake(b + "snake1", ch)
        ch = conv(b + "conv1", None, ch)
    snake("acoustic_encoder.snake1", ch)
    conv("acoustic_encoder.conv2", 3, ch)
    return out

