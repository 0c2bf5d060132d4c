"""Synthetic workload generators: model geometries, seeded random weights and prompts.

Neither product nor oracle: plain data for `bench.py`, the tools, the tests and the golden-vector generators
(there is no network for checkpoints or datasets, so the benchmark runs random-init weights of the exact
NeuTTS-Air / NeuCodec shapes and `randint` prompts; numpy PCG64 so that every machine draws identical values).
`oracle/backbone_ref.py` and `oracle/codec_ref.py` re-export these names; nothing here computes a model.
"""
from __future__ import annotations

import re
from dataclasses import asdict, dataclass
from typing import Dict, List

import numpy as np
import torch


@dataclass(frozen=True)
class BackboneConfig:
    vocab_size: int = 217488          # 151 936 base + 65 536 speech + specials
    hidden_size: int = 896
    intermediate_size: int = 4864
    num_layers: int = 24
    num_heads: int = 14
    num_kv_heads: int = 2
    head_dim: int = 64
    rms_eps: float = 1e-6
    rope_theta: float = 1e6
    attention_bias: bool = True       # Qwen2: q/k/v_proj carry a bias; Llama-style decoders do not
    tie_word_embeddings: bool = True  # False: a separate lm_head.weight
    qk_norm: bool = False             # Qwen3-style: RMSNorm over head_dim on every q and k head before RoPE (self_attn.q_norm / k_norm weights)

    @staticmethod
    def neutts_air(vocab_size: int = 217488) -> "BackboneConfig":
        return BackboneConfig(vocab_size=vocab_size)

    @staticmethod
    def neutts_nano_like(vocab_size: int = 142080) -> "BackboneConfig":
        """A geometry with NeuTTS-Nano's published parameter counts (ref:README.md:44-45: ~120 M active, ~229 M with the
        embedding).  The checkpoint's config.json cannot be fetched offline, so this is an ASSUMED shape, chosen to satisfy
        both numbers with the engine's head_dim of 64: hidden 768, 19 layers, 12 query / 4 kv heads, FFN 2048
        (19 x 6.29 M = 119.6 M) and 142 080 x 768 = 109.1 M tied embedding weights."""
        return BackboneConfig(vocab_size=vocab_size, hidden_size=768, intermediate_size=2048, num_layers=19, num_heads=12,
                              num_kv_heads=4, head_dim=64)

    @staticmethod
    def qwen3_like(vocab_size: int = 8192, num_layers: int = 2) -> "BackboneConfig":
        """Qwen3-0.6B's attention geometry (hidden 1024, 16 query / 8 kv heads of head_dim 128 -- q width 2048 != hidden --, per-head q/k
        RMSNorm, no attention bias, FFN 3072) at a depth and vocabulary the CPU oracle finishes quickly: what the reference's
        AutoModelForCausalLM dispatch (ref:neutts/neutts.py:164) would hand the engine for a Qwen3-based checkpoint."""
        return BackboneConfig(vocab_size=vocab_size, hidden_size=1024, intermediate_size=3072, num_layers=num_layers, num_heads=16,
                              num_kv_heads=8, head_dim=128, attention_bias=False, qk_norm=True)

    @staticmethod
    def tiny(vocab_size: int = 1024, num_layers: int = 2) -> "BackboneConfig":
        """Same head geometry (GQA 7:1, d=64) at a size the oracle finishes in milliseconds."""
        return BackboneConfig(vocab_size=vocab_size, hidden_size=448, intermediate_size=1216,
                              num_layers=num_layers, num_heads=7, num_kv_heads=1, head_dim=64)

    def to_dict(self):
        return asdict(self)


# --------------------------------------------------------------------------------------
# deterministic synthetic weights (numpy PCG64: stable across machines / torch versions)
# --------------------------------------------------------------------------------------
def make_weights(cfg: BackboneConfig, seed: int = 0, init: str = "unit",
                 peak_sigma: float = 0.0, walk_gain: float = 0.0, walk_scale: float = 4.0, walk_range=None) -> Dict[str, torch.Tensor]:
    """HF-named fp32 state dict (hf:models/qwen2/modeling_qwen2.py: names of Qwen2ForCausalLM).

    init="hf":   N(0, 0.02) matrices like HF's `_init_weights`.  With tied
                 embeddings such a model mostly re-predicts its input token -- a weak id test.
    init="unit": N(0, 1/fan_in) matrices (unit gain), embedding N(0, 0.02): layer outputs
                 dominate the residual stream, greedy ids are diverse.  Used for parity + bench.
    peak_sigma:  multiply embedding row j by exp(N(0, peak_sigma)): heavy-tailed logits whose
                 top-1/top-2 gap is many bf16 ulps -> free-running greedy ids are comparable
                 bit-for-bit across implementations with different fp32 summation order.
    walk_gain:   > 0: greedy decoding walks a seeded permutation of the vocabulary -- or of the ids in walk_range = (lo, hi), e.g. the
                 speech tokens of a TTS tokenizer -- with wide margins (_make_walk below).
    Norm weights are 1 + N(0, 0.1) and biases N(0, 0.02) so every multiply/add is exercised.
    lm_head is tied to embed_tokens (hf:...modeling_qwen2.py:407 `_tied_weights_keys`).
    """
    rng = np.random.default_rng(seed)

    def normal(*shape, s=0.02):
        return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * np.float32(s))

    def mat(n_out, n_in):
        return normal(n_out, n_in, s=0.02 if init == "hf" else float(n_in) ** -0.5)

    H, F_, nh, nkv, d = cfg.hidden_size, cfg.intermediate_size, cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
    w: Dict[str, torch.Tensor] = {}
    emb = normal(cfg.vocab_size, H)
    if peak_sigma > 0:
        scale = np.exp(rng.standard_normal(cfg.vocab_size, dtype=np.float32) * np.float32(peak_sigma))
        emb = emb * torch.from_numpy(scale)[:, None]
    w["model.embed_tokens.weight"] = emb
    for i in range(cfg.num_layers):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = 1.0 + normal(H, s=0.1)
        for name, rows in (("q_proj", nh * d), ("k_proj", nkv * d), ("v_proj", nkv * d)):   # (draw order is part of the fixtures)
            w[p + f"self_attn.{name}.weight"] = mat(rows, H)
            if cfg.attention_bias:
                w[p + f"self_attn.{name}.bias"] = normal(rows)
        if cfg.qk_norm:                                         # (drawn only for qk_norm models: the existing fixtures' draw order is untouched)
            w[p + "self_attn.q_norm.weight"] = 1.0 + normal(d, s=0.1)
            w[p + "self_attn.k_norm.weight"] = 1.0 + normal(d, s=0.1)
        w[p + "self_attn.o_proj.weight"] = mat(H, nh * d)
        w[p + "post_attention_layernorm.weight"] = 1.0 + normal(H, s=0.1)
        w[p + "mlp.gate_proj.weight"] = mat(F_, H)
        w[p + "mlp.up_proj.weight"] = mat(F_, H)
        w[p + "mlp.down_proj.weight"] = mat(H, F_)
    w["model.norm.weight"] = 1.0 + normal(H, s=0.1)
    if not cfg.tie_word_embeddings:
        head = normal(cfg.vocab_size, H)
        if peak_sigma > 0:
            head = head * torch.from_numpy(np.exp(rng.standard_normal(cfg.vocab_size, dtype=np.float32) * np.float32(peak_sigma)))[:, None]
        w["lm_head.weight"] = head
    if walk_gain > 0:
        _make_walk(cfg, w, seed, walk_scale, walk_gain, walk_range)
    return w


def walk_permutation(cfg: BackboneConfig, seed: int, walk_range=None) -> np.ndarray:
    """The successor table of make_weights(..., walk_gain > 0): nxt[j] = the token a greedy step emits after token j when nothing but
    the construction speaks: ONE cycle through the ids of walk_range = (lo, hi) -- default: every id except the last one, which the
    fixtures use as EOS; every id outside the range is its own successor."""
    V = cfg.vocab_size
    lo, hi = walk_range if walk_range is not None else (0, V - 1)
    order = lo + np.random.default_rng(seed + 7919).permutation(hi - lo)
    nxt = np.arange(V, dtype=np.int64)
    nxt[order] = np.roll(order, -1)
    return nxt


def _make_walk(cfg: BackboneConfig, w: Dict[str, torch.Tensor], seed: int, scale: float, gain: float, walk_range=None) -> None:
    """Re-shape the TIED embedding and the LAST layer's MLP of a unit-gain model so that greedy decoding walks a seeded permutation of
    the vocabulary with top-1 / top-2 margins of tens of per cent of the top logit (hundreds of bf16 ulps) -- a free-running fixture
    whose 250 ids are all DIFFERENT and cannot hinge on fp32 summation order (random-init logits are flat, 111 of
    250 steps are near-ties; the `peak_sigma` fixtures collapse into fixed points).  Everything else stays what make_weights drew:
    24 layers of attention and MLPs run on every token and perturb the logits; they just cannot overturn the construction.

      hidden dims: A = [0, Hc)   B = [Hc, 2 Hc)   const = H - 1            (Hc = (H - 2) // 2)
      embedding (tied head): E[k] = scale * [ c_k | c_{prev(k)} | 0 | kappa ]     c_j ~ N(0, 1)^Hc, prev = inverse permutation
      last layer MLP, features r < Hc: gate_r = e_const, up_r = e_{A,r}, down[B_r][r] = beta; every other entry of those rows / that
        column is zero.  With x = RMSNorm(h): silu(x_const) ~ x_const (it is ~ kappa >= 8), so the layer adds beta x_const x_A to the
        B half of the residual stream: a copy of the CURRENT token's code c_j, `gain` times larger than the embedding's own entries.
      logits: <norm(h), E[k]> ~ [ scale c_j . c_k ] + [ gain scale c_j . c_{prev(k)} ] + const: the second term peaks at
        prev(k) = j, i.e. k = next(j), with Hc * gain against Hc for the token itself and ~ 4.5 sqrt(Hc) for any other id.
    The features r >= Hc of the last MLP keep their random weights (the layer's arithmetic stays exercised)."""
    V, H, F_ = cfg.vocab_size, cfg.hidden_size, cfg.intermediate_size
    Hc = (H - 2) // 2
    if F_ < Hc or Hc < 64:
        raise ValueError("walk weights need hidden >= 130 and intermediate >= hidden / 2")
    kappa = 8.0
    rng = np.random.default_rng(seed + 104729)
    codes = rng.standard_normal((V, Hc), dtype=np.float32)
    nxt = walk_permutation(cfg, seed, walk_range)
    prev = np.empty(V, dtype=np.int64)
    prev[nxt] = np.arange(V)
    emb = np.zeros((V, H), dtype=np.float32)
    emb[:, :Hc] = codes
    emb[:, Hc:2 * Hc] = codes[prev]
    emb[:, H - 1] = kappa
    emb *= np.float32(scale)
    if not cfg.tie_word_embeddings:
        # an untied head: the two matrices take DIFFERENT halves of the construction (reading the logits off the embedding instead of
        # the head would make every step repeat its input token): embedding [ c_k | noise | 0 | kappa ], head [ noise | c_prev(k) | 0 | 0 ]
        head = emb.copy()
        head[:, :Hc] = rng.standard_normal((V, Hc), dtype=np.float32) * np.float32(0.25 * scale)
        head[:, H - 1] = 0.0
        emb[:, Hc:2 * Hc] = rng.standard_normal((V, Hc), dtype=np.float32) * np.float32(0.25 * scale)
        w["lm_head.weight"] = torch.from_numpy(head)
    w["model.embed_tokens.weight"] = torch.from_numpy(emb)
    p = f"model.layers.{cfg.num_layers - 1}.mlp."
    gate, up, down = w[p + "gate_proj.weight"], w[p + "up_proj.weight"], w[p + "down_proj.weight"]
    gate[:Hc] = 0.0
    gate[:Hc, H - 1] = 1.0
    up[:Hc] = 0.0
    up[torch.arange(Hc), torch.arange(Hc)] = 1.0
    down[:, :Hc] = 0.0
    # x_const ~ kappa * scale / rms(h) and rms(h) ~ scale * sqrt((2 Hc + kappa^2) / H) when the embedding dominates the stream:
    # beta * x_const * x_A = gain * scale * c  (x_A ~ c / that same ratio)  =>  beta = gain * scale * ratio^2 / kappa
    ratio2 = (2.0 * Hc + kappa * kappa) / H
    down[Hc + torch.arange(Hc), torch.arange(Hc)] = float(gain * scale * ratio2 / kappa)


# --------------------------------------------------------------------------------------
# fp8 model variant (static activation scales): plain data, like the weights
# --------------------------------------------------------------------------------------
FP8_MAX = 448.0
FP8_LINEARS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")


def fp8_input_scale_names(cfg: BackboneConfig) -> List[str]:
    names = ["lm_head.input_scale"]
    for i in range(cfg.num_layers):
        names += [f"model.layers.{i}.{t}.input_scale" for t in ("self_attn.q_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.down_proj")]
    return names


def default_fp8_input_scales(cfg: BackboneConfig, norm_out: float = 2.0 ** -5, attn_out: float = 2.0 ** -8,
                             mlp_act: float = 2.0 ** -6) -> Dict[str, float]:
    """Static per-tensor activation scales for the SYNTHETIC weights of make_weights(init="unit") (a real static-fp8
    checkpoint ships calibrated `input_scale` tensors).  e4m3 is a floating-point format -- 3 mantissa bits over
    [2^-6, 448] x scale -- so a power-of-two scale only has to place the bulk of the distribution inside that window:
    RMSNorm outputs are O(1) (window 4.9e-4 .. 14), attention outputs of these weights O(0.1) (6e-5 .. 1.75), the
    SiLU(gate) * up products O(0.1 .. 1) (2.4e-4 .. 7).  tests/test_oracle_fp8.py checks the clipped fraction."""
    out = {"lm_head.input_scale": norm_out}
    for i in range(cfg.num_layers):
        p = f"model.layers.{i}."
        out[p + "self_attn.q_proj.input_scale"] = norm_out
        out[p + "self_attn.o_proj.input_scale"] = attn_out
        out[p + "mlp.gate_proj.input_scale"] = norm_out
        out[p + "mlp.down_proj.input_scale"] = mlp_act
    return out


def cast_weights(w: Dict[str, torch.Tensor], dtype: torch.dtype) -> Dict[str, torch.Tensor]:
    return {k: v.to(dtype) for k, v in w.items()}


def rope_inv_freq(cfg: BackboneConfig) -> torch.Tensor:
    """compute_default_rope_parameters  hf:models/qwen2/modeling_qwen2.py:70-89."""
    d = cfg.head_dim
    return 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float) / d))


def synthetic_prompt(cfg: BackboneConfig, utt_idx: int, length: int = 500) -> List[int]:
    """randint(0, V) with seed 1234 + utt_idx (numpy PCG64 here so that the
    GPU box, which has no /root/reference and maybe another torch, draws identical prompts)."""
    rng = np.random.default_rng(1234 + utt_idx)
    return rng.integers(0, cfg.vocab_size, size=length, dtype=np.int64).tolist()


@dataclass(frozen=True)
class CodecConfig:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_layers: int = 12
    num_heads: int = 16
    head_dim: int = 64
    quantization_dim: int = 2048
    levels: tuple = (4, 4, 4, 4, 4, 4, 4, 4)
    hop_length: int = 480           # ref:neutts/neutts.py:86
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0

    @property
    def n_fft(self) -> int:         # hf:models/xcodec2/configuration_xcodec2.py:113-115
        return self.hop_length * 4

    @staticmethod
    def neucodec() -> "CodecConfig":
        return CodecConfig()

    @staticmethod
    def tiny() -> "CodecConfig":
        """Same structure at a size the CPU (and the SIMT emulator) handles in seconds."""
        return CodecConfig(hidden_size=128, intermediate_size=256, num_layers=2, num_heads=2, head_dim=64,
                           quantization_dim=256, levels=(4, 4, 4, 4), hop_length=24)

    def to_dict(self):
        d = asdict(self)
        d["levels"] = list(self.levels)
        return d


def make_codec_weights(cfg: CodecConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """fp32 state dict with the parameter names of transformers' Xcodec2Model (quantizer.project_out + decoder.*).
    Unit-gain matrices (N(0, 1/fan_in)), perturbed norm weights/biases so every affine term is exercised; the
    ISTFT head is scaled down so exp(magnitude) stays far from the clamp(max=100) except for a few bins."""
    rng = np.random.default_rng(seed)
    H, I = cfg.hidden_size, cfg.intermediate_size

    def n(*shape, s):
        return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * np.float32(s))

    w: Dict[str, torch.Tensor] = {}
    nq = len(cfg.levels)
    w["quantizer.project_out.weight"] = n(cfg.quantization_dim, nq, s=nq ** -0.5)
    w["quantizer.project_out.bias"] = n(cfg.quantization_dim, s=0.1)
    w["decoder.fc.weight"] = n(H, cfg.quantization_dim, s=cfg.quantization_dim ** -0.5)
    w["decoder.fc.bias"] = n(H, s=0.1)
    w["decoder.embed.weight"] = n(H, H, 7, s=(7 * H) ** -0.5)
    w["decoder.embed.bias"] = n(H, s=0.1)
    for net in ("prior_net", "post_net"):
        for b in range(2):
            p = f"decoder.{net}.{b}."
            for k in (1, 2):
                w[p + f"norm{k}.weight"] = 1.0 + n(H, s=0.1)
                w[p + f"norm{k}.bias"] = n(H, s=0.1)
                w[p + f"conv{k}.weight"] = n(H, H, 3, s=(3 * H) ** -0.5)
                w[p + f"conv{k}.bias"] = n(H, s=0.1)
    for i in range(cfg.num_layers):
        p = f"decoder.layers.{i}."
        w[p + "input_layernorm.weight"] = 1.0 + n(H, s=0.1)
        for proj in ("q_proj", "k_proj", "v_proj", "o_proj"):
            w[p + f"self_attn.{proj}.weight"] = n(H, H, s=H ** -0.5)
        w[p + "post_attention_layernorm.weight"] = 1.0 + n(H, s=0.1)
        w[p + "mlp.fc1.weight"] = n(I, H, s=H ** -0.5)
        w[p + "mlp.fc2.weight"] = n(H, I, s=I ** -0.5)
    w["decoder.norm.weight"] = 1.0 + n(H, s=0.1)
    w["decoder.norm.bias"] = n(H, s=0.1)
    w["decoder.head.linear.weight"] = n(cfg.n_fft + 2, H, s=0.5 * H ** -0.5)
    w["decoder.head.linear.bias"] = n(cfg.n_fft + 2, s=0.1)
    return w


# ------------------------------------------------------------------------------------------------------------------
# NeuCodec ENCODER (reference enrolment, ref:neutts/neutts.py:266-271): geometry + seeded synthetic weights
# ------------------------------------------------------------------------------------------------------------------
@dataclass
class EncoderConfig:
    """Geometry of `NeuCodec.encode_code`: w2v-BERT 2.0 conformer (first 16 layers) + semantic adapter + acoustic conv
    encoder + fc + FSQ.  Parameter names / defaults are transformers' Xcodec2Config + Wav2Vec2BertConfig
    (hf:models/xcodec2/configuration_xcodec2.py:58-86)."""
    sem_hidden: int = 1024
    sem_layers: int = 16
    sem_heads: int = 16
    sem_ffn: int = 4096
    sem_feat_dim: int = 160                 # 80 mel bins x stride 2
    sem_conv_kernel: int = 31
    sem_left: int = 64                      # relative_key distance clamp
    sem_right: int = 8
    sem_ln_eps: float = 1e-5
    ac_hidden: int = 48                     # encoder_hidden_size
    ratios: tuple = (2, 2, 4, 4, 5)         # downsampling_ratios: 16 kHz -> 50 Hz
    codec_hidden: int = 1024                # acoustic encoder output channels
    levels: tuple = (4, 4, 4, 4, 4, 4, 4, 4)
    sample_rate: int = 16000

    @property
    def hop(self) -> int:
        return int(np.prod(self.ratios))

    @property
    def cat_dim(self) -> int:               # fc_encoder / quantizer.project_in width
        return self.codec_hidden + self.sem_hidden

    @staticmethod
    def neucodec() -> "EncoderConfig":
        return EncoderConfig()

    @staticmethod
    def tiny() -> "EncoderConfig":
        """Same structure (every ratio, 2 conformer layers, head size 16) at a size the SIMT emulator handles."""
        return EncoderConfig(sem_hidden=64, sem_layers=2, sem_heads=4, sem_ffn=128, ac_hidden=4, codec_hidden=64)

    def to_dict(self):
        d = asdict(self)
        d["ratios"] = list(self.ratios)
        d["levels"] = list(self.levels)
        return d


def make_encoder_weights(cfg: EncoderConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """fp32 state dict with the parameter names of transformers' Xcodec2Model (semantic_encoder.*, semantic_adapter.*,
    acoustic_encoder.*, fc_encoder.*, quantizer.project_in.*).  Unit-gain matrices, perturbed norm affine terms, non-zero
    snake alpha/beta (their init is 0, which would hide the exp())."""
    rng = np.random.default_rng(seed)

    def n(*shape, s):
        return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * np.float32(s))

    w: Dict[str, torch.Tensor] = {}

    def lin(name, out, inp, bias=True, gain=1.0):
        w[name + ".weight"] = n(out, inp, s=gain * inp ** -0.5)
        if bias:
            w[name + ".bias"] = n(out, s=0.05)

    def conv(name, out, inp, k, bias=True, gain=1.0):
        w[name + ".weight"] = n(out, inp, k, s=gain * (inp * k) ** -0.5)
        if bias:
            w[name + ".bias"] = n(out, s=0.05)

    def ln(name, dim):
        w[name + ".weight"] = 1.0 + n(dim, s=0.1)
        w[name + ".bias"] = n(dim, s=0.05)

    H, I = cfg.sem_hidden, cfg.sem_ffn
    p = "semantic_encoder."
    ln(p + "feature_projection.layer_norm", cfg.sem_feat_dim)
    lin(p + "feature_projection.projection", H, cfg.sem_feat_dim)
    hd = H // cfg.sem_heads
    for i in range(cfg.sem_layers):
        q = f"{p}encoder.layers.{i}."
        ln(q + "ffn1_layer_norm", H)
        lin(q + "ffn1.intermediate_dense", I, H)
        lin(q + "ffn1.output_dense", H, I)
        ln(q + "self_attn_layer_norm", H)
        for nm in ("linear_q", "linear_k", "linear_v", "linear_out"):
            lin(q + "self_attn." + nm, H, H)
        w[q + "self_attn.distance_embedding.weight"] = n(cfg.sem_left + cfg.sem_right + 1, hd, s=0.5)
        ln(q + "conv_module.layer_norm", H)
        conv(q + "conv_module.pointwise_conv1", 2 * H, H, 1, bias=False)
        conv(q + "conv_module.depthwise_conv", H, 1, cfg.sem_conv_kernel, bias=False, gain=2.0)
        ln(q + "conv_module.depthwise_layer_norm", H)
        conv(q + "conv_module.pointwise_conv2", H, H, 1, bias=False)
        ln(q + "ffn2_layer_norm", H)
        lin(q + "ffn2.intermediate_dense", I, H)
        lin(q + "ffn2.output_dense", H, I)
        ln(q + "final_layer_norm", H)
    conv("semantic_adapter.conv1", H, H, 3, bias=False)
    conv("semantic_adapter.conv2", H, H, 3)
    conv("semantic_adapter.conv3", H, H, 3)
    conv("semantic_adapter.conv4", H, H, 3, bias=False)

    def snake(name, dim):
        w[name + ".act.alpha"] = n(dim, s=0.3)
        w[name + ".act.beta"] = n(dim, s=0.3)

    a = "acoustic_encoder."
    conv(a + "conv1", cfg.ac_hidden, 1, 7, gain=3.0)
    for bi, stride in enumerate(cfg.ratios):
        dim = cfg.ac_hidden * 2 ** (bi + 1)
        b = f"{a}block.{bi}."
        for u in (1, 2, 3):
            snake(f"{b}res_unit{u}.snake1", dim // 2)
            conv(f"{b}res_unit{u}.conv1", dim // 2, dim // 2, 7, gain=0.7)
            snake(f"{b}res_unit{u}.snake2", dim // 2)
            conv(f"{b}res_unit{u}.conv2", dim // 2, dim // 2, 1, gain=0.7)
        snake(b + "snake1", dim // 2)
        conv(b + "conv1", dim, dim // 2, 2 * stride)
    d_model = cfg.ac_hidden * 2 ** len(cfg.ratios)
    snake(a + "snake1", d_model)
    conv(a + "conv2", cfg.codec_hidden, d_model, 3, gain=0.25)
    lin("fc_encoder", cfg.cat_dim, cfg.cat_dim)
    lin("quantizer.project_in", len(cfg.levels), cfg.cat_dim, gain=0.6)    # latents spread over all levels, few saturated
    return w


def synthetic_speech(n_samples: int, seed: int = 0, sample_rate: int = 16000) -> np.ndarray:
    """A seeded speech-like test signal in [-1, 1]: a few gliding harmonics under a slow envelope + a little noise."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples) / sample_rate
    f0 = 110.0 + 40.0 * np.sin(2 * np.pi * 0.7 * t + rng.uniform(0, 6.28))
    ph = 2 * np.pi * np.cumsum(f0) / sample_rate
    x = sum(a * np.sin(k * ph + rng.uniform(0, 6.28)) for k, a in ((1, 0.5), (2, 0.3), (3, 0.2), (5, 0.1), (9, 0.05)))
    env = 0.55 + 0.45 * np.sin(2 * np.pi * 2.3 * t + rng.uniform(0, 6.28))
    x = x * env + 0.01 * rng.standard_normal(n_samples)
    return (0.5 * x / np.max(np.abs(x))).astype(np.float32)


# --------------------------------------------------------------------------------------
# text front-end stand-ins (class-level tests, golden generators)
# --------------------------------------------------------------------------------------
class ByteTokenizer:
    """Minimal stand-in for the HF tokenizer of a NeuTTS checkpoint (none is reachable offline): every special / `<|speech_N|>`
    token is one id, text is byte-level.  Used by the class-level tests and the golden-vector generators."""
    SPECIALS = ["<|TEXT_REPLACE|>", "<|SPEECH_REPLACE|>", "<|TEXT_PROMPT_START|>", "<|TEXT_PROMPT_END|>",
                "<|SPEECH_GENERATION_START|>", "<|SPEECH_GENERATION_END|>"]

    def __init__(self, n_codes):
        self.base_special = 256
        self.speech_base = self.base_special + len(self.SPECIALS)
        self.n_codes = n_codes
        self.vocab_size = self.speech_base + n_codes

    def convert_tokens_to_ids(self, tok):
        if tok in self.SPECIALS:
            return self.base_special + self.SPECIALS.index(tok)
        m = re.fullmatch(r"<\|speech_(\d+)\|>", tok)
        return self.speech_base + int(m.group(1))

    def encode(self, text, add_special_tokens=True):
        ids, pos = [], 0
        pat = re.compile(r"<\|[A-Za-z_0-9]+\|>")
        for m in pat.finditer(text):
            ids += list(text[pos:m.start()].encode())
            ids.append(self.convert_tokens_to_ids(m.group(0)))
            pos = m.end()
        return ids + list(text[pos:].encode())

    def decode(self, ids, add_special_tokens=False):
        out = []
        for i in ids:
            if i >= self.speech_base:
                out.append(f"<|speech_{i - self.speech_base}|>")
            elif i >= self.base_special:
                out.append(self.SPECIALS[i - self.base_special])
            else:
                out.append(chr(i))
        return "".join(out)


class LowercasePhonemizer:
    """Stand-in for the espeak phonemizer (not installed here; the text front-end is off the hot path)."""

    def phonemize(self, texts):
        return [t.lower() for t in texts]
