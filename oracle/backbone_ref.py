"""CPU restatement of the NeuTTS backbone hot path (TEST INFRASTRUCTURE, see oracle/__init__.py).

What the reference runs on this path (ref:neutts/neutts.py:334-352):
    backbone.generate(prompt[1,S], max_length=2048, eos_token_id=<|SPEECH_GENERATION_END|>,
                      do_sample=True, temperature=1.0, top_k=50, use_cache=True, min_new_tokens=50)
with backbone = transformers.AutoModelForCausalLM (Qwen2ForCausalLM for NeuTTS-Air).  The
arithmetic lives in the un-vendored dependency `transformers` (pinned 4.56.1 in
ref:requirements.txt:8; 5.15.0 installed here).  This file restates it with plain torch CPU
ops in the same order and with the same rounding points, each function citing the
transformers file:line (hf: = transformers/) it follows.  tests/test_oracle_pin.py checks it
bit-for-bit against the real Qwen2ForCausalLM (eager attention), and tests/golden/ holds
fixtures produced by oracle/gen_golden.py from that same dependency.

Rounding contract: every tensor lives in `dtype` (float32 or
bfloat16); each torch op therefore rounds exactly where the HF module rounds.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# model geometry + seeded synthetic weights / prompts are plain data shared with bench.py and the tools (synthetic.py)
from synthetic import BackboneConfig, cast_weights, make_weights, rope_inv_freq, synthetic_prompt  # noqa: F401
from synthetic import FP8_MAX, default_fp8_input_scales  # noqa: F401


# --------------------------------------------------------------------------------------
# fp8 model variant: e4m3 weights (per-output-channel scale) x e4m3 GEMM inputs (static per-tensor scale), fp32 accumulate,
# bf16 everywhere else.  There is no third-party reference for this variant on the path (the reference runs bf16 / fp32
# weights, or llama.cpp's own quantisations): this restatement of the scheme of static-fp8 checkpoints IS the specification
# the HIP path is checked against -- same quantisation points, same rounding (RNE, clamped to +-448), same scale algebra.
# MODEL-level parity is therefore unpinned; the arithmetic building block is pinned: `linear` on fp8 entries agrees with the
# third-party torch._scaled_mm (CPU, oneDNN: e4m3 x e4m3, fp32 accumulation) on the same e4m3 operands up to the order of the
# fp32 sums (tests/test_oracle_pin.py::test_fp8_linear_vs_torch_scaled_mm), the e4m3 conversion is torch's own.
# --------------------------------------------------------------------------------------
def fp8_quantize_weights(w: Dict[str, torch.Tensor], input_scales: Dict[str, float]) -> Dict[str, torch.Tensor]:
    """bf16 state dict -> the same dict with, for every Linear on the path (and the head), three extra entries:
    name + "::q" (float32 tensor holding the e4m3 VALUES of the quantised weight), name + "::scale" (fp32 [N]) and
    name + "::in_scale" (python float).  The original tensors stay (embedding gather, biases, norms)."""
    out = dict(w)

    def quant(name, xs):
        wt = w[name].to(torch.bfloat16).to(torch.float32)
        am = wt.abs().amax(dim=1)
        sc = torch.where(am > 0, am / FP8_MAX, torch.ones_like(am))
        q = (wt / sc[:, None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).to(torch.float32)
        out[name + "::q"], out[name + "::scale"], out[name + "::in_scale"] = q, sc, float(xs)

    n_layers = 1 + max(int(k.split(".")[2]) for k in w if k.startswith("model.layers."))
    for i in range(n_layers):
        p = f"model.layers.{i}."
        for t, sname in (("self_attn.q_proj", "self_attn.q_proj"), ("self_attn.k_proj", "self_attn.q_proj"), ("self_attn.v_proj", "self_attn.q_proj"),
                         ("self_attn.o_proj", "self_attn.o_proj"), ("mlp.gate_proj", "mlp.gate_proj"), ("mlp.up_proj", "mlp.gate_proj"),
                         ("mlp.down_proj", "mlp.down_proj")):
            quant(p + t + ".weight", input_scales[p + sname + ".input_scale"])
    head = "lm_head.weight" if "lm_head.weight" in w else "model.embed_tokens.weight"
    quant(head, input_scales["lm_head.input_scale"])
    if head != "lm_head.weight":
        for suf in ("::q", "::scale", "::in_scale"):
            out["lm_head.weight" + suf] = out.pop(head + suf)
    return out


FP8_ACT_HOOK = None     # tests only (tests/test_oracle_fp8_sensitivity.py): applied to a GEMM input right before its e4m3 quantisation


def fp8_act(x: torch.Tensor, in_scale: float) -> torch.Tensor:
    """e4m3 VALUES (as fp32) of a GEMM input: e4m3(clamp(x / in_scale)), RNE."""
    if FP8_ACT_HOOK is not None:
        x = FP8_ACT_HOOK(x)
    return (x.to(torch.float32) * (1.0 / in_scale)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).to(torch.float32)


def linear(x: torch.Tensor, w: Dict[str, torch.Tensor], name: str, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear in the model dtype (hf:models/qwen2/modeling_qwen2.py:46-48,206-208,233), or -- when `w` carries the fp8
    entries of `name` -- acc = q(x) @ Wq^T in fp32, y = fma(acc, in_scale * w_scale[n], bias) rounded once to the model dtype."""
    if name + "::q" not in w:
        return F.linear(x, w[name], bias)
    xs = w[name + "::in_scale"]
    acc = fp8_act(x, xs) @ w[name + "::q"].t()                                   # exact products, fp32 sums
    sc = (torch.tensor(xs, dtype=torch.float32) * w[name + "::scale"]).to(torch.float64)
    y = acc.to(torch.float64) * sc
    if bias is not None:
        y = y + bias.to(torch.float64)                                           # = one fma per element (fp64 holds the product exactly)
    return y.to(torch.float32).to(x.dtype)


# --------------------------------------------------------------------------------------
# modules
# --------------------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """Qwen2RMSNorm.forward  hf:models/qwen2/modeling_qwen2.py:247-252."""
    dt = x.dtype
    x32 = x.to(torch.float32)
    var = x32.pow(2).mean(-1, keepdim=True)
    x32 = x32 * torch.rsqrt(var + eps)
    return weight * x32.to(dt)


def rope_cos_sin(cfg: BackboneConfig, positions: torch.Tensor, dtype: torch.dtype):
    """Qwen2RotaryEmbedding.forward  hf:models/qwen2/modeling_qwen2.py:91-102.
    positions: int64 [S] -> cos, sin [1, S, d] in `dtype` (fp32 math, then cast)."""
    inv = rope_inv_freq(cfg)[None, :, None].float()            # [1, d/2, 1]
    pos = positions[None, None, :].float()                      # [1, 1, S]
    freqs = (inv @ pos).transpose(1, 2)                         # [1, S, d/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """hf:models/qwen2/modeling_qwen2.py:105-109."""
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin):
    """apply_rotary_pos_emb (unsqueeze_dim=1)  hf:models/qwen2/modeling_qwen2.py:113-135."""
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def repeat_kv(x: torch.Tensor, n_rep: int) -> torch.Tensor:
    """hf:models/qwen2/modeling_qwen2.py:138-147."""
    b, h, s, d = x.shape
    if n_rep == 1:
        return x
    return x[:, :, None, :, :].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


def eager_attention(q, k, v, mask, scaling, n_rep):
    """eager_attention_forward  hf:models/qwen2/modeling_qwen2.py:150-172
    (the oracle pins attn_implementation="eager": fully specified in Python)."""
    k = repeat_kv(k, n_rep)
    v = repeat_kv(v, n_rep)
    aw = torch.matmul(q, k.transpose(2, 3)) * scaling
    if mask is not None:
        aw = aw + mask
    aw = F.softmax(aw, dim=-1, dtype=torch.float32).to(q.dtype)
    out = torch.matmul(aw, v)
    return out.transpose(1, 2).contiguous()


def causal_mask(q_len: int, kv_len: int, dtype: torch.dtype) -> Optional[torch.Tensor]:
    """Additive causal mask as create_causal_mask builds it for the eager path
    (hf:masking_utils.py: 0 where allowed, finfo.min where masked).  For a single query over a
    full cache nothing is masked; HF then passes mask=None or all-zeros (same result)."""
    if q_len == 1:
        return None
    past = kv_len - q_len
    i = torch.arange(q_len)[:, None] + past
    j = torch.arange(kv_len)[None, :]
    m = torch.zeros(q_len, kv_len, dtype=dtype)
    m.masked_fill_(j > i, torch.finfo(dtype).min)
    return m[None, None]


class KVCache:
    """DynamicLayer.update  hf:cache_utils.py:127-146 -- cat along the sequence axis."""

    def __init__(self, n_layers: int):
        self.k: List[Optional[torch.Tensor]] = [None] * n_layers
        self.v: List[Optional[torch.Tensor]] = [None] * n_layers

    def update(self, i, k, v):
        self.k[i] = k if self.k[i] is None else torch.cat([self.k[i], k], dim=-2)
        self.v[i] = v if self.v[i] is None else torch.cat([self.v[i], v], dim=-2)
        return self.k[i], self.v[i]

    def length(self) -> int:
        return 0 if self.k[0] is None else self.k[0].shape[-2]


def decoder_layer(cfg, w, i, h, cos, sin, cache: KVCache, taps=None):
    """Qwen2DecoderLayer.forward  hf:models/qwen2/modeling_qwen2.py:269-298 with
    Qwen2Attention.forward :195-234 and Qwen2MLP.forward :46-48 inlined."""
    p = f"model.layers.{i}."
    B, S, H = h.shape
    nh, nkv, d = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
    resid = h
    x = rms_norm(h, w[p + "input_layernorm.weight"], cfg.rms_eps)
    q = linear(x, w, p + "self_attn.q_proj.weight", w.get(p + "self_attn.q_proj.bias")).view(B, S, nh, d).transpose(1, 2)
    k = linear(x, w, p + "self_attn.k_proj.weight", w.get(p + "self_attn.k_proj.bias")).view(B, S, nkv, d).transpose(1, 2)
    v = linear(x, w, p + "self_attn.v_proj.weight", w.get(p + "self_attn.v_proj.bias")).view(B, S, nkv, d).transpose(1, 2)
    if getattr(cfg, "qk_norm", False):
        # Qwen3Attention.forward  hf:models/qwen3/modeling_qwen3.py: q_norm / k_norm (Qwen3RMSNorm over head_dim, same arithmetic as
        # Qwen2RMSNorm) on the projected heads, BEFORE RoPE; applied on [B, S, heads, d] there, on [B, heads, S, d] here: per (token, head) alike
        q = rms_norm(q, w[p + "self_attn.q_norm.weight"], cfg.rms_eps)
        k = rms_norm(k, w[p + "self_attn.k_norm.weight"], cfg.rms_eps)
    q, k = apply_rope(q, k, cos, sin)
    kk, vv = cache.update(i, k, v)
    mask = causal_mask(S, kk.shape[-2], h.dtype)
    a = eager_attention(q, kk, vv, mask, d ** -0.5, nh // nkv)
    a = a.reshape(B, S, -1).contiguous()
    o = linear(a, w, p + "self_attn.o_proj.weight")
    h = resid + o
    resid = h
    x2 = rms_norm(h, w[p + "post_attention_layernorm.weight"], cfg.rms_eps)
    g = linear(x2, w, p + "mlp.gate_proj.weight")
    u = linear(x2, w, p + "mlp.up_proj.weight")
    m = linear(F.silu(g) * u, w, p + "mlp.down_proj.weight")
    out = resid + m
    if taps is not None:
        taps.append(dict(x=x, q=q, k=k, v=v, attn=a, o=o, h_mid=h, x2=x2, act=F.silu(g) * u, mlp=m, h_out=out))
    return out


def model_forward(cfg: BackboneConfig, w, ids: torch.Tensor, cache: KVCache, taps=None) -> torch.Tensor:
    """Qwen2Model.forward :342-402 + Qwen2ForCausalLM.forward :423-477 (logits of the LAST
    position only, which is all generate() consumes: hf:generation/utils.py:2894).
    ids: int64 [1, q].  Returns logits [1, V] in the model dtype."""
    dtype = w["model.embed_tokens.weight"].dtype
    past = cache.length()
    h = F.embedding(ids, w["model.embed_tokens.weight"])
    pos = torch.arange(past, past + ids.shape[1])
    cos, sin = rope_cos_sin(cfg, pos, dtype)
    for i in range(cfg.num_layers):
        lt = [] if taps is not None else None
        h = decoder_layer(cfg, w, i, h, cos, sin, cache, lt)
        if taps is not None:
            taps.append(lt[0])
    h = rms_norm(h, w["model.norm.weight"], cfg.rms_eps)
    if "lm_head.weight::q" in w:
        return linear(h[:, -1, :], w, "lm_head.weight")
    return F.linear(h[:, -1, :], w.get("lm_head.weight", w["model.embed_tokens.weight"]))  # tied unless the checkpoint unties it


def gemm_input_amax(cfg: BackboneConfig, w, prompts) -> Dict[str, float]:
    """max |x| of every GEMM's input over the prompt passes of `prompts`, under the `input_scale` names of a static-fp8 checkpoint -- what
    the engine's calibration mode records (ntts_backbone_calibrate; tests/test_emu_variants.py checks one against the other).  The
    lm_head's record covers the final norm of EVERY position: generation feeds it one new position per step (hf:generation/utils.py:2894),
    and a record of the prompts' last positions alone would rest on one row per calibration prompt."""
    out: Dict[str, float] = {}

    def upd(k, t):
        out[k] = max(out.get(k, 0.0), float(t.abs().max()))
    with torch.no_grad():
        for p in prompts:
            taps: list = []
            model_forward(cfg, w, torch.tensor([list(p)], dtype=torch.long), KVCache(cfg.num_layers), taps)
            for i, t in enumerate(taps):
                pre = f"model.layers.{i}."
                upd(pre + "self_attn.q_proj.input_scale", t["x"])
                upd(pre + "self_attn.o_proj.input_scale", t["attn"])
                upd(pre + "mlp.gate_proj.input_scale", t["x2"])
                upd(pre + "mlp.down_proj.input_scale", t["act"])
            upd("lm_head.input_scale", rms_norm(taps[-1]["h_out"], w["model.norm.weight"], cfg.rms_eps))
    return out


@dataclass
class GenResult:
    ids: List[int]                 # generated ids only (prompt stripped, like ref:neutts/neutts.py:348-351)
    margins: List[float]           # top1-top2 fp32 logit gap at every step (after EOS masking)
    logits: Optional[List[torch.Tensor]] = None


def generate(cfg: BackboneConfig, w, prompt_ids: List[int], max_length: int, eos_id: int,
             min_new_tokens: int = 50, do_sample: bool = False, top_k: int = 50,
             generator: Optional[torch.Generator] = None, keep_logits: bool = False,
             force_ids: Optional[List[int]] = None) -> GenResult:
    """GenerationMixin._sample  hf:generation/utils.py:2783-2973, as the reference calls it.

    - logits of the last position are copied to fp32                               (:2894)
    - MinNewTokensLengthLogitsProcessor: eos=-inf while new tokens < min_new_tokens (hf:generation/logits_process.py:164-236)
    - greedy: argmax (:2925); sampling: TopK(50) (:542-595) -> softmax -> multinomial (:2920-2923)
    - stop at EOS or when total length reaches max_length (hf:generation/stopping_criteria.py:58-84,534-582)
    `force_ids` teacher-forces the continuation (used by the margin-aware parity tests).
    """
    cache = KVCache(cfg.num_layers)
    ids = torch.tensor([list(prompt_ids)], dtype=torch.long)
    cur = ids
    out: List[int] = []
    margins: List[float] = []
    kept: List[torch.Tensor] = []
    n_prompt = len(prompt_ids)
    with torch.no_grad():
        while n_prompt + len(out) < max_length:
            logits = model_forward(cfg, w, cur, cache).to(torch.float32)[0].clone()
            if len(out) < min_new_tokens:
                logits[eos_id] = -float("inf")
            top2 = torch.topk(logits, 2).values
            margins.append(float(top2[0] - top2[1]))
            if keep_logits:
                kept.append(logits)
            if do_sample:
                kth = torch.topk(logits, min(top_k, logits.numel())).values[-1]
                sc = logits.masked_fill(logits < kth, -float("inf"))
                probs = F.softmax(sc, dim=-1)
                nxt = int(torch.multinomial(probs, 1, generator=generator))
            else:
                nxt = int(torch.argmax(logits))
            # teacher forcing: record the oracle's own choice, feed the forced id
            nxt_fed = nxt if force_ids is None else int(force_ids[len(out)])
            out.append(nxt)
            cur = torch.tensor([[nxt_fed]], dtype=torch.long)
            if force_ids is None and nxt == eos_id:
                break
            if force_ids is not None and len(out) >= len(force_ids):
                break
    return GenResult(out, margins, kept if keep_logits else None)


def bf16_ulp(x: float) -> float:
    x = abs(float(x))
    return 2.0 ** -133 if x == 0 else 2.0 ** (np.floor(np.log2(x)) - 7)


def assert_free_run_matches(got_ids: List[int], ref: GenResult, max_ulps: float = 2.0) -> None:
    """Free-running greedy ids of an implementation vs a GenResult produced with keep_logits=True: identical, or identical
    up to the first step where the ORACLE's own top-2 logits are within `max_ulps` bf16 ulps (exact bf16 ties do occur
    with synthetic weights; which side wins then depends on fp32 summation order -- even between two CPUs running this
    oracle), in which case the implementation's token must be one of those two."""
    want = ref.ids
    k = next((i for i in range(min(len(got_ids), len(want))) if got_ids[i] != want[i]), None)
    if k is None:
        assert len(got_ids) == len(want), (len(got_ids), len(want))
        return
    top = torch.topk(ref.logits[k], 2)
    gap = float(top.values[0] - top.values[1])
    assert gap <= max_ulps * bf16_ulp(float(top.values[0])), \
        f"step {k}: got {got_ids[k]}, oracle {want[k]}, oracle top-2 gap {gap}"
    assert got_ids[k] in top.indices.tolist(), (k, got_ids[k], top.indices.tolist())


