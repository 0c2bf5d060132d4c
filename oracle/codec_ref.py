"""CPU restatement of the NeuCodec decoder hot path (TEST INFRASTRUCTURE, see oracle/__init__.py).

What the reference runs (ref:neutts/neutts.py:273-295):  codec.decode_code(LongTensor[1,1,T]) -> float32[1,1,480*T]
with codec = neucodec.NeuCodec (`neucodec>=0.0.4`, ref:requirements.txt:2) -- an un-vendored dependency that is NOT
installable here.  Its decoder (FSQ de-index -> Linear 8->2048 -> Linear 2048->1024 -> Vocos backbone -> ISTFT head)
is the XCodec2 decoder at hop 480; transformers 5.15 ships that architecture as models/xcodec2, and this file
restates it op by op, each function citing hf:models/xcodec2/modeling_xcodec2.py.  tests/test_oracle_pin.py checks
the restatement against the live Xcodec2Quantizer/Xcodec2Decoder modules at NeuCodec geometry and
tests/golden/codec_*.npz holds fixtures generated from them (oracle/gen_golden_codec.py).

PARITY UNPINNED at one boundary: the equivalence "NeuCodec.decode_code == xcodec2 decoder @ hop 480" rests on the
survey author's reading of the neucodec source and cannot be verified offline.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from synthetic import CodecConfig  # noqa: F401  (geometry + seeded synthetic weights are plain data: synthetic.py)
from synthetic import make_codec_weights as make_weights  # noqa: F401


def fsq_codebook(cfg: CodecConfig) -> torch.Tensor:
    """Xcodec2FiniteScalarQuantization._compute_buffers  hf:...modeling_xcodec2.py:676-690: digit d of index i in
    base `levels`, value (d - L//2) / (L//2)."""
    levels = torch.tensor(cfg.levels, dtype=torch.int32)
    basis = torch.cumprod(torch.tensor([1] + list(cfg.levels[:-1])), dim=0, dtype=torch.int32)
    idx = torch.arange(int(np.prod(cfg.levels))).unsqueeze(-1)
    digits = (idx // basis) % levels
    half = levels // 2
    return (digits - half) / half


def from_codes(cfg, w, codes: torch.Tensor) -> torch.Tensor:
    """Xcodec2Quantizer.from_codes  hf:...modeling_xcodec2.py:806-809.  codes int64 [B, T] -> [B, T, quantization_dim]."""
    cb = fsq_codebook(cfg).to(w["quantizer.project_out.weight"].dtype)
    return F.linear(cb[codes], w["quantizer.project_out.weight"], w["quantizer.project_out.bias"])


def resnet_block(w, p, x):
    """Xcodec2ResNetBlock.forward  hf:...modeling_xcodec2.py:650-660  (x: [B, T, C]; dropout is off in eval)."""
    h = x.transpose(1, 2)
    res = h
    h = F.group_norm(h, 32, w[p + "norm1.weight"], w[p + "norm1.bias"], eps=1e-6)
    h = F.silu(h)
    h = F.conv1d(h, w[p + "conv1.weight"], w[p + "conv1.bias"], padding=1)
    h = F.group_norm(h, 32, w[p + "norm2.weight"], w[p + "norm2.bias"], eps=1e-6)
    h = F.silu(h)
    h = F.conv1d(h, w[p + "conv2.weight"], w[p + "conv2.bias"], padding=1)
    return (h + res).transpose(1, 2)


def rms_norm(x, weight, eps):
    """Xcodec2RMSNorm.forward  hf:...modeling_xcodec2.py:320-325."""
    dt = x.dtype
    x32 = x.to(torch.float32)
    x32 = x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + eps)
    return weight * x32.to(dt)


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def decoder_layer(cfg, w, i, h, cos, sin):
    """Xcodec2DecoderLayer.forward :345-373 with Xcodec2Attention.forward :268-308 (non-causal, RoPE whose
    "positions" are the HEAD indices, unsqueeze_dim=2 -- hf:...modeling_xcodec2.py:284-287,849-854) and
    Xcodec2MLP.forward :162-166 (fc2(silu(fc1)), no gate)."""
    p = f"decoder.layers.{i}."
    B, T, H = h.shape
    nh, d = cfg.num_heads, cfg.head_dim
    res = h
    x = rms_norm(h, w[p + "input_layernorm.weight"], cfg.rms_eps)
    q = F.linear(x, w[p + "self_attn.q_proj.weight"]).view(B, T, nh, d).transpose(1, 2)
    k = F.linear(x, w[p + "self_attn.k_proj.weight"]).view(B, T, nh, d).transpose(1, 2)
    v = F.linear(x, w[p + "self_attn.v_proj.weight"]).view(B, T, nh, d).transpose(1, 2)
    c, s = cos.unsqueeze(2), sin.unsqueeze(2)               # [1, nh, 1, d]
    q = q * c + rotate_half(q) * s
    k = k * c + rotate_half(k) * s
    aw = torch.matmul(q, k.transpose(2, 3)) * d ** -0.5
    aw = F.softmax(aw, dim=-1, dtype=torch.float32).to(q.dtype)
    a = torch.matmul(aw, v).transpose(1, 2).contiguous().reshape(B, T, -1)
    h = res + F.linear(a, w[p + "self_attn.o_proj.weight"])
    res = h
    x = rms_norm(h, w[p + "post_attention_layernorm.weight"], cfg.rms_eps)
    x = F.linear(F.silu(F.linear(x, w[p + "mlp.fc1.weight"])), w[p + "mlp.fc2.weight"])
    return res + x


def istft_head(cfg, w, h):
    """Xcodec2ISTFTHead.forward  hf:...modeling_xcodec2.py:762-796 ("same"-padded ISTFT from Vocos)."""
    n_fft, hop = cfg.n_fft, cfg.hop_length
    pad = (n_fft - hop) // 2
    window = torch.hann_window(n_fft)
    spec = F.linear(h, w["decoder.head.linear.weight"], w["decoder.head.linear.bias"]).transpose(1, 2)
    mag, phase = spec.chunk(2, dim=1)
    mag = torch.exp(mag.float()).clamp(max=1e2)
    cplx = torch.polar(mag, phase.float())
    frames = torch.fft.irfft(cplx, n_fft, dim=1, norm="backward") * window[None, :, None]
    T = cplx.shape[-1]
    out_size = (T - 1) * hop + n_fft
    audio = F.fold(frames, output_size=(1, out_size), kernel_size=(1, n_fft), stride=(1, hop))[:, 0, 0, pad:-pad]
    env = F.fold(window.square().expand(1, T, -1).transpose(1, 2), output_size=(1, out_size),
                 kernel_size=(1, n_fft), stride=(1, hop)).squeeze()[pad:-pad]
    return (audio / env.clamp(min=1e-11)).unsqueeze(1)


def decode_code(cfg: CodecConfig, w, codes: torch.Tensor, taps=None) -> torch.Tensor:
    """NeuCodec.decode_code as restated by Xcodec2Model.decode :1028-1049 + Xcodec2Decoder.forward :838-862.
    codes: int64 [B, 1, T] (the layout ref:neutts/neutts.py:288 builds) -> float32 [B, 1, hop*T]."""
    with torch.no_grad():
        x = from_codes(cfg, w, codes[:, 0, :])
        h = F.linear(x, w["decoder.fc.weight"], w["decoder.fc.bias"])
        h = F.conv1d(h.transpose(1, 2), w["decoder.embed.weight"], w["decoder.embed.bias"], padding=3).transpose(1, 2)
        if taps is not None: taps["embed"] = h
        for b in range(2):
            h = resnet_block(w, f"decoder.prior_net.{b}.", h)
        if taps is not None: taps["prior"] = h
        # RoPE "positions" = arange(num_heads): cos/sin [1, nh, d]   (hf:...modeling_xcodec2.py:849-854)
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float) / cfg.head_dim))
        fr = torch.arange(cfg.num_heads, dtype=torch.float)[:, None] * inv[None, :]
        emb = torch.cat((fr, fr), dim=-1)[None]
        cos, sin = emb.cos().to(h.dtype), emb.sin().to(h.dtype)
        for i in range(cfg.num_layers):
            h = decoder_layer(cfg, w, i, h, cos, sin)
        if taps is not None: taps["layers"] = h
        for b in range(2):
            h = resnet_block(w, f"decoder.post_net.{b}.", h)
        if taps is not None: taps["post"] = h
        h = F.layer_norm(h, (cfg.hidden_size,), w["decoder.norm.weight"], w["decoder.norm.bias"], eps=1e-6)
        if taps is not None: taps["norm"] = h
        return istft_head(cfg, w, h)


def linear_overlap_add(frames, stride: int) -> np.ndarray:
    """Restatement of ref:neutts/neutts.py:46-70 (_linear_overlap_add, from encodec): triangular-weight cross-fade."""
    dtype = frames[0].dtype
    total = max(stride * i + f.shape[-1] for i, f in enumerate(frames))
    sum_w = np.zeros(total, dtype=dtype)
    out = np.zeros(total, dtype=dtype)
    off = 0
    for f in frames:
        n = f.shape[-1]
        t = np.linspace(0, 1, n + 2, dtype=dtype)[1:-1]
        wgt = np.abs(0.5 - (t - 0.5))
        out[off:off + n] += wgt * f
        sum_w[off:off + n] += wgt
        off += stride
    assert sum_w.min() > 0
    return out / sum_w
