"""CPU restatement of the NeuCodec ENCODER path (TEST INFRASTRUCTURE, see oracle/__init__.py).

What the reference runs (ref:neutts/neutts.py:266-271):
    wav, _ = librosa.load(path, sr=16000, mono=True);  codec.encode_code(wav[1,1,L]) -> int codes [1,1,T],  T = L // 320 + 1
with codec = neucodec.NeuCodec (`neucodec>=0.0.4`, ref:requirements.txt:2), un-vendored and not installable here.  Its
encoder (zero-pad to a hop multiple -> [w2v-BERT 2.0 fbank features -> conformer layers 1..16 -> semantic adapter] ||
[acoustic conv encoder] -> concat -> Linear -> FSQ) is the XCodec2 encoder; transformers 5.15 ships that architecture
as models/xcodec2 + models/wav2vec2_bert, and this file restates it op by op, each function citing the hf: source it
follows.  tests/test_oracle_pin.py checks the restatement against the live `Xcodec2Model.encode` (same weights, same
features) and the fbank front-end against the live `SeamlessM4TFeatureExtractor` (the extractor neucodec instantiates
for "facebook/w2v-bert-2.0"; transformers' own Xcodec2FeatureExtractor needs torchaudio, absent here).

PARITY UNPINNED at one boundary: "NeuCodec.encode_code == xcodec2 encoder" rests on the survey author's reading of the
neucodec source, like the decoder's; it cannot be verified offline.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from synthetic import EncoderConfig  # noqa: F401  (geometry + seeded synthetic weights are plain data: synthetic.py)
from synthetic import make_encoder_weights as make_weights  # noqa: F401

FRAME, SHIFT, NFFT, NMEL = 400, 160, 512, 80


# ---------------------------------------------------------------------------------------------------- audio -> features
def pad_audio(wav: np.ndarray, hop: int = 320) -> np.ndarray:
    """hf:models/xcodec2/feature_extraction_xcodec2.py:149-158 (and xcodec2's original `pad_for_wav = 320 - L % 320`):
    one zero is appended, then zeros up to the next hop multiple -- a clip that already is a multiple grows by a whole hop."""
    L = wav.shape[-1]
    Lp = (L + 1 + hop - 1) // hop * hop
    return np.pad(wav.astype(np.float32), (0, Lp - L))


def povey_window() -> np.ndarray:
    """hf:audio_utils.py window_function("povey", periodic=False): hann(400, symmetric) ** 0.85."""
    n = np.arange(FRAME, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / (FRAME - 1))) ** 0.85


def kaldi_mel_filters(sample_rate: int = 16000) -> np.ndarray:
    """hf:audio_utils.py mel_filter_bank(257, 80, 20, sr/2, sr, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    -> [257, 80]: triangles drawn in mel space, mel(f) = 1127 ln(1 + f/700)."""
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)   # noqa: E731
    nb = NFFT // 2 + 1
    edges = np.linspace(mel(20.0), mel(sample_rate // 2), NMEL + 2)
    bins = mel(sample_rate / ((nb - 1) * 2) * np.arange(nb))
    diff = np.diff(edges)
    slopes = edges[None, :] - bins[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def fbank_features(wav_padded: np.ndarray) -> np.ndarray:
    """SeamlessM4TFeatureExtractor.__call__ on F.pad(wav_padded, (160, 160))  (neucodec: `feature_extractor(F.pad(y, (160, 160)))`;
    hf:models/seamless_m4t/feature_extraction_seamless_m4t.py:115-140,254-290 + hf:audio_utils.py spectrogram()):
    x * 2^15 -> frames of 400 every 160 (no centring) -> minus frame mean -> pre-emphasis 0.97 (first sample * 0.03) ->
    povey window -> |rfft_512|^2 -> kaldi mel (floor 1.19e-7) -> ln -> per-bin (x - mean) / sqrt(var_ddof1 + 1e-7) over
    the frames -> frame pairs stacked: [T, 160] float32."""
    x = np.pad(wav_padded.astype(np.float32), (SHIFT, SHIFT)).astype(np.float64) * 2.0 ** 15
    nfr = 1 + (x.size - FRAME) // SHIFT
    win, melf = povey_window(), kaldi_mel_filters()
    feats = np.empty((nfr, NMEL), dtype=np.float64)
    buf = np.zeros(NFFT)
    for i in range(nfr):
        fr = x[i * SHIFT:i * SHIFT + FRAME].copy()
        fr -= fr.mean()
        fr[1:] -= 0.97 * fr[:-1].copy()
        fr[0] *= 1.0 - 0.97
        buf[:FRAME] = fr * win
        spec = np.fft.rfft(buf).astype(np.complex64)           # the reference stores the spectrum as complex64
        p = np.abs(spec, dtype=np.float64) ** 2.0
        feats[i] = np.log(np.maximum(1.192092955078125e-07, melf.T @ p))
    feats = feats.astype(np.float32)
    feats = (feats - feats.mean(0, keepdims=True)) / np.sqrt(feats.var(0, ddof=1, keepdims=True) + 1e-7)
    nfr -= nfr % 2
    return feats[:nfr].reshape(nfr // 2, 2 * NMEL).astype(np.float32)


# ---------------------------------------------------------------------------------------------------- semantic encoder
def conformer_layer(cfg: EncoderConfig, w: Dict[str, torch.Tensor], p: str, x: torch.Tensor) -> torch.Tensor:
    """Wav2Vec2BertEncoderLayer.forward  hf:models/wav2vec2_bert/modeling_wav2vec2_bert.py:423-461 (x: [T, H], no mask)."""
    eps, H, nh = cfg.sem_ln_eps, cfg.sem_hidden, cfg.sem_heads
    hd = H // nh

    def ln(name, v):
        return F.layer_norm(v, (v.shape[-1],), w[p + name + ".weight"], w[p + name + ".bias"], eps)

    def ffn(name, v):                                                    # Wav2Vec2BertFeedForward :147-154, act = swish
        v = F.silu(F.linear(v, w[p + name + ".intermediate_dense.weight"], w[p + name + ".intermediate_dense.bias"]))
        return F.linear(v, w[p + name + ".output_dense.weight"], w[p + name + ".output_dense.bias"])

    x = ffn("ffn1", ln("ffn1_layer_norm", x)) * 0.5 + x                 # :432-436
    # self-attention with "relative_key" positions  :263-337
    h = ln("self_attn_layer_norm", x)
    T = h.shape[0]
    q, k, v = (F.linear(h, w[p + f"self_attn.linear_{c}.weight"], w[p + f"self_attn.linear_{c}.bias"]).view(T, nh, hd).transpose(0, 1)
               for c in "qkv")
    scores = q @ k.transpose(-2, -1) / math.sqrt(hd)
    dist = torch.arange(T).view(1, -1) - torch.arange(T).view(-1, 1)      # key - query
    dist = dist.clamp(-cfg.sem_left, cfg.sem_right) + cfg.sem_left
    pe = w[p + "self_attn.distance_embedding.weight"][dist]              # [T, T, hd]
    scores = scores + torch.einsum("hld,lrd->hlr", q, pe) / math.sqrt(hd)
    h = (torch.softmax(scores, dim=-1) @ v).transpose(0, 1).reshape(T, H)
    x = F.linear(h, w[p + "self_attn.linear_out.weight"], w[p + "self_attn.linear_out.bias"]) + x
    # convolution module  :196-226
    h = ln("conv_module.layer_norm", x).t().unsqueeze(0)                 # [1, H, T]
    h = F.glu(F.conv1d(h, w[p + "conv_module.pointwise_conv1.weight"]), dim=1)
    h = F.pad(h, (cfg.sem_conv_kernel - 1, 0))                           # causal: all padding on the left
    h = F.conv1d(h, w[p + "conv_module.depthwise_conv.weight"], groups=H)
    h = F.silu(ln("conv_module.depthwise_layer_norm", h[0].t()))
    h = F.conv1d(h.t().unsqueeze(0), w[p + "conv_module.pointwise_conv2.weight"])[0].t()
    x = x + h
    x = ffn("ffn2", ln("ffn2_layer_norm", x)) * 0.5 + x                 # :454-458
    return ln("final_layer_norm", x)


def semantic_encoder(cfg: EncoderConfig, w, feats: torch.Tensor) -> torch.Tensor:
    """Wav2Vec2BertModel.forward (:1008-1028): LayerNorm(160) -> Linear(160 -> H) -> `sem_layers` conformer layers; the last
    hidden state of a 16-layer model == `hidden_states[16]` of the 24-layer w2v-BERT 2.0 neucodec reads."""
    p = "semantic_encoder."
    x = F.layer_norm(feats, (feats.shape[-1],), w[p + "feature_projection.layer_norm.weight"],
                     w[p + "feature_projection.layer_norm.bias"], cfg.sem_ln_eps)
    x = F.linear(x, w[p + "feature_projection.projection.weight"], w[p + "feature_projection.projection.bias"])
    for i in range(cfg.sem_layers):
        x = conformer_layer(cfg, w, f"{p}encoder.layers.{i}.", x)
    return x


def semantic_adapter(w, x: torch.Tensor) -> torch.Tensor:
    """Xcodec2SemanticAdapter.forward  hf:models/xcodec2/modeling_xcodec2.py:899-908  (x: [T, H] -> [T, H])."""
    h = x.t().unsqueeze(0)
    h = F.relu(F.conv1d(h, w["semantic_adapter.conv1.weight"], padding=1))
    r = h
    h = F.relu(F.conv1d(h, w["semantic_adapter.conv2.weight"], w["semantic_adapter.conv2.bias"], padding=1))
    h = F.conv1d(h, w["semantic_adapter.conv3.weight"], w["semantic_adapter.conv3.bias"], padding=1) + r
    return F.conv1d(h, w["semantic_adapter.conv4.weight"], padding=1)[0].t()


# ---------------------------------------------------------------------------------------------------- acoustic encoder
def kaiser_sinc_filter(cutoff: float, half_width: float, ksize: int) -> torch.Tensor:
    """kaiser_sinc_filter1d  hf:models/xcodec2/modeling_xcodec2.py:416-460 (12 taps, cutoff 0.25, half width 0.3)."""
    half = ksize // 2
    att = 2.285 * (half - 1) * math.pi * 4 * half_width + 7.95
    beta = 0.1102 * (att - 8.7) if att > 50.0 else (0.5842 * (att - 21) ** 0.4 + 0.07886 * (att - 21.0) if att >= 21.0 else 0.0)
    win = torch.kaiser_window(ksize, beta=beta, periodic=False, dtype=torch.float32)
    t = (torch.arange(-half, half) + 0.5) if ksize % 2 == 0 else (torch.arange(ksize) - half)
    f = 2 * cutoff * win * torch.sinc(2 * cutoff * t)
    return f / f.sum()


def snake_aa(w, name: str, x: torch.Tensor) -> torch.Tensor:
    """Xcodec2AntiAliasedActivation1d(SnakeBeta)  :524-545 on x [1, C, T]: 2x up-sample (replicate pad 5, transposed conv with
    the 12-tap filter x 2, crop 15 / 15 :497-521) -> x + sin^2(x e^alpha) / (e^beta + 1e-9)  (:400-413) -> replicate pad (5, 6),
    12-tap filter, stride 2 (:463-494)."""
    C = x.shape[1]
    f = kaiser_sinc_filter(0.25, 0.3, 12).view(1, 1, 12).expand(C, -1, -1)
    h = F.pad(x, (5, 5), mode="replicate")
    h = 2 * F.conv_transpose1d(h, f, stride=2, groups=C)[..., 15:-15]
    a = torch.exp(w[name + ".act.alpha"]).view(1, -1, 1)
    b = torch.exp(w[name + ".act.beta"]).view(1, -1, 1)
    h = h + (1.0 / (b + 1e-9)) * torch.sin(h * a) ** 2
    h = F.pad(h, (5, 6), mode="replicate")
    return F.conv1d(h, f, stride=2, groups=C)


def acoustic_encoder(cfg: EncoderConfig, w, wav: torch.Tensor) -> torch.Tensor:
    """Xcodec2Encoder.forward  :627-636 (+ EncoderBlock :598-604, ResidualUnit :561-581).  wav [L] -> [T, codec_hidden]."""
    a = "acoustic_encoder."
    h = F.conv1d(wav.view(1, 1, -1), w[a + "conv1.weight"], w[a + "conv1.bias"], padding=3)
    for bi, stride in enumerate(cfg.ratios):
        b = f"{a}block.{bi}."
        for u, dil in ((1, 1), (2, 3), (3, 9)):
            r = f"{b}res_unit{u}."
            y = F.conv1d(snake_aa(w, r + "snake1", h), w[r + "conv1.weight"], w[r + "conv1.bias"], dilation=dil, padding=3 * dil)
            y = F.conv1d(snake_aa(w, r + "snake2", y), w[r + "conv2.weight"], w[r + "conv2.bias"])
            h = h + y
        h = F.conv1d(snake_aa(w, b + "snake1", h), w[b + "conv1.weight"], w[b + "conv1.bias"], stride=stride,
                     padding=math.ceil(stride / 2))
    h = F.conv1d(snake_aa(w, a + "snake1", h), w[a + "conv2.weight"], w[a + "conv2.bias"], padding=1)
    return h[0].t()


# ---------------------------------------------------------------------------------------------------- quantiser
def fsq_bound(cfg: EncoderConfig, z: torch.Tensor, eps: float = 1e-3) -> torch.Tensor:
    """Xcodec2FiniteScalarQuantization.bound  :703-722."""
    levels = torch.tensor(cfg.levels, dtype=torch.int32)
    half_range = (levels - 1) * (1 + eps) / 2
    offset = torch.where(levels % 2 == 0, 0.5, 0.0)
    shift = (offset / half_range).atanh()
    return (z + shift).tanh() * half_range - offset


def fsq_indices(cfg: EncoderConfig, w, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Xcodec2Quantizer.forward :811-818 + FSQ.forward :724-743: project_in -> bound TWICE (the quantizer bounds, then the FSQ
    module bounds again) -> round -> sum(digit * basis).  Returns (codes int32 [T], the twice-bounded latents [T, n])."""
    z = F.linear(x, w["quantizer.project_in.weight"], w["quantizer.project_in.bias"]).float()
    z = fsq_bound(cfg, fsq_bound(cfg, z))
    levels = torch.tensor(cfg.levels, dtype=torch.int32)
    basis = torch.cumprod(torch.tensor([1] + list(cfg.levels[:-1])), dim=0, dtype=torch.int32)
    half = levels // 2
    codes = ((z.round() / half * half + half) * basis).sum(-1).to(torch.int32)
    return codes, z


@torch.no_grad()
def encode(cfg: EncoderConfig, w: Dict[str, torch.Tensor], wav: np.ndarray, return_parts: bool = False):
    """Xcodec2Model.encode  :974-1049 (== neucodec `encode_code`): wav float32 [L] at 16 kHz -> int32 codes [L // hop + 1]."""
    wp = pad_audio(wav, cfg.hop)
    feats = torch.from_numpy(fbank_features(wp))
    sem = semantic_adapter(w, semantic_encoder(cfg, w, feats))
    ac = acoustic_encoder(cfg, w, torch.from_numpy(wp))
    cat = torch.cat([sem, ac], dim=-1)
    h = F.linear(cat, w["fc_encoder.weight"], w["fc_encoder.bias"])
    codes, z = fsq_indices(cfg, w, h)
    if return_parts:
        return codes.numpy(), {"features": feats.numpy(), "semantic": sem.numpy(), "acoustic": ac.numpy(), "fc": h.numpy(), "latents": z.numpy()}
    return codes.numpy()


# ---------------------------------------------------------------------------------------------------- live HF model (pinning)
def hf_model(cfg: EncoderConfig, w: Dict[str, torch.Tensor]):
    """The live transformers Xcodec2Model at this geometry with the encoder weights `w` loaded (decoder left at its init)."""
    from transformers import Xcodec2Config, Xcodec2Model
    sem = dict(hidden_size=cfg.sem_hidden, num_hidden_layers=cfg.sem_layers, num_attention_heads=cfg.sem_heads,
               intermediate_size=cfg.sem_ffn, feature_projection_input_dim=cfg.sem_feat_dim,
               conv_depthwise_kernel_size=cfg.sem_conv_kernel, left_max_position_embeddings=cfg.sem_left,
               right_max_position_embeddings=cfg.sem_right, layer_norm_eps=cfg.sem_ln_eps, position_embeddings_type="relative_key")
    hc = Xcodec2Config(hidden_size=cfg.codec_hidden, intermediate_size=2 * cfg.codec_hidden, num_hidden_layers=1,
                       num_attention_heads=max(1, cfg.codec_hidden // 64), num_key_value_heads=max(1, cfg.codec_hidden // 64),
                       head_dim=64, encoder_hidden_size=cfg.ac_hidden, downsampling_ratios=list(cfg.ratios),
                       semantic_model_config=sem, quantization_dim=cfg.cat_dim, quantization_levels=list(cfg.levels))
    m = Xcodec2Model(hc).eval()
    missing, unexpected = m.load_state_dict(w, strict=False)
    assert not unexpected, unexpected
    enc_missing = [k for k in missing if not k.startswith(("acoustic_decoder.", "quantizer.project_out.")) and "masked_spec_embed" not in k]
    assert not enc_missing, enc_missing
    return m
