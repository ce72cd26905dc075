"""TEST INFRASTRUCTURE — CPU oracle for the enrollment embedder (BASELINE configs[4]).

PARITY: FRONT END + HEAD PINNED, TRUNK BLOCKS UNPINNED.  oracle/check_embedder_against_reference.py runs the
reference's own `Stft` copy (src/models/tfgridnet_orig/stft.py:32-233) and the reference's own
`EmbedTFGridNet.forward` lines (tfgridnet.py:100-127) around a stub trunk and finds this file's `spec` tap bit-equal
and its embedding equal to 4e-16 (fp64); fixtures in tests/golden/embedder_pinned_golden.npz.  What stays unpinned is
the inside of the trunk blocks:  `EmbedTFGridNet` (reference src/models/tfgridnet_orig/tfgridnet.py:88-127) subclasses
`espnet2.enh.separator.tfgridnet_separator.TFGridNet`, an un-vendored, un-pinned third-party dependency
(`requirements.txt:19` lists `espnet` without a version) that is not installed here and whose source is not under
/root/reference.  The trunk below (STFT encoder, Conv2d+GroupNorm, non-causal GridNetBlock with emb_ks=4 unfold
BiLSTMs, per-head 1x1-conv Q/K/V with (C,F) LayerNorm, full T x T attention) is therefore a restatement of the
published espnet2 algorithm (TF-GridNet, Wang et al. 2022; espnet2 tfgridnet_separator.py as of espnet 202301..202402)
from its documented structure; only the head (std-normalisation, channel stacking, Linear(65*64 -> 256) + LayerNorm,
mean over frames) follows reference lines that exist in the tree.  No reference output exists to pin it against:
confirm the parameter manifest against a real `runs/embed` checkpoint's keys before trusting it (SURVEY.md
Appendix C).  The separator oracle (oracle/tfgridnet_oracle.py) is unaffected by this caveat.

Layout inside: espnet2's own `[B, C, T, F]`.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as TF

EMBED_PARAMS = dict(embed_dim=256, num_ch=2, n_fft=128, stride=64, num_blocks=3)    # configs/embed.json:5-11


class ECfg:
    def __init__(self, embed_dim=256, num_ch=2, n_fft=128, stride=64, num_blocks=3):
        self.embed_dim, self.M, self.nfft, self.hop, self.nblk = embed_dim, num_ch, n_fft, stride, num_blocks
        self.F = n_fft // 2 + 1           # 65
        self.C = 64                       # emb_dim (tfgridnet.py:92)
        self.H = 64                       # lstm_hidden_units
        self.nh = 4                       # attn_n_head default
        self.E = math.ceil(512 / self.F)  # 8
        self.Vd = self.C // self.nh       # 16
        self.ks, self.hs = 4, 1           # emb_ks / emb_hs defaults
        self.eps = 1e-5


def param_manifest(cfg: ECfg) -> Dict[str, tuple]:
    """state-dict names/shapes of EmbedTFGridNet as espnet2 would register them (n_srcs = 1)."""
    C, F, H, nh, E, Vd, ks = cfg.C, cfg.F, cfg.H, cfg.nh, cfg.E, cfg.Vd, cfg.ks
    m = {"conv.0.weight": (C, 2 * cfg.M, 3, 3), "conv.0.bias": (C,), "conv.1.weight": (C,), "conv.1.bias": (C,)}
    for i in range(cfg.nblk):
        p = f"blocks.{i}."
        for ax in ("intra", "inter"):
            m[p + f"{ax}_norm.gamma"] = (1, C, 1, 1)
            m[p + f"{ax}_norm.beta"] = (1, C, 1, 1)
            for sfx in ("", "_reverse"):
                m[p + f"{ax}_rnn.weight_ih_l0{sfx}"] = (4 * H, C * ks)
                m[p + f"{ax}_rnn.weight_hh_l0{sfx}"] = (4 * H, H)
                m[p + f"{ax}_rnn.bias_ih_l0{sfx}"] = (4 * H,)
                m[p + f"{ax}_rnn.bias_hh_l0{sfx}"] = (4 * H,)
            m[p + f"{ax}_linear.weight"] = (2 * H, C, ks)          # ConvTranspose1d weight [in, out, k]
            m[p + f"{ax}_linear.bias"] = (C,)
        for h in range(nh):
            for nm, d in (("Q", E), ("K", E), ("V", Vd)):
                q = p + f"attn_conv_{nm}_{h}."
                m[q + "0.weight"] = (d, C, 1, 1)
                m[q + "0.bias"] = (d,)
                m[q + "1.weight"] = (1,)
                m[q + "2.gamma"] = (1, d, 1, F)
                m[q + "2.beta"] = (1, d, 1, F)
        q = p + "attn_concat_proj."
        m[q + "0.weight"] = (C, C, 1, 1)
        m[q + "0.bias"] = (C,)
        m[q + "1.weight"] = (1,)
        m[q + "2.gamma"] = (1, C, 1, F)
        m[q + "2.beta"] = (1, C, 1, F)
    m["deconv.weight"] = (C, 2, 3, 3)      # registered by the espnet2 trunk, unused by EmbedTFGridNet.forward
    m["deconv.bias"] = (2,)
    m["embed_proj.0.weight"] = (cfg.embed_dim, F * C)
    m["embed_proj.0.bias"] = (cfg.embed_dim,)
    m["embed_proj.1.weight"] = (cfg.embed_dim,)
    m["embed_proj.1.bias"] = (cfg.embed_dim,)
    return m


def synthetic_state_dict(cfg: ECfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    sd = {}
    for idx, (name, shape) in enumerate(sorted(param_manifest(cfg).items())):
        g = torch.Generator().manual_seed(seed * 100003 + 7919 + idx)
        u = torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1
        if name.endswith(".gamma") or name in ("conv.1.weight", "embed_proj.1.weight"):
            t = 1.0 + 0.25 * u
        elif name.endswith(".beta") or name in ("conv.1.bias", "embed_proj.1.bias"):
            t = 0.1 * u
        elif name.endswith(".1.weight") and shape == (1,):
            t = 0.25 + 0.1 * u
        elif len(shape) == 1:
            t = u / 8.0
        else:
            fan_in = shape[1] * (shape[2] if len(shape) > 2 else 1) * (shape[3] if len(shape) > 3 else 1)
            if name.endswith("_linear.weight"):
                fan_in = shape[0] * shape[2]
            t = u / math.sqrt(fan_in)
        sd[name] = t.float()
    return sd


def _ln4d(x, g, b, eps):            # over C of [B,C,T,F]
    mu = x.mean(1, keepdim=True)
    var = x.var(1, unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def _ln4dcf(x, g, b, eps):          # over (C,F) of [B,C,T,F]
    mu = x.mean((1, 3), keepdim=True)
    var = x.var((1, 3), unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def _prelu(x, a):
    return torch.where(x >= 0, x, a * x)


def _bilstm(x, p, pre):
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    flat = [p[pre + n] for n in names] + [p[pre + n + "_reverse"] for n in names]
    h0 = x.new_zeros(2, x.shape[0], flat[1].shape[1])
    out, _, _ = torch._VF.lstm(x, (h0, h0.clone()), flat, True, 1, 0.0, False, True, True)
    return out


def _axis_path(cfg, p, pre, ax, x):
    """LN over C -> unfold(ks) along the last axis -> BiLSTM -> ConvTranspose1d(ks) -> + residual.
    x [N, C, L] where L is the scanned axis (frequency for intra, time for inter)."""
    N, C, L = x.shape
    u = TF.unfold(x[..., None], (cfg.ks, 1), stride=(cfg.hs, 1))           # [N, C*ks, L-ks+1], feature = c*ks + k
    h = _bilstm(u.transpose(1, 2), p, pre + f"{ax}_rnn.")                  # [N, L', 2H]
    return TF.conv_transpose1d(h.transpose(1, 2), p[pre + f"{ax}_linear.weight"], p[pre + f"{ax}_linear.bias"],
                               stride=cfg.hs)                              # [N, C, L]


def block(cfg: ECfg, p: dict, pre: str, x, taps=None):
    B, C, T, Q = x.shape
    assert (T - cfg.ks) % cfg.hs == 0 and (Q - cfg.ks) % cfg.hs == 0       # hs = 1: no padding branch
    y = _ln4d(x, p[pre + "intra_norm.gamma"], p[pre + "intra_norm.beta"], cfg.eps)
    y = _axis_path(cfg, p, pre, "intra", y.transpose(1, 2).reshape(B * T, C, Q)).view(B, T, C, Q).transpose(1, 2)
    x1 = y + x
    if taps is not None:
        taps[pre + 'x1'] = x1.permute(0, 2, 3, 1)
    y = _ln4d(x1, p[pre + "inter_norm.gamma"], p[pre + "inter_norm.beta"], cfg.eps)
    y = _axis_path(cfg, p, pre, "inter", y.permute(0, 3, 1, 2).reshape(B * Q, C, T)).view(B, Q, C, T).permute(0, 2, 3, 1)
    x2 = y + x1
    if taps is not None:
        taps[pre + 'x2'] = x2.permute(0, 2, 3, 1)

    def head(nm, h):
        q = pre + f"attn_conv_{nm}_{h}."
        z = TF.conv2d(x2, p[q + "0.weight"], p[q + "0.bias"])
        return _ln4dcf(_prelu(z, p[q + "1.weight"]), p[q + "2.gamma"], p[q + "2.beta"], cfg.eps)

    Qh = torch.cat([head("Q", h) for h in range(cfg.nh)], 0)               # [nh*B, E, T, F] (head-major)
    Kh = torch.cat([head("K", h) for h in range(cfg.nh)], 0)
    Vh = torch.cat([head("V", h) for h in range(cfg.nh)], 0)               # [nh*B, Vd, T, F]
    Qf = Qh.transpose(1, 2).flatten(2)                                     # [nh*B, T, E*F]  (e-major, f-minor)
    Kf = Kh.transpose(1, 2).flatten(2)
    Vt = Vh.transpose(1, 2)
    att = torch.softmax(Qf @ Kf.transpose(1, 2) / math.sqrt(Qf.shape[-1]), dim=2)     # full T x T, no mask
    O = (att @ Vt.flatten(2)).reshape(Vt.shape).transpose(1, 2)            # [nh*B, Vd, T, F]
    O = O.view(cfg.nh, B, cfg.Vd, T, Q).transpose(0, 1).reshape(B, cfg.nh * cfg.Vd, T, Q)
    if taps is not None:
        taps[pre + 'Q'], taps[pre + 'K'], taps[pre + 'V'] = Qh, Kh, Vh        # [nh*B, d, T, F], head-major batch
        taps[pre + 'O'] = O.permute(0, 2, 3, 1)                               # [B, T, F, 64]
    q = pre + "attn_concat_proj."
    z = _prelu(TF.conv2d(O, p[q + "0.weight"], p[q + "0.bias"]), p[q + "1.weight"])
    return _ln4dcf(z, p[q + "2.gamma"], p[q + "2.beta"], cfg.eps) + x2


def forward(cfg: ECfg, sd: dict, x, dtype=torch.float32, taps=None):
    """EmbedTFGridNet.forward (tfgridnet_orig/tfgridnet.py:100-127): x [B, M, N] -> [B, embed_dim]."""
    p = {k: v.detach().to("cpu", dtype) for k, v in sd.items()}
    x = x.detach().to("cpu", dtype).transpose(1, 2)                        # [B, N, M]
    x = x / torch.std(x, dim=(1, 2), keepdim=True)                         # unbiased std, :109-110
    B, N, M = x.shape
    win = torch.hann_window(cfg.nfft, dtype=dtype)                         # espnet2 Stft: hann, center, reflect pad
    spec = torch.stft(x.transpose(1, 2).reshape(B * M, N), cfg.nfft, cfg.hop, cfg.nfft, win, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)   # [B*M, F, T]
    spec = spec.view(B, M, cfg.F, -1).permute(0, 1, 3, 2)                  # [B, M, T, F]
    z = torch.cat([spec.real, spec.imag], dim=1)                           # [B, 2M, T, F]
    z = TF.conv2d(z, p["conv.0.weight"], p["conv.0.bias"], padding=(1, 1))
    if taps is not None:
        taps['spec'] = torch.cat([spec.real, spec.imag], dim=1)
        taps['zraw'] = z.permute(0, 2, 3, 1)
    z = TF.group_norm(z, 1, p["conv.1.weight"], p["conv.1.bias"], cfg.eps)
    if taps is not None:
        taps['z0'] = z.permute(0, 2, 3, 1)
    for i in range(cfg.nblk):
        z = block(cfg, p, f"blocks.{i}.", z, taps)
        if taps is not None:
            taps[f'blocks.{i}.out'] = z.permute(0, 2, 3, 1)
    T = z.shape[2]
    e = z.permute(0, 2, 1, 3).reshape(B, T, cfg.C * cfg.F)                 # [B, T, C*F] (c-major)
    e = e @ p["embed_proj.0.weight"].t() + p["embed_proj.0.bias"]
    e = TF.layer_norm(e, (cfg.embed_dim,), p["embed_proj.1.weight"], p["embed_proj.1.bias"], cfg.eps)
    return e.mean(1)
