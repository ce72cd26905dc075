"""TEST INFRASTRUCTURE — CPU oracle for the LookOnceToHear separator forward path.

This file is the checker, never the product: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.

It is a from-scratch *restatement* (index-notation level)
of the reference forward path
    Net.forward / predict / mod_pad      reference src/models/tfgridnet_realtime/net.py:8-76
    TFGridNet.forward                    reference .../tfgridnet_causal.py:188-283
    GridNetBlock.forward                 reference .../tfgridnet_causal.py:489-590
in plain PyTorch-CPU tensor algebra (fp32 or fp64).  It is self-contained so it can
travel to the GPU box, where `/root/reference` does not exist.

Pinning: the reference ships no tests, golden vectors or checkpoints for this path
(SURVEY.md §4, §8c), so the oracle is pinned against *outputs of the reference code
itself*, imported unmodified in the build container under the stubs of
`oracle/ref_stubs.py` (`oracle/check_against_reference.py`, and the committed fixtures
in `tests/golden/` written by `oracle/gen_golden.py`).  The STFT filterbank is an
un-vendored third-party dependency (asteroid-filterbanks, unpinned); its arithmetic is
restated in `stft_filters()` and is additionally treated as *data* (state-dict buffer
`enc.filterbank._filters`), so a real checkpoint overrides it.

All activations inside the oracle are channel-last `[B, T, F, C]`.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as TF

# ----------------------------------------------------------------------------------------
# configuration (configs/tsh.json:5-19 -> Net.__init__ net.py:21-49)
# ----------------------------------------------------------------------------------------


class Cfg:
    """Shape constants derived from the reference `model_params`."""

    def __init__(self, stft_chunk_size=128, stft_pad_size=64, embed_dim=256, num_ch=2, D=64, B=3,
                 I=1, J=1, L=4, H=64, use_attn=True, lookahead=True, local_atten_len=50,
                 chunk_causal=True, num_src=2):
        assert use_attn and chunk_causal, "only the configs/tsh.json operating mode is restated"
        self.hop = stft_chunk_size
        self.pad = stft_pad_size
        self.nfft = stft_chunk_size + stft_pad_size
        self.F = self.nfft // 2 + 1
        self.M = num_ch
        self.C = D
        self.nblk = B
        self.nh = L
        self.H = H
        self.E = math.ceil(512 * 1.0 / self.F)      # tfgridnet_causal.py:320-322 (approx_qk_dim=512)
        self.Vd = D // L                            # :324
        self.L = local_atten_len
        self.S = num_src
        self.embed_dim = embed_dim
        self.lookahead = lookahead
        self.eps = 1e-5


TSH_PARAMS = dict(embed_dim=256, stft_chunk_size=128, stft_pad_size=64, num_ch=2, D=64, L=4, I=1, J=1,
                  B=3, H=64, local_atten_len=50, use_attn=True, lookahead=True, chunk_causal=True)


def stft_filters(nfft: int, hop: int) -> torch.Tensor:
    """asteroid-filterbanks `STFTFB(n_filters=nfft, kernel_size=nfft, stride=hop)` rows, [nfft+2, nfft].

    Row k<F: window[n]*cos(2 pi k n/nfft)/scale ; row F+k: -window[n]*sin(2 pi k n/nfft)/scale ;
    window = sqrt(periodic Hann); scale = 0.5*sqrt(nfft*nfft/hop); DC and Nyquist cosine rows / sqrt(2).
    Used by the reference at tfgridnet_causal.py:131-135 (third-party arithmetic).
    """
    F = nfft // 2 + 1
    n = np.arange(nfft)
    win = np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * n / nfft))
    k = np.arange(F)[:, None]
    ang = 2 * np.pi * k * n[None, :] / nfft
    re = np.cos(ang)
    im = -np.sin(ang)
    re[0] /= np.sqrt(2)
    re[nfft // 2] /= np.sqrt(2)
    scale = 0.5 * np.sqrt(nfft * nfft / hop)
    return torch.from_numpy(np.vstack([re, im]) * win[None, :] / scale).float()


# ----------------------------------------------------------------------------------------
# parameter manifest + deterministic synthetic weights (no checkpoint ships with the reference)
# ----------------------------------------------------------------------------------------


def param_manifest(cfg: Cfg) -> Dict[str, tuple]:
    """state-dict names/shapes of `Net` (SURVEY.md §8b checkpoint surface), prefix `tfgridnet.`."""
    C, F, H, nh, E, Vd = cfg.C, cfg.F, cfg.H, cfg.nh, cfg.E, cfg.Vd
    m = {
        "enc.filterbank._filters": (cfg.nfft + 2, 1, cfg.nfft),
        "dec.filterbank._filters": (cfg.nfft + 2, 1, cfg.nfft),
        "conv.0.weight": (C, 2 * cfg.M, 3, 3), "conv.0.bias": (C,),
    }
    for i in range(cfg.nblk):
        p = f"blocks.{i}."
        m[p + "intra_norm.norm.weight"] = (C,)
        m[p + "intra_norm.norm.bias"] = (C,)
        for sfx in ("", "_reverse"):
            m[p + f"intra_rnn.weight_ih_l0{sfx}"] = (4 * H, C)
            m[p + f"intra_rnn.weight_hh_l0{sfx}"] = (4 * H, H)
            m[p + f"intra_rnn.bias_ih_l0{sfx}"] = (4 * H,)
            m[p + f"intra_rnn.bias_hh_l0{sfx}"] = (4 * H,)
        m[p + "intra_linear.weight"] = (C, 2 * H)
        m[p + "intra_linear.bias"] = (C,)
        m[p + "inter_norm.norm.weight"] = (C,)
        m[p + "inter_norm.norm.bias"] = (C,)
        m[p + "inter_rnn.weight_ih_l0"] = (4 * H, C)
        m[p + "inter_rnn.weight_hh_l0"] = (4 * H, H)
        m[p + "inter_rnn.bias_ih_l0"] = (4 * H,)
        m[p + "inter_rnn.bias_hh_l0"] = (4 * H,)
        m[p + "inter_linear.weight"] = (C, H)
        m[p + "inter_linear.bias"] = (C,)
        for nm, od, ld in (("Q", nh * E, F * E), ("K", nh * E, F * E), ("V", nh * Vd, F * Vd)):
            m[p + f"attn_conv_{nm}.0.weight"] = (od, C)
            m[p + f"attn_conv_{nm}.0.bias"] = (od,)
            m[p + f"attn_conv_{nm}.1.weight"] = (1,)
            m[p + f"attn_conv_{nm}.3.norm.weight"] = (ld,)
            m[p + f"attn_conv_{nm}.3.norm.bias"] = (ld,)
        m[p + "attn_concat_proj.0.weight"] = (C, C)
        m[p + "attn_concat_proj.0.bias"] = (C,)
        m[p + "attn_concat_proj.1.weight"] = (1,)
        m[p + "attn_concat_proj.3.norm.weight"] = (F * C,)
        m[p + "attn_concat_proj.3.norm.bias"] = (F * C,)
    m["embed_to_feats_proj.0.weight"] = (C * F, cfg.embed_dim)
    m["embed_to_feats_proj.0.bias"] = (C * F,)
    m["embed_to_feats_proj.1.weight"] = (C * F,)
    m["embed_to_feats_proj.1.bias"] = (C * F,)
    m["deconv.weight"] = (C, 2 * cfg.S, 3, 3)
    m["deconv.bias"] = (2 * cfg.S,)
    return {"tfgridnet." + k: v for k, v in m.items()}


def synthetic_state_dict(cfg: Cfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic, name-keyed random weights (reproducible on any box with this torch build).

    Scales follow torch's default initialisers (uniform(+-1/sqrt(fan_in))) so activations stay in the
    trained-model regime, but norm affines / PReLU slopes / biases are perturbed away from their
    1/0/0.25 defaults so that every learned tensor influences the output (a layout bug in an affine
    that is all-ones would otherwise be invisible).
    """
    sd = {}
    for idx, (name, shape) in enumerate(sorted(param_manifest(cfg).items())):
        g = torch.Generator().manual_seed(seed * 100003 + idx)
        if name.endswith("_filters"):
            sd[name] = stft_filters(cfg.nfft, cfg.hop).unsqueeze(1)
            continue
        u = torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1
        if ".norm.weight" in name or name.endswith("embed_to_feats_proj.1.weight"):
            t = 1.0 + 0.25 * u
        elif ".norm.bias" in name or name.endswith("embed_to_feats_proj.1.bias"):
            t = 0.1 * u
        elif name.endswith(".1.weight") and len(shape) == 1 and shape[0] == 1:
            t = 0.25 + 0.1 * u                      # PReLU slope
        else:
            if len(shape) == 1:                     # biases
                fan_in = 64
            elif "deconv.weight" in name:           # ConvTranspose2d sums over in_ch * k*k terms
                fan_in = shape[0] * shape[2] * shape[3]
            else:
                fan_in = int(np.prod(shape[1:]))
            t = u / math.sqrt(fan_in)
        sd[name] = t.float()
    return sd


# ----------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------


def _ln(x, w, b, eps):
    """LayerNorm over the last axis (biased variance), nn.LayerNorm semantics."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _prelu(x, a):
    return torch.where(x >= 0, x, a * x)


def lstm_scan(x, w_ih, w_hh, b_ih, b_hh, h0, c0, reverse=False):
    """Single-layer LSTM over axis 1 of x [N,S,I]; PyTorch gate order i,f,g,o.  Returns (hs [N,S,H], h, c)."""
    N, S, _ = x.shape
    H = w_hh.shape[1]
    gx = x @ w_ih.t() + (b_ih + b_hh)
    h, c = h0, c0
    out = x.new_empty(N, S, H)
    order = range(S - 1, -1, -1) if reverse else range(S)
    for s in order:
        g = gx[:, s] + h @ w_hh.t()
        i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[:, s] = h
    return out, h, c


def _lstm_fast(x, p, prefix, h0, c0, bidirectional):
    """Same arithmetic through torch's fused CPU LSTM (used for the timed cpu_baseline leg)."""
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    flat = [p[prefix + n] for n in names]
    if bidirectional:
        flat += [p[prefix + n + "_reverse"] for n in names]
    nd = 2 if bidirectional else 1
    if h0 is None:
        h0 = x.new_zeros(nd, x.shape[0], flat[1].shape[1])
        c0 = h0.clone()
    out, h, c = torch._VF.lstm(x, (h0, c0), flat, True, 1, 0.0, False, bidirectional, True)
    return out, h, c


def init_state(cfg: Cfg, batch: int, dtype=torch.float32) -> dict:
    """Zero state with the reference `init_buffers` shapes (tfgridnet_causal.py:173-186, 408-427)."""
    st = dict(conv_buf=torch.zeros(batch, 2 * cfg.M, 2, cfg.F, dtype=dtype),
              deconv_buf=torch.zeros(batch, cfg.C, 2, cfg.F, dtype=dtype),
              istft_buf=torch.zeros(batch, cfg.S, 2 * cfg.F, 1, dtype=dtype),
              gridnet_bufs={})
    for i in range(cfg.nblk):
        st["gridnet_bufs"][f"buf{i}"] = dict(
            K_buf=torch.zeros(batch * cfg.nh, cfg.L - 1, cfg.E * cfg.F, dtype=dtype),
            V_buf=torch.zeros(batch * cfg.nh, cfg.L - 1, cfg.Vd * cfg.F, dtype=dtype),
            c0=torch.zeros(1, batch * cfg.F, cfg.H, dtype=dtype),
            h0=torch.zeros(1, batch * cfg.F, cfg.H, dtype=dtype))
    return st


def front_end(cfg: Cfg, p: dict, x, conv_buf):
    """A.1: STFT analysis + re/im channel split + causal 3x3 conv.  x [B,M,N'] -> Z0 [B,T,F,C]."""
    B, M, N = x.shape
    Wfb = p["enc.filterbank._filters"][:, 0]                       # [2F, nfft]
    frames = x.unfold(-1, cfg.nfft, cfg.hop)                       # [B,M,T,nfft]
    spec = frames @ Wfb.t()                                        # [B,M,T,2F]
    Fq = cfg.F
    xin = torch.cat([spec[..., :Fq], spec[..., Fq:]], dim=1)       # [B,2M,T,F]: re_m0,re_m1,im_m0,im_m1
    xbuf = torch.cat([conv_buf, xin], dim=2)                       # [B,2M,T+2,F]
    new_buf = xbuf[:, :, -2:, :].clone()
    z = TF.conv2d(xbuf, p["conv.0.weight"], p["conv.0.bias"], padding=(0, 1))   # [B,C,T,F]
    return z.permute(0, 2, 3, 1).contiguous(), new_buf, xin


def speaker_gain(cfg: Cfg, p: dict, emb):
    """A.2: LayerNorm(Linear(emb)) reshaped [B,C,F] (C-major) -> returned as [B,1,F,C]."""
    g = emb @ p["embed_to_feats_proj.0.weight"].t() + p["embed_to_feats_proj.0.bias"]
    g = _ln(g, p["embed_to_feats_proj.1.weight"], p["embed_to_feats_proj.1.bias"], cfg.eps)
    return g.reshape(-1, cfg.C, cfg.F).permute(0, 2, 1).unsqueeze(1).contiguous()


def gridnet_block(cfg: Cfg, p: dict, pre: str, X, st: dict, fast_lstm=False, taps=None):
    """A.3: one causal GridNet block.  X [B,T,F,C] -> out [B,T,F,C]; mutates st (K_buf,V_buf,h0,c0)."""
    B, T, Fq, C = X.shape
    H, nh, E, Vd, L = cfg.H, cfg.nh, cfg.E, cfg.Vd, cfg.L

    # 1. intra (full-band) BiLSTM over frequency, zero initial state
    U = _ln(X, p[pre + "intra_norm.norm.weight"], p[pre + "intra_norm.norm.bias"], cfg.eps).reshape(B * T, Fq, C)
    if fast_lstm:
        hs, _, _ = _lstm_fast(U, p, pre + "intra_rnn.", None, None, True)
    else:
        z = U.new_zeros(B * T, H)
        hf, _, _ = lstm_scan(U, p[pre + "intra_rnn.weight_ih_l0"], p[pre + "intra_rnn.weight_hh_l0"],
                             p[pre + "intra_rnn.bias_ih_l0"], p[pre + "intra_rnn.bias_hh_l0"], z, z)
        hb, _, _ = lstm_scan(U, p[pre + "intra_rnn.weight_ih_l0_reverse"], p[pre + "intra_rnn.weight_hh_l0_reverse"],
                             p[pre + "intra_rnn.bias_ih_l0_reverse"], p[pre + "intra_rnn.bias_hh_l0_reverse"], z, z,
                             reverse=True)
        hs = torch.cat([hf, hb], -1)
    Y1 = (hs @ p[pre + "intra_linear.weight"].t() + p[pre + "intra_linear.bias"]).reshape(B, T, Fq, C) + X
    if taps is not None:
        taps[pre + "Y1"] = Y1

    # 2. inter (sub-band) causal LSTM over time with carried state, sequence index b*F+f
    Vn = _ln(Y1, p[pre + "inter_norm.norm.weight"], p[pre + "inter_norm.norm.bias"], cfg.eps)
    Vn = Vn.transpose(1, 2).reshape(B * Fq, T, C)
    if fast_lstm:
        hs, h, c = _lstm_fast(Vn, p, pre + "inter_rnn.", st["h0"], st["c0"], False)
        st["h0"], st["c0"] = h, c
    else:
        hs, h, c = lstm_scan(Vn, p[pre + "inter_rnn.weight_ih_l0"], p[pre + "inter_rnn.weight_hh_l0"],
                             p[pre + "inter_rnn.bias_ih_l0"], p[pre + "inter_rnn.bias_hh_l0"],
                             st["h0"][0], st["c0"][0])
        st["h0"], st["c0"] = h.unsqueeze(0), c.unsqueeze(0)
    Y2 = (hs @ p[pre + "inter_linear.weight"].t() + p[pre + "inter_linear.bias"]).reshape(B, Fq, T, C)
    Y2 = Y2.transpose(1, 2) + Y1
    if taps is not None:
        taps[pre + "Y2"] = Y2

    # 3. Q/K/V: pointwise Linear + PReLU, head split, joint LayerNorm over (f, e)
    def proj(nm, d):
        y = _prelu(Y2 @ p[pre + f"attn_conv_{nm}.0.weight"].t() + p[pre + f"attn_conv_{nm}.0.bias"],
                   p[pre + f"attn_conv_{nm}.1.weight"])                       # [B,T,F,nh*d]
        y = y.reshape(B, T, Fq, nh, d).permute(0, 3, 1, 2, 4).reshape(B * nh, T, Fq * d)
        return _ln(y, p[pre + f"attn_conv_{nm}.3.norm.weight"], p[pre + f"attn_conv_{nm}.3.norm.bias"], cfg.eps)

    Q, K, V = proj("Q", E), proj("K", E), proj("V", Vd)

    # 4. history rings
    Kx = torch.cat([st["K_buf"], K], 1)                                          # [B*nh, T+L-1, F*E]
    Vx = torch.cat([st["V_buf"], V], 1)
    st["K_buf"] = Kx[:, -(L - 1):].clone()
    st["V_buf"] = Vx[:, -(L - 1):].clone()

    # 5. local attention over exactly L slots (frames t-L+1..t), NO mask: zero history rows take part
    scale = 1.0 / math.sqrt(Fq * E)
    sc = torch.stack([(Q * Kx[:, j:j + T]).sum(-1) for j in range(L)], dim=-1) * scale   # [B*nh,T,L]
    pr = torch.softmax(sc, dim=-1)
    O = torch.zeros_like(V)
    for j in range(L):
        O = O + pr[:, :, j:j + 1] * Vx[:, j:j + T]
    if taps is not None:
        taps[pre + "Q"], taps[pre + "K"], taps[pre + "V"], taps[pre + "O"] = Q, K, V, O

    # 6. head merge, projection, joint LayerNorm over (f, c), residual
    Mg = O.reshape(B, nh, T, Fq, Vd).permute(0, 2, 3, 1, 4).reshape(B, T, Fq, nh * Vd)
    P = _prelu(Mg @ p[pre + "attn_concat_proj.0.weight"].t() + p[pre + "attn_concat_proj.0.bias"],
               p[pre + "attn_concat_proj.1.weight"]).reshape(B, T, Fq * C)
    P = _ln(P, p[pre + "attn_concat_proj.3.norm.weight"], p[pre + "attn_concat_proj.3.norm.bias"], cfg.eps)
    return Y2 + P.reshape(B, T, Fq, C)


def back_end(cfg: Cfg, p: dict, Y, deconv_buf, istft_buf):
    """A.4: causal transposed 3x3 conv + spectrum re-pack + iSTFT synthesis/overlap-add.  Y [B,T,F,C]."""
    B, T, Fq, C = Y.shape
    ybuf = torch.cat([deconv_buf, Y.permute(0, 3, 1, 2)], dim=2)               # [B,C,T+2,F]
    new_dbuf = ybuf[:, :, -2:, :].clone()
    D = TF.conv_transpose2d(ybuf, p["deconv.weight"], p["deconv.bias"], padding=(2, 1))   # [B,2S,T,F]
    D = D.reshape(B, cfg.S, 2, T, Fq)
    So = torch.cat([D[:, :, 0], D[:, :, 1]], dim=-1)                           # [B,S,T,2F] (re | im)
    Sx = torch.cat([istft_buf[..., 0].unsqueeze(2), So], dim=2)                # [B,S,T+1,2F]
    new_ibuf = Sx[:, :, -1, :].unsqueeze(-1).clone()                           # [B,S,2F,1]
    Wd = p["dec.filterbank._filters"][:, 0]                                    # [2F, nfft]
    fr = Sx @ Wd                                                               # [B,S,T+1,nfft]
    out = Y.new_zeros(B, cfg.S, T * cfg.hop + cfg.nfft)
    for t in range(T + 1):
        out[:, :, t * cfg.hop:t * cfg.hop + cfg.nfft] += fr[:, :, t]
    return out[:, :, cfg.hop:], new_dbuf, new_ibuf                             # drop first hop samples


# ----------------------------------------------------------------------------------------
# entry points mirroring Net.predict / Net.forward
# ----------------------------------------------------------------------------------------


def strip_prefix(sd: dict, dtype=torch.float32) -> dict:
    out = {}
    for k, v in sd.items():
        k2 = k
        for pre in ("model.tfgridnet.", "tfgridnet."):
            if k2.startswith(pre):
                k2 = k2[len(pre):]
                break
        out[k2] = v.detach().to("cpu", dtype)
    return out


def predict(cfg: Cfg, sd: dict, x, embed, state: Optional[dict], pad=True, dtype=torch.float32,
            fast_lstm=False, taps: Optional[dict] = None):
    """Net.predict (net.py:54-66): x [B,M,N], embed [B,E] -> (y [B,S,N], next_state)."""
    p = strip_prefix(sd, dtype)
    x = x.detach().to("cpu", dtype)
    embed = embed.detach().to("cpu", dtype)
    if state is None:
        state = init_state(cfg, x.shape[0], dtype)
    mod = 0
    if pad:
        if x.shape[-1] % cfg.hop:
            mod = cfg.hop - x.shape[-1] % cfg.hop
        x = TF.pad(x, (0, mod))
        if cfg.lookahead:
            x = TF.pad(x, (0, cfg.pad))
    Z, state["conv_buf"], xin = front_end(cfg, p, x, state["conv_buf"])
    if taps is not None:
        taps["spec"], taps["Z0"] = xin, Z
    G = speaker_gain(cfg, p, embed)
    if taps is not None:
        taps["G"] = G
    for i in range(cfg.nblk):
        if i == 1:
            Z = Z * G
        Z = gridnet_block(cfg, p, f"blocks.{i}.", Z, state["gridnet_bufs"][f"buf{i}"], fast_lstm, taps)
        if taps is not None:
            taps[f"blocks.{i}.out"] = Z
    y, state["deconv_buf"], state["istft_buf"] = back_end(cfg, p, Z, state["deconv_buf"], state["istft_buf"])
    y = y[..., :-cfg.pad]
    if mod:
        y = y[..., :-mod]
    return y, state


def forward(cfg: Cfg, sd: dict, x, embeds, input_state=None, pad=True, dtype=torch.float32, fast_lstm=False,
            taps=None):
    """Net.forward (net.py:68-76): embeds [B,1,E]."""
    y, _ = predict(cfg, sd, x, embeds[:, 0], input_state, pad, dtype, fast_lstm, taps)
    return y


# ----------------------------------------------------------------------------------------
# metrics restated from torchmetrics (absent here) as used by src/ts_hear_test.py:144-146
# ----------------------------------------------------------------------------------------


def si_snr(pred, target):
    """scale_invariant_signal_noise_ratio = zero-mean SI-SDR over the last axis."""
    eps = torch.finfo(pred.dtype).eps
    pred = pred - pred.mean(-1, keepdim=True)
    target = target - target.mean(-1, keepdim=True)
    alpha = ((pred * target).sum(-1, keepdim=True) + eps) / ((target ** 2).sum(-1, keepdim=True) + eps)
    ts = alpha * target
    noise = ts - pred
    return 10 * torch.log10(((ts ** 2).sum(-1) + eps) / ((noise ** 2).sum(-1) + eps))


def si_snr_i(outputs, mixture, target):
    """per-utterance mean over the 2 channels of si_snr(out,tgt) - si_snr(mix,tgt) (ts_hear_test.py:145-146)."""
    d = si_snr(outputs, target) - si_snr(mixture, target)
    return d.reshape(d.shape[0], -1).mean(1)


# ----------------------------------------------------------------------------------------
# helpers shared by the golden generator and the tests
# ----------------------------------------------------------------------------------------


def random_state(cfg: Cfg, batch: int, seed: int, dtype=torch.float32, scale: float = 0.5) -> dict:
    """Non-zero streaming state (every ring / tail / LSTM state populated), deterministic in `seed`."""
    g = torch.Generator().manual_seed(seed)
    st = init_state(cfg, batch, dtype)

    def fill(t):
        t.copy_((torch.randn(t.shape, generator=g, dtype=torch.float64) * scale).to(dtype))

    fill(st["conv_buf"]); fill(st["deconv_buf"]); fill(st["istft_buf"])
    for i in range(cfg.nblk):
        b = st["gridnet_bufs"][f"buf{i}"]
        for k in ("K_buf", "V_buf", "c0", "h0"):
            fill(b[k])
    return st


def clone_state(st: dict) -> dict:
    return {k: (clone_state(v) if isinstance(v, dict) else v.clone()) for k, v in st.items()}


def flat_state(st: dict, pre: str = "") -> dict:
    out = {}
    for k, v in st.items():
        if isinstance(v, dict):
            out.update(flat_state(v, pre + k + "."))
        else:
            out[pre + k] = v
    return out


def subsample(t: torch.Tensor, n: int = 512) -> torch.Tensor:
    """Fixed strided subsample used to keep golden fixtures small."""
    f = t.detach().reshape(-1)
    stride = max(1, f.numel() // n)
    return f[::stride][:n].clone()
