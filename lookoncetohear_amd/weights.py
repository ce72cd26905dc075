"""Packing of reference-named parameters (SURVEY.md §8b checkpoint surface) into the device images the
kernels expect.  Pure index shuffles (torch ops on whatever device the parameters live on), done once per
parameter version; the arithmetic stays in the HIP kernels.

MFMA B-operand image (v_mfma_f32_16x16x4_f32): for an output-column tile `nt` and k-step `ks`, lane `l`
holds  W[n = nt*16 + (l & 15)][k = (l >> 4) * (K/4) + ks]  — the k axis is split into four contiguous chunks,
one per 16-lane group, so the matching A operand is 32 (or 16) contiguous floats per lane in LDS.
"""
from __future__ import annotations

import torch

QK_PAD = 608          # Q / K feature rows padded 582 -> 608 (76 blocks of 8; lh_common.h DQKP)
KV_PAD_ROWS = 48      # zero rows behind the T+49 rows of kx / vx (include/lookonce_hip.h LH_KV_PAD_ROWS)


def unsplit_qk(rows: torch.Tensor, n: int = 582) -> torch.Tensor:
    """Split-precision q / kx rows [..., 1216] fp16 ([76 blocks][hi 8 | lo 8]) -> fp32 [..., n] (hi + lo)."""
    r = rows.reshape(*rows.shape[:-1], QK_PAD // 8, 2, 8).float()
    return (r[..., 0, :] + r[..., 1, :]).reshape(*rows.shape[:-1], QK_PAD)[..., :n]


def unsplit_v(rows: torch.Tensor) -> torch.Tensor:
    """Split-precision vx rows [..., 3104] fp16 ([388 quads][hi 4 | lo 4]) -> fp32 [..., 1552]."""
    r = rows.reshape(*rows.shape[:-1], rows.shape[-1] // 8, 2, 4).float()
    return (r[..., 0, :] + r[..., 1, :]).reshape(*rows.shape[:-1], rows.shape[-1] // 2)


def pack_linear(w: torch.Tensor) -> torch.Tensor:
    """w [N, K] (nn.Linear weight) -> [N/16, K/4, 64] fp32 B-operand image."""
    N, K = w.shape
    assert N % 16 == 0 and K % 16 == 0
    lane = torch.arange(64, device=w.device)
    n = torch.arange(N // 16, device=w.device)[:, None, None] * 16 + (lane & 15)[None, None, :]
    k = (lane >> 4)[None, None, :] * (K // 4) + torch.arange(K // 4, device=w.device)[None, :, None]
    return w[n, k].contiguous().float()


def pack_lstm(w_ih: torch.Tensor, w_hh: torch.Tensor) -> torch.Tensor:
    """w_ih [4H, I], w_hh [4H, H] (gate order i,f,g,o) -> [4 waves, 4 gates, 32 ksteps, 64 lanes].

    Wave `w` owns hidden units 16w..16w+15 of every gate: column tile (w, g) = rows g*H + 16w + (l & 15) of
    the concatenated [W_ih | W_hh]; k = (l >> 4) * 32 + ks over the 128 inputs [x(64) | h(64)].
    """
    H = w_hh.shape[1]
    assert H == 64 and tuple(w_ih.shape) == (4 * H, 64)
    wcat = torch.cat([w_ih, w_hh], dim=1)                       # [256, 128]
    lane = torch.arange(64, device=wcat.device)
    wave = torch.arange(4, device=wcat.device)[:, None, None, None]
    gate = torch.arange(4, device=wcat.device)[None, :, None, None]
    ks = torch.arange(32, device=wcat.device)[None, None, :, None]
    col = gate * H + wave * 16 + (lane & 15)[None, None, None, :]
    k = (lane >> 4)[None, None, None, :] * 32 + ks
    return wcat[col, k].contiguous().float()


SPLIT_SCALE = 2048.0      # 2^11: fp16 has 11 significant bits; lo = fp16((v - fp16(v)) * 2^11)


def split_f16(v: torch.Tensor):
    """fp32 -> (hi, lo) fp16 pair with v ~= hi + lo / 2^11 (about 22 significant bits)."""
    hi = v.float().half()
    lo = ((v.float() - hi.float()) * SPLIT_SCALE).half()
    return hi, lo


def split_f16_unscaled(v: torch.Tensor):
    """fp32 -> (hi, lo) fp16 pair with v ~= hi + lo, lo NOT rescaled (it is an fp16 subnormal for |v| < 2^-3; the
    matrix core takes subnormals at full value).  The recurrent kernels (lh_lstm.hip) use this form: all three partial
    products then go into one accumulator."""
    hi = v.float().half()
    lo = (v.float() - hi.float()).half()
    return hi, lo


LOG2E = 1.4426950408889634


def gate_prescale(n_hidden: int, device) -> torch.Tensor:
    """Per-row factor of [W_ih | W_hh] and of the summed bias in the split-precision recurrent kernels (lh_lstm.hip,
    lh_common.h lstm_cell_pre): sigma(a) = 1 / (1 + 2^(-log2e a)), tanh(a) = 2 / (1 + 2^(-2 log2e a)) - 1, so rows of the
    gates i, f, o carry -log2e and rows of g carry -2 log2e and the kernel feeds the accumulators straight to v_exp_f32."""
    f = torch.full((4, n_hidden), -LOG2E, dtype=torch.float64, device=device)
    f[2] *= 2.0
    return f.reshape(-1)


def pack_lstm_f16x3(w_ih: torch.Tensor, w_hh: torch.Tensor) -> torch.Tensor:
    """Split-precision image for v_mfma_f32_16x16x32_f16: [4 waves, 4 gates, 4 ksteps, 64 lanes, 2 (hi|lo), 8] fp16
    (lo unscaled, `split_f16_unscaled`).

    Lane l of (wave w, gate g, k-step ks) holds column g*H + 16w + (l & 15) of [W_ih | W_hh] at
    k = ks*32 + (l >> 4)*8 + j, j = 0..7 (x channels 0..63, then hidden units 0..63); rows pre-scaled by
    `gate_prescale` (the exponent scale of the gate non-linearities)."""
    H = w_hh.shape[1]
    assert H == 64 and tuple(w_ih.shape) == (4 * H, 64)
    wcat = torch.cat([w_ih, w_hh], dim=1).double()              # [256, 128]
    wcat = (wcat * gate_prescale(H, wcat.device)[:, None]).float()
    dev = wcat.device
    lane = torch.arange(64, device=dev)
    wave = torch.arange(4, device=dev)[:, None, None, None, None]
    gate = torch.arange(4, device=dev)[None, :, None, None, None]
    ks = torch.arange(4, device=dev)[None, None, :, None, None]
    j = torch.arange(8, device=dev)[None, None, None, None, :]
    col = gate * H + wave * 16 + (lane & 15)[None, None, None, :, None]
    k = ks * 32 + (lane >> 4)[None, None, None, :, None] * 8 + j
    frag = wcat[col, k]                                          # [4,4,4,64,8]
    hi, lo = split_f16_unscaled(frag)
    return torch.stack([hi, lo], dim=4).contiguous()             # [4,4,4,64,2,8]


def pack_lstm_f16x3_w8(w_ih: torch.Tensor, w_hh: torch.Tensor) -> torch.Tensor:
    """Image of the eight-wave fused kernel (lh_lstm.hip k_lstm_lin8p), where the weights are the MFMA **A** operand of a
    transposed gate GEMM: [8 waves, 2 tiles, 4 ksteps, 64 lanes, 2 (hi|lo), 8] fp16, lo unscaled, rows pre-scaled by
    `gate_prescale`.  Lane l of (wave v, tile m, k-step ks) holds row  gate*64 + unit  of [W_ih | W_hh] with
    gate = (l & 15) & 3, unit = 8v + 2 ((l & 15) >> 2) + m, at k = ks*32 + (l >> 4)*8 + j: the accumulator tile is then
    [16 rows = (unit, gate)] x [16 sequences], a lane's four registers are the four gates of one unit, and its two tiles
    hold ADJACENT units (round 4: h leaves the cell update as one packed 4-byte LDS store per half)."""
    H = w_hh.shape[1]
    assert H == 64 and tuple(w_ih.shape) == (4 * H, 64)
    wcat = torch.cat([w_ih, w_hh], dim=1).double()
    wcat = (wcat * gate_prescale(H, wcat.device)[:, None]).float()
    dev = wcat.device
    lane = torch.arange(64, device=dev)
    wave = torch.arange(8, device=dev)[:, None, None, None, None]
    tile = torch.arange(2, device=dev)[None, :, None, None, None]
    ks = torch.arange(4, device=dev)[None, None, :, None, None]
    j = torch.arange(8, device=dev)[None, None, None, None, :]
    rho = (lane & 15)[None, None, None, :, None]
    row = (rho & 3) * H + wave * 8 + 2 * (rho >> 2) + tile
    k = ks * 32 + (lane >> 4)[None, None, None, :, None] * 8 + j
    hi, lo = split_f16_unscaled(wcat[row, k])
    return torch.stack([hi, lo], dim=4).contiguous()             # [8,2,4,64,2,8]


def pack_linear_f16x3(w: torch.Tensor, unscaled: bool = False) -> torch.Tensor:
    """w [N, K] -> split-precision B image [N/16, K/32, 64 lanes, 2 (hi|lo), 8] fp16 for v_mfma_f32_16x16x32_f16:
    lane l of (n-tile nt, k-step ks) holds W[nt*16 + (l & 15)][ks*32 + (l >> 4)*8 + j], j = 0..7.
    `unscaled`: lo = fp16(w - hi) instead of fp16((w - hi) * 2^11) — the form the fused recurrent kernels take."""
    N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0
    dev = w.device
    lane = torch.arange(64, device=dev)
    nt = torch.arange(N // 16, device=dev)[:, None, None, None]
    ks = torch.arange(K // 32, device=dev)[None, :, None, None]
    j = torch.arange(8, device=dev)[None, None, None, :]
    n = nt * 16 + (lane & 15)[None, None, :, None]
    k = ks * 32 + (lane >> 4)[None, None, :, None] * 8 + j
    hi, lo = (split_f16_unscaled if unscaled else split_f16)(w.float()[n, k])
    return torch.stack([hi, lo], dim=3).contiguous()


def pack_linear_sep(w: torch.Tensor) -> torch.Tensor:
    """`pack_linear_f16x3` in the separator's split form (lo un-rescaled, lh_split.h); the embedder kernels
    (lh_embed.hip) still take the rescaled form."""
    return pack_linear_f16x3(w, unscaled=True)


def pack_block(sd: dict, pre: str) -> dict:
    g = lambda k: sd[pre + k].detach()
    out = {}
    out["intra_ln_w"], out["intra_ln_b"] = g("intra_norm.norm.weight"), g("intra_norm.norm.bias")
    out["intra_w"] = torch.stack([pack_lstm(g("intra_rnn.weight_ih_l0"), g("intra_rnn.weight_hh_l0")),
                                  pack_lstm(g("intra_rnn.weight_ih_l0_reverse"), g("intra_rnn.weight_hh_l0_reverse"))])
    # split-precision images with the LayerNorm affine folded in:  LN(x) W^T = xhat (W * ln_w)^T + W ln_b
    iw, ib = g("intra_norm.norm.weight").double(), g("intra_norm.norm.bias").double()
    ew, eb = g("inter_norm.norm.weight").double(), g("inter_norm.norm.bias").double()
    fold_w = lambda w, lw: (w.double() * lw[None, :]).float()
    fold_b = lambda w, lb, b1, b2: (b1.double() + b2.double() + w.double() @ lb).float()
    out["intra_w16"] = torch.stack([
        pack_lstm_f16x3(fold_w(g("intra_rnn.weight_ih_l0"), iw), g("intra_rnn.weight_hh_l0")),
        pack_lstm_f16x3(fold_w(g("intra_rnn.weight_ih_l0_reverse"), iw), g("intra_rnn.weight_hh_l0_reverse"))])
    intra_b = torch.stack([
        fold_b(g("intra_rnn.weight_ih_l0"), ib, g("intra_rnn.bias_ih_l0"), g("intra_rnn.bias_hh_l0")),
        fold_b(g("intra_rnn.weight_ih_l0_reverse"), ib, g("intra_rnn.bias_ih_l0_reverse"),
               g("intra_rnn.bias_hh_l0_reverse"))])
    inter_b = fold_b(g("inter_rnn.weight_ih_l0"), eb, g("inter_rnn.bias_ih_l0"), g("inter_rnn.bias_hh_l0"))
    gps = gate_prescale(64, intra_b.device)
    out["intra_b16"] = (intra_b.double() * gps[None, :]).float()          # same row scale as the images (gate_prescale)
    out["inter_w16"] = pack_lstm_f16x3(fold_w(g("inter_rnn.weight_ih_l0"), ew), g("inter_rnn.weight_hh_l0")).unsqueeze(0)
    out["inter_b16"] = (inter_b.double() * gps).float()
    out["inter_w8"] = pack_lstm_f16x3_w8(fold_w(g("inter_rnn.weight_ih_l0"), ew), g("inter_rnn.weight_hh_l0")).unsqueeze(0)
    # streaming intra kernel (lh_stream.hip): gate columns in the order n = 32 w + 8 r + 4 u + g  <->  gate g of hidden
    # unit 8 w + 2 r + u (PyTorch row g*64 + unit); W_hh additionally split into the two k halves of thread 2n + kh
    n = torch.arange(256, device=iw.device)
    perm = (n & 3) * 64 + (n >> 5) * 8 + ((n >> 3) & 3) * 2 + ((n >> 2) & 1)
    out["intra_s_wih"] = torch.stack([pack_linear_sep(fold_w(g("intra_rnn.weight_ih_l0"), iw)[perm]),
                                      pack_linear_sep(fold_w(g("intra_rnn.weight_ih_l0_reverse"), iw)[perm])])
    out["intra_s_b"] = intra_b[:, perm]
    out["intra_s_whh"] = torch.stack([g("intra_rnn.weight_hh_l0")[perm].reshape(512, 32),
                                      g("intra_rnn.weight_hh_l0_reverse")[perm].reshape(512, 32)])
    # the same layout for the per-sequence inter kernel (lh_inter_matvec, small batches)
    out["inter_s_wih"] = pack_linear_sep(fold_w(g("inter_rnn.weight_ih_l0"), ew)[perm])
    out["inter_s_b"] = inter_b[perm]
    out["inter_s_whh"] = g("inter_rnn.weight_hh_l0")[perm].reshape(512, 32)
    out["intra_b"] = torch.stack([g("intra_rnn.bias_ih_l0") + g("intra_rnn.bias_hh_l0"),
                                  g("intra_rnn.bias_ih_l0_reverse") + g("intra_rnn.bias_hh_l0_reverse")])
    out["intra_lin_w"], out["intra_lin_b"] = pack_linear_sep(g("intra_linear.weight")), g("intra_linear.bias")
    # fused kernels: per-direction halves of the bidirectional projection [2 passes][4][2][64][2][8]
    out["intra_lin_w2"] = torch.stack([pack_linear_sep(g("intra_linear.weight")[:, :64].contiguous()),
                                       pack_linear_sep(g("intra_linear.weight")[:, 64:].contiguous())])
    out["inter_ln_w"], out["inter_ln_b"] = g("inter_norm.norm.weight"), g("inter_norm.norm.bias")
    out["inter_w"] = pack_lstm(g("inter_rnn.weight_ih_l0"), g("inter_rnn.weight_hh_l0")).unsqueeze(0)
    out["inter_b"] = g("inter_rnn.bias_ih_l0") + g("inter_rnn.bias_hh_l0")
    out["inter_lin_w"], out["inter_lin_b"] = pack_linear_sep(g("inter_linear.weight")), g("inter_linear.bias")
    out["inter_lin_wu"] = pack_linear_sep(g("inter_linear.weight"))      # fused kernel (lh_inter_block)
    out["qkv_w"] = pack_linear_sep(torch.cat([g("attn_conv_Q.0.weight"), g("attn_conv_K.0.weight"),
                                                g("attn_conv_V.0.weight")], 0))
    out["qkv_b"] = torch.cat([g("attn_conv_Q.0.bias"), g("attn_conv_K.0.bias"), g("attn_conv_V.0.bias")])
    out["qkv_slopes"] = torch.cat([g("attn_conv_Q.1.weight"), g("attn_conv_K.1.weight"), g("attn_conv_V.1.weight")])
    for nm in "QKV":
        # Q / K affines zero-padded 582 -> 608: the kernel's pad features then come out as exact zeros
        padn = (QK_PAD - g(f"attn_conv_{nm}.3.norm.weight").numel()) if nm != "V" else 0
        out[f"ln{nm.lower()}_w"] = torch.nn.functional.pad(g(f"attn_conv_{nm}.3.norm.weight"), (0, padn))
        out[f"ln{nm.lower()}_b"] = torch.nn.functional.pad(g(f"attn_conv_{nm}.3.norm.bias"), (0, padn))
    out["proj_w"], out["proj_b"] = pack_linear_sep(g("attn_concat_proj.0.weight")), g("attn_concat_proj.0.bias")
    out["proj_slope"] = g("attn_concat_proj.1.weight")
    out["proj_ln_w"], out["proj_ln_b"] = g("attn_concat_proj.3.norm.weight"), g("attn_concat_proj.3.norm.bias")
    return {k: (v.contiguous() if v.dtype == torch.float16 else v.contiguous().float()) for k, v in out.items()}


def pack_mfma_f32(w_kn: torch.Tensor, k_pad: int = None) -> torch.Tensor:
    """fp32 MFMA (16x16x4) B image of a [K, N] matrix: [N/16 tiles][K/4 ksteps][64 lanes] with lane l of (nt, ks)
    holding w_kn[k = (l >> 4) * (K/4) + ks][n = nt*16 + (l & 15)] (rows >= K / columns >= N are zero)."""
    K0, N = w_kn.shape
    K = k_pad or K0
    assert K % 4 == 0 and K >= K0
    n_pad = (N + 15) // 16 * 16
    w = torch.zeros(K, n_pad, device=w_kn.device, dtype=torch.float32)
    w[:K0, :N] = w_kn.float()
    lane = torch.arange(64, device=w.device)
    k = (lane >> 4)[None, None, :] * (K // 4) + torch.arange(K // 4, device=w.device)[None, :, None]
    n = torch.arange(n_pad // 16, device=w.device)[:, None, None] * 16 + (lane & 15)[None, None, :]
    return w[k, n].contiguous()


def conv_taps_k64(w: torch.Tensor) -> torch.Tensor:
    """conv.0.weight [64, 4 ch, 3 kt, 3 kf] -> [64, K = 64] in the front end's K order (lh_frontend.hip): k = 16 kf + 4 kt + ch,
    the other 28 slots zero (they multiply finite neighbours of the spectrum tile)."""
    o, ch, kt, kf = w.shape
    assert (o, ch, kt, kf) == (64, 4, 3, 3)
    out = torch.zeros(64, 64, dtype=torch.float32, device=w.device)
    k = (16 * torch.arange(3)[None, None, :] + 4 * torch.arange(3)[None, :, None] + torch.arange(4)[:, None, None]).reshape(-1)
    out[:, k.to(w.device)] = w.float().reshape(64, -1)
    return out


def pack_all(sd: dict, n_blocks: int, prefix: str = "tfgridnet.") -> dict:
    """sd: state-dict-like mapping with the reference names -> dict of packed fp32 tensors."""
    g = lambda k: sd[prefix + k].detach()
    out = {
        # analysis filterbank as a split-precision B image: W[n = filter row][k = sample], 194 rows padded to 208
        "wfb_t": pack_linear_sep(torch.nn.functional.pad(g("enc.filterbank._filters")[:, 0], (0, 0, 0, 208 - 194))),  # [13][6][64][2][8]
        # synthesis filterbank as a split-precision B image: W[n = sample][k = spectrum row], 194 rows padded to 224
        "wfb_dec": pack_linear_sep(torch.nn.functional.pad(g("dec.filterbank._filters")[:, 0].t(), (0, 224 - 194))),   # [12][7][64][2][8]
        "conv_w": pack_linear_sep(conv_taps_k64(g("conv.0.weight"))),          # [4][2][64][2][8]
        "conv_b": g("conv.0.bias"),
        "emb_w": g("embed_to_feats_proj.0.weight"), "emb_b": g("embed_to_feats_proj.0.bias"),
        "emb_ln_w": g("embed_to_feats_proj.1.weight"), "emb_ln_b": g("embed_to_feats_proj.1.bias"),
        # transposed-conv taps as a split-precision B image: W[n = (kt,kf,o)][k = c], 36 columns padded to 48
        "deconv_w": pack_linear_sep(torch.nn.functional.pad(g("deconv.weight").permute(0, 2, 3, 1).reshape(-1, 36).t(),
                                                              (0, 0, 0, 12))),   # [3][2][64][2][8]
        "deconv_b": g("deconv.bias"),
    }
    out = {k: (v.contiguous() if v.dtype == torch.float16 else v.contiguous().float()) for k, v in out.items()}
    out["blocks"] = [pack_block(sd, f"{prefix}blocks.{i}.") for i in range(n_blocks)]
    return out
