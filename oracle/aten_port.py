"""TEST INFRASTRUCTURE — the reference's CPU path, op for op, as the `cpu_baseline` of bench.py.

`oracle/tfgridnet_oracle.py` restates the *algorithm* (index notation, a 50-slot attention loop, no unfold copy): right
for checking kernels, wrong as a statement of what the reference's CPU path costs.  This file issues the SAME ATen
operator sequence as the reference forward, with the same shapes, layouts and materialisations:

    F.conv1d (asteroid encoder)            tfgridnet_causal.py:229        cat / transpose / cat(conv_buf)    :231-239
    F.conv2d 3x3                           :242                           F.linear + F.layer_norm (embed)    :247-248
    per block (:489-590):
      permute, F.layer_norm(C), reshape, aten::lstm (bidirectional, nn.LSTM's own call), F.linear, reshape, add
      F.layer_norm(C), transpose+reshape, aten::lstm with (h0, c0), F.linear, view, transpose, add
      F.linear + F.prelu + reshape/permute/reshape + F.layer_norm(F*E)          (Q, K, V: :354-387)
      cat(K_buf, K), cat(V_buf, V), transpose + unfold(2, 50, 1) + transpose + reshape  (the 3.2 GB/clip copy, :429-454)
      matmul / sqrt(582), softmax(dim=2), matmul, reshape / transpose chain (:564-581)
      F.linear + F.prelu + reshape + F.layer_norm(6208), add, permute           (:583-588)
    cat(deconv_buf), F.conv_transpose2d, view / transpose / cat, cat(istft_buf), F.conv_transpose1d, slice  (:256-273)
    mod_pad / look-ahead pad / trims of Net.predict                              net.py:8-18, 54-66

written as plain functions over the state dict (the reference is an nn.Module tree; no code is shared).  Because the
operator sequence is the same, the result is BIT-IDENTICAL to the unmodified reference on the same machine:
`python -m oracle.aten_port` asserts that in the build container (fp32, offline + state + streaming), where
/root/reference can be imported under oracle/ref_stubs.py.  bench.py times `forward` (kind "port": the reference itself
cannot travel to the GPU box) with all host cores stated.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

TSH_PARAMS = dict(embed_dim=256, stft_chunk_size=128, stft_pad_size=64, num_ch=2, D=64, L=4, I=1, J=1,
                  B=3, H=64, local_atten_len=50, use_attn=True, lookahead=True, chunk_causal=True)


class Dims:
    def __init__(self, p=TSH_PARAMS):
        self.hop, self.pad = p["stft_chunk_size"], p["stft_pad_size"]
        self.nfft = self.hop + self.pad
        self.F = self.nfft // 2 + 1
        self.M, self.C, self.H, self.nh, self.nblk = p["num_ch"], p["D"], p["H"], p["L"], p["B"]
        self.E = math.ceil(512 / self.F)
        self.Vd = self.C // self.nh
        self.win = p["local_atten_len"]
        self.S = p.get("num_src", 2)
        self.lookahead = p["lookahead"]


def init_state(d: Dims, B: int, dtype=torch.float32, device=None):
    z = lambda *s: torch.zeros(*s, dtype=dtype, device=device)
    bufs = {f"buf{i}": dict(K_buf=z(B * d.nh, d.win - 1, d.E * d.F), V_buf=z(B * d.nh, d.win - 1, d.Vd * d.F),
                            c0=z(1, B * d.F, d.H), h0=z(1, B * d.F, d.H)) for i in range(d.nblk)}
    return dict(conv_buf=z(B, 2 * d.M, 2, d.F), deconv_buf=z(B, d.C, 2, d.F), istft_buf=z(B, d.S, 2 * d.F, 1),
                gridnet_bufs=bufs)


def _lstm(x, hx, sd, pre, bidirectional):
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    flat = [sd[pre + n] for n in names]
    if bidirectional:
        flat += [sd[pre + n + "_reverse"] for n in names]
    # nn.LSTM.forward -> _VF.lstm(input, hx, flat_weights, bias, num_layers, dropout, train, bidirectional, batch_first)
    return torch._VF.lstm(x, hx, flat, True, 1, 0.0, False, bidirectional, True)


def _proj(sd, pre, x, heads: Optional[int], width: int):
    """Sequential(Linear, PReLU, Lambda(reshape...), LayerNorm) of tfgridnet_causal.py:354-396."""
    y = F.prelu(F.linear(x, sd[pre + "0.weight"], sd[pre + "0.bias"]), sd[pre + "1.weight"])
    if heads is None:
        y = y.reshape(y.shape[0], y.shape[1], y.shape[2] * y.shape[3])
    else:
        y = y.reshape(y.shape[0], y.shape[1], y.shape[2], heads, width).permute(0, 3, 1, 2, 4) \
             .reshape(y.shape[0] * heads, y.shape[1], y.shape[2] * width)
    return F.layer_norm(y, (y.shape[-1],), sd[pre + "3.norm.weight"], sd[pre + "3.norm.bias"], 1e-5)


def _unfold_chunks(d: Dims, x):
    x = x.transpose(1, 2)
    if x.shape[-1] == d.win:
        return x
    x = x.unfold(2, d.win, 1)
    B, CF, N, L = x.shape
    return x.transpose(1, 2).reshape(B * N, CF, L)


def block(d: Dims, sd: Dict[str, torch.Tensor], pre: str, x, st):
    B, C, T, Q = x.shape
    inp = x.permute(0, 2, 3, 1)
    y = F.layer_norm(inp, (C,), sd[pre + "intra_norm.norm.weight"], sd[pre + "intra_norm.norm.bias"], 1e-5)
    y = y.reshape(B * T, Q, C)
    z2 = y.new_zeros(2, B * T, d.H)
    y = _lstm(y, (z2, z2), sd, pre + "intra_rnn.", True)[0]
    y = F.linear(y, sd[pre + "intra_linear.weight"], sd[pre + "intra_linear.bias"])
    y = y.reshape(B, T, Q, C) + inp

    inp = y
    u = F.layer_norm(y, (C,), sd[pre + "inter_norm.norm.weight"], sd[pre + "inter_norm.norm.bias"], 1e-5)
    u = u.transpose(1, 2).reshape(B * Q, T, C)
    u, h0, c0 = _lstm(u, (st["h0"], st["c0"]), sd, pre + "inter_rnn.", False)
    st["h0"], st["c0"] = h0, c0
    u = F.linear(u, sd[pre + "inter_linear.weight"], sd[pre + "inter_linear.bias"])
    u = u.view([B, Q, T, C]).transpose(1, 2) + inp

    out = u
    Qm = _proj(sd, pre + "attn_conv_Q.", u, d.nh, d.E)
    K = _proj(sd, pre + "attn_conv_K.", u, d.nh, d.E)
    V = _proj(sd, pre + "attn_conv_V.", u, d.nh, d.Vd)
    K = torch.cat([st["K_buf"], K], dim=1)
    s0 = K.shape[1] - (d.win - 1)
    st["K_buf"] = K[:, s0:s0 + d.win - 1]
    V = torch.cat([st["V_buf"], V], dim=1)
    s0 = V.shape[1] - (d.win - 1)
    st["V_buf"] = V[:, s0:s0 + d.win - 1]
    Qm = Qm.reshape(Qm.shape[0] * Qm.shape[1], 1, Qm.shape[2])
    K = _unfold_chunks(d, K)
    V = _unfold_chunks(d, V)
    att = torch.matmul(Qm, K) / (Qm.shape[-1] ** 0.5)
    att = F.softmax(att, dim=2)
    V = torch.matmul(att, V.transpose(1, 2))
    V = V.reshape(-1, T, V.shape[-1]).transpose(1, 2)
    m = V.reshape(B, d.nh, d.F, d.Vd, T).transpose(2, 3).reshape(B, d.nh * d.Vd, d.F, T).permute(0, 3, 2, 1)
    m = _proj(sd, pre + "attn_concat_proj.", m, None, 0)
    m = m.reshape(m.shape[0], m.shape[1], out.shape[2], -1)
    return (out + m).permute(0, 3, 1, 2), st


def tfgridnet(d: Dims, sd, x, emb, state):
    """TFGridNet.forward (tfgridnet_causal.py:188-283): x [B, M, N'] -> ([B, S, n], state)."""
    p = "tfgridnet."
    if state is None:
        state = init_state(d, x.shape[0], x.dtype, x.device)
    shp = x.shape
    b = F.conv1d(x.reshape(-1, 1, shp[-1]), sd[p + "enc.filterbank._filters"], stride=d.hop)
    b = b.view(*shp[:-1], b.shape[-2], b.shape[-1])
    b = torch.cat((b[..., :d.F, :], b[..., d.F:, :]), dim=1).transpose(2, 3)
    nb, _, nfr, nfq = b.shape
    b = torch.cat((state["conv_buf"], b), dim=2)
    state["conv_buf"] = b[:, :, -2:, :]
    b = F.conv2d(b, sd[p + "conv.0.weight"], sd[p + "conv.0.bias"], padding=(0, 1))
    e = F.linear(emb, sd[p + "embed_to_feats_proj.0.weight"], sd[p + "embed_to_feats_proj.0.bias"])
    e = F.layer_norm(e, (e.shape[-1],), sd[p + "embed_to_feats_proj.1.weight"], sd[p + "embed_to_feats_proj.1.bias"], 1e-5)
    e = e.reshape([nb, d.C, nfq]).unsqueeze(2)
    for i in range(d.nblk):
        if i == 1:
            b = b * e
        b, state["gridnet_bufs"][f"buf{i}"] = block(d, sd, f"{p}blocks.{i}.", b, state["gridnet_bufs"][f"buf{i}"])
    b = torch.cat((state["deconv_buf"], b), dim=2)
    state["deconv_buf"] = b[:, :, -2:, :]
    b = F.conv_transpose2d(b, sd[p + "deconv.weight"], sd[p + "deconv.bias"], padding=(2, 1))
    b = b.view([nb, d.S, 2, nfr, nfq]).transpose(3, 4)
    b = torch.cat([b[:, :, 0], b[:, :, 1]], dim=2)
    b = torch.cat([state["istft_buf"], b], dim=3)
    state["istft_buf"] = b[..., -1:]
    shp = b.shape
    y = F.conv_transpose1d(b.reshape(-1, shp[-2], shp[-1]), sd[p + "dec.filterbank._filters"], stride=d.hop)
    y = y.view(*shp[:-2], -1)
    return y[..., d.hop:], state


def predict(d: Dims, sd, x, embed, state, pad=True):
    """Net.predict (net.py:54-66)."""
    mod = 0
    if pad:
        if x.shape[-1] % d.hop != 0:
            mod = d.hop - (x.shape[-1] % d.hop)
        x = F.pad(F.pad(x, (0, mod)), (0, d.pad) if d.lookahead else (0, 0))
    y, state = tfgridnet(d, sd, x, embed, state)
    if d.lookahead:
        y = y[..., :-d.pad]
    if mod != 0:
        y = y[:, :, :-mod]
    return y, state


def forward(d: Dims, sd, x, embeds, state=None, pad=True):
    """Net.forward (net.py:68-76)."""
    with torch.no_grad():
        return predict(d, sd, x, embeds[:, 0], state, pad)[0]


def _check_against_reference():
    """Build container only: bit-equality with the unmodified reference module (same ATen ops => same bits)."""
    from oracle import ref_stubs
    from lookoncetohear_amd import synth
    torch.set_num_threads(8)
    Net = ref_stubs.reference_net_class()
    torch.manual_seed(0)
    ref = Net(**TSH_PARAMS).eval()
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    d = Dims()
    ok = True
    with torch.no_grad():
        for idx, n in (([0, 1], 8000), ([2], 8100)):
            b = synth.batch(idx, n)
            yr = ref(b["mixture"], b["embedding_gt"])
            yp = forward(d, sd, b["mixture"], b["embedding_gt"])
            same = torch.equal(yr, yp)
            print(f"offline n={n} B={len(idx)}: bit-identical {same}  (max diff {float((yr - yp).abs().max()):.1e})")
            ok &= same
        # streaming with carried state: 12 chunks
        b = synth.batch([5], 128 * 12 + 64)
        sr, sp = ref.init_buffers(1, "cpu"), init_state(d, 1)
        same = True
        for i in range(12):
            c = b["mixture"][:, :, i * 128:i * 128 + 192]
            yr, sr = ref.predict(c, b["embedding_gt"][:, 0], sr, pad=False)
            yp, sp = predict(d, sd, c, b["embedding_gt"][:, 0], sp, pad=False)
            same &= torch.equal(yr, yp)
        for k in ("K_buf", "V_buf", "h0", "c0"):
            same &= torch.equal(sr["gridnet_bufs"]["buf2"][k], sp["gridnet_bufs"]["buf2"][k])
        print(f"streaming 12 chunks + final state: bit-identical {same}")
        ok &= same
    assert ok, "aten_port differs from the reference"
    print("oracle/aten_port.py == reference (bitwise, fp32, this machine)")


if __name__ == "__main__":
    _check_against_reference()
