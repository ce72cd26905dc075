"""`EmbedTFGridNet` — MI355X-native drop-in for the reference enrollment embedder
`src.models.tfgridnet_orig.tfgridnet.EmbedTFGridNet` (configs/embed.json:4: `"model": "lookoncetohear_amd.embed_net.EmbedTFGridNet"`).

Same constructor (`embed_dim, num_ch, n_fft, stride, num_blocks`, reference tfgridnet_orig/tfgridnet.py:89) and call
(`forward(input [B, M, N]) -> [B, embed_dim]`, :100-127; used as `enroll_model.model(enrollments)` at
src/ts_hear_test.py:129,134).  Parameter names follow the espnet2 `TFGridNet` module tree the reference subclasses
(`conv.0/1`, `blocks.i.{intra,inter}_{norm,rnn,linear}`, `attn_conv_{Q,K,V}_h.{0,1,2}`, `attn_concat_proj`, `deconv`,
plus the reference's own `embed_proj`), as restated in oracle/embedder_oracle.py — espnet2 itself is not available
here, so that manifest is unverified against a real checkpoint (PARITY UNPINNED, see the oracle's header).

Like `Net`, the torch modules are parameter containers; the arithmetic runs in HIP kernels (lh_embed.hip and the
templated frame kernels of lh_pointwise.hip) behind the C ABI.  No CPU fallback.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _cabi
from .weights import pack_linear_sep as pack_linear_f16x3, pack_mfma_f32, split_f16_unscaled as split_f16      # un-rescaled split (lh_embed.hip ESPLIT = 1)


class _LN4D(nn.Module):
    def __init__(self, shape):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(*shape))
        self.beta = nn.Parameter(torch.zeros(*shape))


def _head_conv(c_in, c_out, n_freqs):
    return nn.Sequential(nn.Conv2d(c_in, c_out, 1), nn.PReLU(), _LN4D((1, c_out, 1, n_freqs)))


class _EBlock(nn.Module):
    def __init__(self, emb_dim, emb_ks, n_freqs, hidden, n_head):
        super().__init__()
        E = math.ceil(512 / n_freqs)
        self.intra_norm = _LN4D((1, emb_dim, 1, 1))
        self.intra_rnn = nn.LSTM(emb_dim * emb_ks, hidden, 1, batch_first=True, bidirectional=True)
        self.intra_linear = nn.ConvTranspose1d(hidden * 2, emb_dim, emb_ks, stride=1)
        self.inter_norm = _LN4D((1, emb_dim, 1, 1))
        self.inter_rnn = nn.LSTM(emb_dim * emb_ks, hidden, 1, batch_first=True, bidirectional=True)
        self.inter_linear = nn.ConvTranspose1d(hidden * 2, emb_dim, emb_ks, stride=1)
        for h in range(n_head):
            self.add_module(f"attn_conv_Q_{h}", _head_conv(emb_dim, E, n_freqs))
            self.add_module(f"attn_conv_K_{h}", _head_conv(emb_dim, E, n_freqs))
            self.add_module(f"attn_conv_V_{h}", _head_conv(emb_dim, emb_dim // n_head, n_freqs))
        self.attn_concat_proj = _head_conv(emb_dim, emb_dim, n_freqs)


def stft_rows(n_fft: int) -> torch.Tensor:
    """[n_fft samples, n_fft + 2 rows]: periodic-hann-windowed DFT rows, cos (re) then -sin (im): torch.stft semantics."""
    n = np.arange(n_fft, dtype=np.float64)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)
    k = np.arange(n_fft // 2 + 1)
    ang = 2.0 * np.pi * np.outer(n, k) / n_fft
    return torch.from_numpy(np.concatenate([np.cos(ang), -np.sin(ang)], axis=1) * win[:, None]).float()


class EmbedTFGridNet(_cabi.HipHost, nn.Module):
    _host_name = "EmbedTFGridNet"

    def __init__(self, embed_dim, num_ch, n_fft, stride, num_blocks):
        super().__init__()
        if (embed_dim, num_ch, n_fft, stride) != (256, 2, 128, 64):
            raise NotImplementedError("the gfx950 embedder kernels are specialised on configs/embed.json "
                                      f"(embed_dim 256, 2 mics, n_fft 128, stride 64); got {(embed_dim, num_ch, n_fft, stride)}")
        self.embed_dim, self.n_imics, self.n_fft, self.stride, self.n_layers = embed_dim, num_ch, n_fft, stride, num_blocks
        self.emb_dim, self.n_freqs, self.n_head, self.hidden, self.emb_ks = 64, n_fft // 2 + 1, 4, 64, 4
        self.E = math.ceil(512 / self.n_freqs)
        self.conv = nn.Sequential(nn.Conv2d(2 * num_ch, 64, (3, 3), padding=(1, 1)), nn.GroupNorm(1, 64, eps=1e-5))
        self.blocks = nn.ModuleList([_EBlock(64, 4, self.n_freqs, 64, 4) for _ in range(num_blocks)])
        self.deconv = nn.ConvTranspose2d(64, 2, (3, 3), padding=(1, 1))      # registered by the espnet2 trunk, unused here
        self.embed_proj = nn.Sequential(nn.Linear(self.n_freqs * 64, embed_dim), nn.LayerNorm(embed_dim))
        self._pack_key = None
        self._packed = None
        # axis path: k_emb_rec (input GEMM inside the recurrence, round 4) or the round-1 three-kernel form with the gate
        # pre-activations through HBM (LOOKONCE_EMB_FUSED=0; A/B runs against a -DLH_LEGACY lab build only: the product
        # library does not contain it and the call fails loudly)
        self.fused_axis = os.environ.get("LOOKONCE_EMB_FUSED", "1") != "0"
        self.n_streams = int(os.environ.get("LOOKONCE_EMB_STREAMS", "2"))
        # inter-axis recurrence: one workgroup per (sequence, direction) (k_emb_inter_mv: ~0.4 us per step, up to three workgroups per
        # CU) while those fit this many workgroups — B <= 7: one 5 s enrollment 5.66 -> 3.07 ms, two 6.18 -> 4.13, three 6.84 -> 4.80,
        # four 7.45 -> 5.91, five 8.54 -> 6.96, six 8.90 -> 8.73, seven 9.81 -> 9.67 (profiles/r06i_embed_small_batch.txt) — 16-sequence
        # tiles (k_emb_rec: ~1.4 us per step) above.  LOOKONCE_EMB_MV_MAX_WGS overrides.
        self.inter_mv_max_wgs = int(os.environ.get("LOOKONCE_EMB_MV_MAX_WGS", "1024"))
        self._side = None
        self._debug_taps: Optional[dict] = None
        self._prof: Optional[list] = None  # bench.py: (C-ABI call, start event, end event) per launch

    # ------------------------------------------------------------------------------------------------
    def _weights(self, device) -> dict:
        tensors = list(self.parameters())
        key = (str(device),) + tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self._pack_key:
            with torch.no_grad():
                sd = {k: v.detach() for k, v in self.state_dict().items()}
                if os.environ.get("LOOKONCE_PACK_ON_HOST") == "1":
                    # the packers are index arithmetic + fp16 splits (~6000 tiny indexing launches on the device): under
                    # rocprofv3 --pmc that dispatch storm crashes the profiler (round 5, profiles/README.md), so the
                    # profiling recipe packs on the host and uploads the images — same bits, one copy per tensor
                    dev = next(iter(sd.values())).device
                    to_dev = lambda o: (o.to(dev) if torch.is_tensor(o) else
                                        ({k: to_dev(v) for k, v in o.items()} if isinstance(o, dict) else [to_dev(v) for v in o]))
                    self._packed = to_dev(pack_embedder({k: v.cpu() for k, v in sd.items()}, self.n_layers))
                else:
                    self._packed = pack_embedder(sd, self.n_layers)
            self._pack_key = key
        return self._packed

    def forward(self, input):
        """[B, M, N] -> [B, embed_dim].  Utterances are independent (per-utterance std, GroupNorm, attention, frame mean), so
        with `n_streams` >= 2 (default 2; LOOKONCE_EMB_STREAMS; GPU only, >= 16 utterances per part) the batch runs as parts on
        separate HIP streams: measured 65.2 -> 61.6 ms at B = 64 with bit-identical embeddings —
        the ragged last round of workgroups of one half's recurrent / GEMM launches (the inter-axis recurrence of 64 clips
        is 520 workgroups of one-per-CU on 256 CUs: a third round with 8 of them) is filled by the other half's kernels."""
        ns = self.n_streams
        ns = min(ns, input.shape[0] // 16)
        if ns > 1 and self._multi_stream(input) and self._prof is None and self._debug_taps is None:
            dev = input.device
            with torch.no_grad():
                self._weights(dev)                                      # packed once, on the caller's stream
            cur = torch.cuda.current_stream(dev)
            if self._side is None or self._side[0] != dev or len(self._side[1]) < ns - 1:
                # sized for the configured stream count, not for this call's batch-capped `ns` (a first call at
                # B = 32 gave one side stream and a later B = 64 call with n_streams = 4 indexed past it)
                self._side = (dev, [torch.cuda.Stream(device=dev) for _ in range(max(ns, self.n_streams) - 1)])
            parts = list(input.chunk(ns))
            outs = [None] * len(parts)
            for i in range(1, len(parts)):
                s_ = self._side[1][i - 1]
                s_.wait_stream(cur)
                with torch.cuda.stream(s_):
                    outs[i] = self._forward_one(parts[i])
            outs[0] = self._forward_one(parts[0])
            for i in range(1, len(parts)):
                cur.wait_stream(self._side[1][i - 1])
                outs[i].record_stream(cur)
            return torch.cat(outs, 0)
        return self._forward_one(input)

    def _multi_stream(self, input) -> bool:
        return True

    def _forward_one(self, input):
        lib = self._lib(input)
        dev = input.device
        x = input.contiguous().float()
        B, M, N = x.shape
        if M != self.n_imics:
            raise ValueError(f"expected {self.n_imics} microphones, got {M}")
        T = N // self.stride + 1
        if T < self.emb_ks or N <= self.n_fft // 2:
            raise ValueError(f"enrollment of {N} samples is too short: needs > {self.n_fft // 2} samples (reflect padding) "
                             f"and >= {self.emb_ks} STFT frames (unfold kernel), like the reference")
        F_, C_ = self.n_freqs, 64
        if self.training and torch.is_grad_enabled():
            raise RuntimeError("lookoncetohear_amd.EmbedTFGridNet is an inference-only drop-in (no autograd): call .eval() "
                               "and/or run under torch.no_grad()")
        with torch.no_grad(), self._device_ctx(x):
            pk = self._weights(dev)
            st = self._stream(dev)
            P = lambda t: t.data_ptr()
            e = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
            taps = self._debug_taps
            prof = self._prof
            if prof is not None:
                lib_ = lib

                class lib:                      # HIP events on the launch stream around each C-ABI call
                    @staticmethod
                    def call(name, *args):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        lib_.call(name, *args)
                        e1.record()
                        tag = name
                        if name == "lh_emb_axis":
                            tag = name + (".inter" if args[12] else ".intra")
                        elif name == "lh_emb_axis_fused":
                            tag = "lh_emb_axis" + (".inter" if args[10] else ".intra")
                        elif name == "lh_emb_axis_mv":
                            tag = "lh_emb_axis.inter"
                        prof.append((tag, e0, e1))
            za, zb, zc = e(B, T, F_, C_), e(B, T, F_, C_), e(B, T, F_, C_)
            inv_std = e(B)
            tiles = B * ((T + 13) // 14)
            gn_part = torch.empty(tiles * 2, device=dev, dtype=torch.float64)
            xsp = e(B, T, F_, C_)              # fp16 hi | lo images of the normalised axis input (same bytes as fp32)
            lib.call("lh_emb_frontend", P(x), P(inv_std), P(pk["wfb"]), P(pk["conv_w"]), P(pk["conv_b"]), P(pk["gn_w"]),
                     P(pk["gn_b"]), P(gn_part), P(za), P(xsp) if self.fused_axis else None, B, T, N, st)
            if taps is not None:
                taps["z0"] = za.clone()
            P_i, P_e = F_ - 3, T - 3
            gx = e(max(B * T * P_i, B * F_ * P_e) * 512) if not self.fused_axis else None
            hbuf = e(max(B * T * P_i, B * F_ * P_e) * 128)
            nb, Tp = self.n_head * B, (T + 63) // 64 * 64
            h16 = lambda n: torch.empty(n, device=dev, dtype=torch.float16)
            qb, kb = h16(2 * nb * T * 544), h16(2 * nb * T * 544)       # fp16 hi | lo images, rows padded 520 -> 544
            vb = e(B * T * F_ * C_)
            vtb, scb, pb = h16(2 * nb * 1040 * Tp), e(nb * T * Tp), h16(2 * nb * T * Tp)
            for i in range(self.n_layers):
                bp = pk["blocks"][i]
                if self.fused_axis:
                    # the normalised, split input of an axis call is emitted by whichever kernel wrote that activation: the
                    # intra call's transposed conv for the inter call, the attention block's projection for the next
                    # block's intra call, the front end's GroupNorm pass for block 0
                    lib.call("lh_emb_axis_fused", P(za), P(bp["intra_wrec"]), P(bp["intra_brec"]), P(bp["intra_wct"]),
                             P(bp["intra_bct"]), P(xsp), P(hbuf), P(zb), B, T, 0, 1, 1, st)
                    if 2 * B * F_ <= self.inter_mv_max_wgs:
                        # few sequences: one workgroup per (sequence, direction), mat-vec recurrence (k_emb_inter_mv)
                        lib.call("lh_emb_axis_mv", P(zb), P(bp["inter_wih"]), P(bp["inter_bih"]), P(bp["inter_whh_mv"]),
                                 P(bp["inter_wct"]), P(bp["inter_bct"]), P(xsp), P(hbuf), P(zc), B, T, 1, 0, st)
                    else:
                        lib.call("lh_emb_axis_fused", P(zb), P(bp["inter_wrec"]), P(bp["inter_brec"]), P(bp["inter_wct"]),
                                 P(bp["inter_bct"]), P(xsp), P(hbuf), P(zc), B, T, 1, 1, 0, st)
                else:
                    lib.call("lh_emb_axis", P(za), P(bp["intra_wih"]), P(bp["intra_bih"]), P(bp["intra_whh"]), P(bp["intra_wct"]),
                             P(bp["intra_bct"]), P(xsp), P(gx), P(hbuf), P(zb), B, T, 0, st)
                    lib.call("lh_emb_axis", P(zb), P(bp["inter_wih"]), P(bp["inter_bih"]), P(bp["inter_whh"]), P(bp["inter_wct"]),
                             P(bp["inter_bct"]), P(xsp), P(gx), P(hbuf), P(zc), B, T, 1, st)
                if taps is not None:
                    taps[f"blocks.{i}.x1"], taps[f"blocks.{i}.x2"] = zb.clone(), zc.clone()
                lib.call("lh_emb_attn_block", P(zc), P(bp["wqkv"]), P(bp["bqkv"]), P(bp["slopes"]), P(bp["lnq_w"]),
                         P(bp["lnq_b"]), P(bp["lnk_w"]), P(bp["lnk_b"]), P(bp["lnv_w"]), P(bp["lnv_b"]), P(bp["wproj"]),
                         P(bp["bproj"]), P(bp["slope_p"]), P(bp["lnp_w"]), P(bp["lnp_b"]), P(qb), P(kb), P(vb), P(vtb), P(scb), P(pb), P(zb), P(za),
                         P(xsp) if (self.fused_axis and i + 1 < self.n_layers) else None, B, T, st)
                if taps is not None:
                    taps[f"blocks.{i}.O"], taps[f"blocks.{i}.out"] = zb.clone(), za.clone()
            emb = e(B, self.embed_dim)
            part = e(B * ((T + 63) // 64) * self.embed_dim)
            lib.call("lh_emb_head", P(za), P(pk["head_w"]), P(pk["head_b"]), P(pk["head_ln_w"]), P(pk["head_ln_b"]), P(part),
                     P(emb), B, T, st)
            return emb


def _pack_whh_f16x3(w_hh: torch.Tensor) -> torch.Tensor:
    """[4 waves][4 gates][2 ksteps][64 lanes][hi|lo][8]: W_hh[g*64 + 16w + (l & 15)][ks*32 + (l >> 4)*8 + j]."""
    dev = w_hh.device
    lane = torch.arange(64, device=dev)
    wave = torch.arange(4, device=dev)[:, None, None, None, None]
    gate = torch.arange(4, device=dev)[None, :, None, None, None]
    ks = torch.arange(2, device=dev)[None, None, :, None, None]
    j = torch.arange(8, device=dev)[None, None, None, None, :]
    col = gate * 64 + wave * 16 + (lane & 15)[None, None, None, :, None]
    k = ks * 32 + (lane >> 4)[None, None, None, :, None] * 8 + j
    hi, lo = split_f16(w_hh.float()[col, k])
    return torch.stack([hi, lo], dim=4).contiguous()


def pack_rec(sd, pre, ax):
    """Weights of k_emb_rec (lh_embed.hip): the transposed gate GEMM's MFMA A fragments, fp16 hi | lo (lo un-rescaled),
    [2 dirs][8 waves][40 fragments][64 lanes][8], fragment order per wave: tile j = 0, 1: { window slot k4 = 0..3:
    k-step 0, 1: hi, lo } then { W_hh k-step 0, 1: hi, lo }.  Lane l holds row  gate*64 + unit  (gate = l & 3,
    unit = 8 wave + 2 ((l & 15) >> 2) + j) at k = 32 ks + 8 (l >> 4) + e.  W_ih columns: channel k of window slot k4 =
    espnet2's unfolded feature k*4 + tap with tap = k4 for the forward direction and 3 - k4 for the reverse one (the
    kernel walks mirrored positions), times the LayerNorm gamma; rows times the gate's exponent factor (weights.py
    gate_prescale).  Returns (image, bias [2][256] in (unit, gate) order with W_ih beta + b_ih + b_hh, same factor)."""
    from .weights import gate_prescale
    g = lambda k: sd[pre + k].double()
    dev = sd[pre + f"{ax}_rnn.weight_ih_l0"].device
    lw, lb = g(f"{ax}_norm.gamma").reshape(-1), g(f"{ax}_norm.beta").reshape(-1)
    scale = gate_prescale(64, dev)                                                 # [256] rows gate*64 + unit
    lane = torch.arange(64, device=dev)
    m, q = lane & 15, lane >> 4
    e = torch.arange(8, device=dev)
    imgs, biases = [], []
    for d, sfx in enumerate(("", "_reverse")):
        w = g(f"{ax}_rnn.weight_ih_l0{sfx}").reshape(256, 64, 4)                   # [row, channel, tap]
        b = (g(f"{ax}_rnn.bias_ih_l0{sfx}") + g(f"{ax}_rnn.bias_hh_l0{sfx}") + (w * lb[None, :, None]).sum((1, 2))) * scale
        w = w * lw[None, :, None] * scale[:, None, None]
        whh = g(f"{ax}_rnn.weight_hh_l0{sfx}") * scale[:, None]                    # [row, 64]
        per_wave = []
        for wave in range(8):
            frags = []
            for j in range(2):
                row = (m & 3) * 64 + 8 * wave + 2 * (m >> 2) + j                   # [64 lanes]
                for k4 in range(4):
                    tap = k4 if d == 0 else 3 - k4
                    for ks in range(2):
                        kk = ks * 32 + q[:, None] * 8 + e[None, :]                 # [64, 8]
                        v = w[row[:, None], kk, tap].float()
                        hi, lo = split_f16(v)
                        frags += [hi, lo]
                for ks in range(2):
                    kk = ks * 32 + q[:, None] * 8 + e[None, :]
                    hi, lo = split_f16(whh[row[:, None], kk].float())
                    frags += [hi, lo]
            per_wave.append(torch.stack(frags))                                    # [40, 64, 8]
        imgs.append(torch.stack(per_wave))
        biases.append(b.reshape(4, 64).t().reshape(-1).float())                    # (unit, gate)
    return torch.stack(imgs).contiguous(), torch.stack(biases).contiguous()


def _pack_axis(sd, pre, ax):
    """Input GEMM of one axis path: both directions, LN affine folded, features reordered from espnet's unfold order
    (c*4 + k) to window-major (k*64 + c), output columns reordered to (direction, unit, gate)."""
    g = lambda k: sd[pre + k].double()
    lw, lb = g(f"{ax}_norm.gamma").reshape(-1), g(f"{ax}_norm.beta").reshape(-1)
    ws, bs, hh = [], [], []
    for sfx in ("", "_reverse"):
        w = g(f"{ax}_rnn.weight_ih_l0{sfx}").reshape(256, 64, 4)                 # [col, c, k]
        b = g(f"{ax}_rnn.bias_ih_l0{sfx}") + g(f"{ax}_rnn.bias_hh_l0{sfx}") + (w * lb[None, :, None]).sum((1, 2))
        w = (w * lw[None, :, None]).permute(0, 2, 1).reshape(256, 256)           # [col, k*64 + c]
        perm = (torch.arange(4, device=w.device)[None, :] * 64 + torch.arange(64, device=w.device)[:, None]).reshape(-1)  # new unit*4+gate <- gate*64+unit
        ws.append(w[perm]); bs.append(b[perm])
        hh.append(g(f"{ax}_rnn.weight_hh_l0{sfx}")[perm].float())               # rows 4 unit + gate (lh_quad.h quad_load_w)
    out = {
        f"{ax}_wih": pack_linear_f16x3(torch.cat(ws, 0).float()), f"{ax}_bih": torch.cat(bs).float().contiguous(),
        f"{ax}_whh": torch.stack([_pack_whh_f16x3(sd[pre + f"{ax}_rnn.weight_hh_l0"]),
                                  _pack_whh_f16x3(sd[pre + f"{ax}_rnn.weight_hh_l0_reverse"])]),
        f"{ax}_wct": pack_linear_f16x3(sd[pre + f"{ax}_linear.weight"].permute(1, 2, 0).reshape(64, 512).float().contiguous()),
        f"{ax}_bct": sd[pre + f"{ax}_linear.bias"].float().contiguous(),
    }
    out[f"{ax}_whh_mv"] = torch.stack(hh).contiguous()                           # fp32 [2][256][64]: k_emb_inter_mv
    out[f"{ax}_wrec"], out[f"{ax}_brec"] = pack_rec(sd, pre, ax)
    return out


def pack_embedder(sd: Dict[str, torch.Tensor], n_blocks: int) -> dict:
    g = lambda k: sd[k]
    dev = g("conv.0.weight").device
    out = {
        "wfb": pack_mfma_f32(stft_rows(128).to(dev)),                                   # [128, 130] -> [9][32][64]
        "conv_w": pack_mfma_f32(g("conv.0.weight").reshape(64, 36).t()),                # [36, 64] -> [4][9][64]
        "conv_b": g("conv.0.bias").float().contiguous(),
        "gn_w": g("conv.1.weight").float().contiguous(), "gn_b": g("conv.1.bias").float().contiguous(),
    }
    out["blocks"] = []
    for i in range(n_blocks):
        pre = f"blocks.{i}."
        bp = {}
        bp.update(_pack_axis(sd, pre, "intra"))
        bp.update(_pack_axis(sd, pre, "inter"))
        bp.update(_pack_attn(sd, pre))
        out["blocks"].append(bp)
    # head: Linear input features (c*65 + f) -> (f*64 + c)
    w = g("embed_proj.0.weight").float()
    out["head_w"] = pack_linear_f16x3(w.reshape(-1, 64, 65).permute(0, 2, 1).reshape(w.shape[0], -1).contiguous())
    out["head_b"] = g("embed_proj.0.bias").float().contiguous()
    out["head_ln_w"], out["head_ln_b"] = g("embed_proj.1.weight").float().contiguous(), g("embed_proj.1.bias").float().contiguous()
    return out


def _pack_attn(sd, pre, n_head=4):
    """Stacked Q|K|V head convs (columns Q h*8+e, K 32+h*8+e, V 64+h*16+v), per-column PReLU slopes, LayerNorm affines
    re-ordered from [channel][bin] to (bin*d + channel)."""
    g = lambda k: sd[pre + k].float()
    ws, bs, sl, ln = [], [], [], {}
    for nm in "QKV":
        lw, lb = [], []
        for h in range(n_head):
            q = f"attn_conv_{nm}_{h}."
            w = g(q + "0.weight")
            d = w.shape[0]
            ws.append(w.reshape(d, 64)); bs.append(g(q + "0.bias")); sl.append(g(q + "1.weight").reshape(1).expand(d))
            lw.append(g(q + "2.gamma").reshape(d, -1).t().reshape(-1)); lb.append(g(q + "2.beta").reshape(d, -1).t().reshape(-1))
        ln[nm] = (torch.stack(lw).contiguous(), torch.stack(lb).contiguous())
    q = "attn_concat_proj."
    return {
        "wqkv": pack_linear_f16x3(torch.cat(ws, 0).contiguous()), "bqkv": torch.cat(bs).contiguous(),
        "slopes": torch.cat(sl).contiguous(),
        "lnq_w": ln["Q"][0], "lnq_b": ln["Q"][1], "lnk_w": ln["K"][0], "lnk_b": ln["K"][1], "lnv_w": ln["V"][0], "lnv_b": ln["V"][1],
        "wproj": pack_linear_f16x3(g(q + "0.weight").reshape(64, 64).contiguous()), "bproj": g(q + "0.bias").contiguous(),
        "slope_p": g(q + "1.weight").reshape(1).contiguous(),
        "lnp_w": g(q + "2.gamma").reshape(64, -1).t().reshape(-1).contiguous(),
        "lnp_b": g(q + "2.beta").reshape(64, -1).t().reshape(-1).contiguous(),
    }
