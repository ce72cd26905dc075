"""`Net` — MI355X-native drop-in for the reference separator `src.models.tfgridnet_realtime.net.Net`.

Installation = changing one string in the reference config (configs/tsh.json:4):

    "model": "lookoncetohear_amd.net.Net"

because the reference builds its model with `utils.import_attr(model)(**model_params)`
(reference src/ts_hear_embed_pl_module.py:25).  This class therefore mirrors, exactly:

  * the constructor keywords and defaults of reference net.py:21-24 (configs/tsh.json:5-19 pass unchanged);
  * `forward(x, embeds, input_state=None, pad=True)`, `predict(x, embed, input_state, pad=True)`,
    `init_buffers(batch_size, device)` (reference net.py:51-76) with the same state-dict-of-tensors layout
    (reference tfgridnet_causal.py:173-186, 408-427), mutated and returned like the reference does;
  * the parameter / buffer names and shapes of the reference module tree (`tfgridnet.blocks.0.intra_rnn.
    weight_ih_l0`, ... SURVEY.md §8b), so a reference Lightning checkpoint loads with strict=True, and the same
    construction order, so `torch.manual_seed(s); Net(**params)` draws the same initial weights.

The torch modules below are parameter containers only: they are never called.  All arithmetic runs in the
hand-written gfx950 kernels behind the C ABI of include/lookonce_hip.h; there is no CPU fallback (the product
path raises if the HIP library is missing or the tensors are not on the GPU).
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _cabi
from .weights import KV_PAD_ROWS, QK_PAD, pack_all, unsplit_qk, unsplit_v


def mod_pad(x: torch.Tensor, chunk_size: int, pad: Tuple[int, int]):
    """Right-pad to a multiple of `chunk_size`, then apply `pad` (reference net.py:8-18)."""
    mod = 0
    if x.shape[-1] % chunk_size != 0:
        mod = chunk_size - (x.shape[-1] % chunk_size)
    # one zero-filled copy for both pads (the reference's two F.pad calls cost two copies of the batch; with nothing to pad,
    # none: the kernels only read x)
    if mod or pad[0] or pad[1]:
        x = F.pad(x, (pad[0], mod + pad[1]))
    return x, mod


def stft_filterbank(n_fft: int, hop: int) -> torch.Tensor:
    """Analysis/synthesis filterbank buffer `[n_fft + 2, 1, n_fft]` (asteroid-filterbanks STFTFB as called at
    reference tfgridnet_causal.py:131-135): sqrt-periodic-Hann windowed real-DFT rows, 97 cosine rows then 97
    negative-sine rows, DC/Nyquist cosine rows scaled by 1/sqrt(2), overall 1/(0.5*sqrt(n_fft^2/hop))."""
    nbin = n_fft // 2 + 1
    n = np.arange(n_fft)
    window = np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft))
    ang = 2 * np.pi * np.arange(nbin)[:, None] * n[None, :] / n_fft
    re, im = np.cos(ang), -np.sin(ang)
    re[0] /= np.sqrt(2)
    re[n_fft // 2] /= np.sqrt(2)
    rows = np.vstack([re, im]) * window[None, :] / (0.5 * np.sqrt(n_fft * n_fft / hop))
    return torch.from_numpy(rows).float().unsqueeze(1)


class _FilterBank(nn.Module):
    def __init__(self, n_fft, hop):
        super().__init__()
        self.register_buffer("_filters", stft_filterbank(n_fft, hop))


class _Coder(nn.Module):          # name holder for `enc.filterbank._filters` / `dec.filterbank._filters`
    def __init__(self, n_fft, hop):
        super().__init__()
        self.filterbank = _FilterBank(n_fft, hop)


class _Norm(nn.Module):           # name holder for `*.norm.{weight,bias}`
    def __init__(self, n, eps):
        super().__init__()
        self.norm = nn.LayerNorm(n, eps=eps)


def _proj(n_in, n_out, n_norm, eps):
    # Linear, PReLU (single slope, espnet2 get_layer("prelu")()), [reshape], joint LayerNorm -> indices 0,1,3
    return nn.Sequential(nn.Linear(n_in, n_out), nn.PReLU(), nn.Identity(), _Norm(n_norm, eps))


class _GridNetBlockParams(nn.Module):
    """Parameters of one causal GridNet block, created in the reference order (tfgridnet_causal.py:334-396)."""

    def __init__(self, emb_dim, n_freqs, hidden, n_head, eps):
        super().__init__()
        E = math.ceil(512 * 1.0 / n_freqs)
        self.intra_norm = _Norm(emb_dim, eps)
        self.intra_rnn = nn.LSTM(emb_dim, hidden, 1, batch_first=True, bidirectional=True)
        self.intra_linear = nn.Linear(hidden * 2, emb_dim)
        self.inter_norm = _Norm(emb_dim, eps)
        self.inter_rnn = nn.LSTM(emb_dim, hidden, 1, batch_first=True, bidirectional=False)
        self.inter_linear = nn.Linear(hidden, emb_dim)
        self.attn_conv_Q = _proj(emb_dim, E * n_head, n_freqs * E, eps)
        self.attn_conv_K = _proj(emb_dim, E * n_head, n_freqs * E, eps)
        self.attn_conv_V = _proj(emb_dim, (emb_dim // n_head) * n_head, n_freqs * (emb_dim // n_head), eps)
        self.attn_concat_proj = _proj(emb_dim, emb_dim, n_freqs * emb_dim, eps)


class _TFGridNetParams(nn.Module):
    """Parameter tree of the reference causal TFGridNet (tfgridnet_causal.py:113-171), same names/order."""

    def __init__(self, n_fft, stride, n_imics, n_srcs, emb_dim, n_layers, hidden, n_head, spk_emb_dim, eps=1.0e-5):
        super().__init__()
        n_freqs = n_fft // 2 + 1
        self.enc = _Coder(n_fft, stride)
        self.dec = _Coder(n_fft, stride)
        self.conv = nn.Sequential(nn.Conv2d(2 * n_imics, emb_dim, (3, 3), padding=(0, 1)))
        self.blocks = nn.ModuleList(
            [_GridNetBlockParams(emb_dim, n_freqs, hidden, n_head, eps) for _ in range(n_layers)])
        self.embed_to_feats_proj = nn.Sequential(nn.Linear(spk_emb_dim, emb_dim * n_freqs),
                                                 nn.LayerNorm(emb_dim * n_freqs))
        self.deconv = nn.ConvTranspose2d(emb_dim, n_srcs * 2, (3, 3), padding=(2, 1))


class _Lanes:
    """K HIP streams forked from and joined to the current stream of `dev` (the windows of `Net.time_chunks`).  The only
    stream / event calls of the chunked block loop live here; tests/hipemu/hosts.py substitutes a serial stand-in."""
    serial = False

    def __init__(self, dev, side_streams):
        # lane 0 is the CURRENT stream (the chip has four hardware queues per process by default: a fifth stream would share
        # one and serialise behind it — profiles/r06e: the fourth side stream's first kernel started 300 us late)
        self.dev = dev
        self.main = torch.cuda.current_stream(dev)
        self.streams = [self.main] + list(side_streams)

    def fork(self):
        ev = torch.cuda.Event()
        ev.record(self.main)
        for s_ in self.streams[1:]:
            s_.wait_event(ev)

    def on(self, k):
        return torch.cuda.stream(self.streams[k])

    def signal(self, k):
        ev = torch.cuda.Event()
        ev.record(self.streams[k])
        return ev

    def wait(self, k, ev):
        self.streams[k].wait_event(ev)

    def join(self):
        for s_ in self.streams[1:]:
            self.main.wait_stream(s_)


class Net(_cabi.HipHost, nn.Module):
    _host_name = "Net"

    def __init__(self, stft_chunk_size=160, stft_pad_size=120, embed_dim=256,
                 num_ch=2, D=64, B=6, I=1, J=1, L=0, H=128,
                 use_attn=False, lookahead=True, local_atten_len=100,
                 chunk_causal=False, num_src=2):
        super().__init__()
        self.ctor_params = dict(stft_chunk_size=stft_chunk_size, stft_pad_size=stft_pad_size, embed_dim=embed_dim,
                                num_ch=num_ch, D=D, B=B, I=I, J=J, L=L, H=H, use_attn=use_attn, lookahead=lookahead,
                                local_atten_len=local_atten_len, chunk_causal=chunk_causal, num_src=num_src)
        self.stft_chunk_size = stft_chunk_size
        self.stft_pad_size = stft_pad_size
        self.num_ch = num_ch
        self.lookahead = lookahead
        self.nfft = stft_chunk_size + stft_pad_size
        # I, J (emb_ks / emb_hs) are accepted and unused, like the reference causal block (:399-400)
        self.n_blocks, self.emb_dim, self.hidden, self.n_head = B, D, H, L
        self.n_freqs = self.nfft // 2 + 1
        self.local_atten_len = local_atten_len
        self.n_srcs = num_src
        self.spk_emb_dim = embed_dim
        if not (use_attn and chunk_causal):
            raise NotImplementedError("only the chunk-causal attention mode of configs/tsh.json is implemented")
        shape = (self.nfft, stft_chunk_size, num_ch, D, H, L, local_atten_len, num_src, embed_dim)
        if shape != (192, 128, 2, 64, 64, 4, 50, 2, 256):
            raise NotImplementedError(
                f"the gfx950 kernels are specialised on the configs/tsh.json shapes; got (nfft, hop, mics, D, H, L, "
                f"window, srcs, embed_dim) = {shape}")
        self.E = math.ceil(512 * 1.0 / self.n_freqs)
        self.V_dim = D // L
        self.tfgridnet = _TFGridNetParams(self.nfft, stft_chunk_size, num_ch, num_src, D, B, H, L, embed_dim)
        # contraction arithmetic of the two RECURRENCES: "f16x3" = split-precision fp16 MFMA (hi/lo operands, ~22 mantissa
        # bits, ~5x the fp32-MFMA rate), "f32rec" = exact fp32 MFMA in the intra / inter LSTMs (the name says what it
        # covers: the frame kernels — STFT / conv, Q/K/V, attention, projection, deconv / iSTFT — are split-precision in
        # every mode); "f32all" = additionally the frame kernels on plain-fp32 reference kernels (lh_ref32.hip: slow,
        # test-only — the all-fp32 run that separates "split-precision error" from "kernel bug" on a real checkpoint).
        # LOOKONCE_GEMM overrides.
        self.gemm_mode = os.environ.get("LOOKONCE_GEMM", "f16x3")
        self._check_gemm_mode()                 # a stale LOOKONCE_GEMM=f32 (the spelling removed in round 5) fails HERE
        # range guard of the split-precision kernels (include/lookonce_hip.h): the frame kernels scale every row / tile by
        # a power of two before they split it, so any finite input is in range; non-finite output samples (inf / NaN in
        # the input, a true fp32 overflow) reach the caller AS THEY ARE, like from the reference's plain-fp32 forward
        # (`keep_nonfinite`, ABI 13: nothing is hidden even when nobody looks at the flag — a `Streamer` stores 0 instead,
        # silence for a listener) and raise THIS Net's flag word (`_range_flag`: pinned host memory the kernel stores to
        # directly, one per device, never shared with another Net or a Streamer).
        #   range_check = True / "deferred" (default): the forward stays asynchronous; the word is looked at when the NEXT
        #       forward starts and in `range_status()`, and a set word raises LH_ERR_RANGE there.  (Round 3 ended every
        #       forward with a host wait: 7.58 -> 7.36 ms per batch-32 step without it, profiles/r04e_range_check_sync_cost.txt.)
        #   range_check = "sync": wait for the stream at the end of every forward and raise from the forward that
        #       produced the samples (LOOKONCE_RANGE_CHECK=sync); False / LOOKONCE_RANGE_CHECK=0: never look.
        rc = os.environ.get("LOOKONCE_RANGE_CHECK", "1")
        self.range_check = False if rc == "0" else ("sync" if rc == "sync" else True)
        self._range_flags: Dict[str, torch.Tensor] = {}
        # fused LSTM + Linear + residual kernels (f16x3 mode only); LOOKONCE_FUSE=0 selects the unfused pair
        self.fuse_linear = os.environ.get("LOOKONCE_FUSE", "1") != "0"
        # fused intra kernel (two launches, one per direction) from here on, the unfused pair (both directions concurrently) below:
        # 8192 frames in rounds 1-5; with time windows the crossover is lower (profiles/r06h: B = 13 x 625 frames 3.40 -> 3.08 ms,
        # B = 10 2.66 -> 2.60, B = 8 2.32 -> 2.36)
        self.fuse_intra_min_frames = int(os.environ.get("LOOKONCE_FUSE_INTRA_MIN_FRAMES", "6000"))
        # inter LSTM: one workgroup per sequence (mat-vec recurrence, 0.39 us per step) while the sequences fit ONE round of CUs
        # (batch <= 2); above, the 16-sequence tile kernel in time windows wins (round 6, profiles/r06h: B = 4 2.27 -> 1.67 ms;
        # rounds 3-5 used the per-sequence kernel up to two rounds = batch 5).  LOOKONCE_MATVEC_MAX_SEQS overrides.
        self.inter_matvec_max_seqs = int(os.environ.get("LOOKONCE_MATVEC_MAX_SEQS", "256"))
        self.stream_intra_max_frames = 128      # up to here one workgroup per (frame, direction) still finds its own CU
        # time-axis pipelining of the three blocks (round 6; include/lookonce_hip.h "time windows"): every stage is causal in
        # t, so the frames can be cut into K windows, window k on its own HIP stream, and block i on window k + 1 runs beside
        # block i + 1 on window k.  State is handed over on the device: (h, c) per window, the K / V history = the previous
        # window's rows of the history-extended buffers (one pair per block, so a window that runs ahead cannot overwrite rows a
        # slower one still reads); the output is bit-identical to the whole-clip forward.  What it buys depends on how much of
        # the chip one launch fills (profiles/r06b, r06e, r06h): the inter LSTM is a 625-step dependent chain on ceil(97 B / 16)
        # CUs, and every small-batch launch is latency-bound —
        #   time_chunks        batches on the fused kernels (B T >= 6000 frames: B >= 10 at 5 s).  0 = automatic: 2 windows while
        #                      the inter launch leaves more than ~30 % of the 256 CUs dark (B <= 29: -19 % at B = 14, -15 % at 16,
        #                      -11 % at 20, -5 % at 24, -1 % at 28), one from there on (B = 32: +-0, and the per-launch figures of
        #                      bench.py's roofline object stay those of kernels that run alone); K >= 1 forces K.
        #   time_chunks_small  smaller batches (unfused intra pair; per-sequence inter kernel up to B = 5, tiled one above),
        #                      windows on multiples of the attention tile / of 64 frames.  0 = automatic: 3 windows (B = 6 … 13:
        #                      -21 … -27 %), 2 for one utterance (1.31 -> 1.12 ms per 5 s clip; every hand-over between windows is
        #                      a cross-stream event of 10-20 us on this runtime), 1 where the per-sequence kernel's last round of
        #                      workgroups already fills the chip (B = 2, 5); K >= 1 forces K.
        # LOOKONCE_TIME_CHUNKS / LOOKONCE_TIME_CHUNKS_SMALL override.
        self.n_cus = 256                        # MI355X (the library is gfx950-only)
        self.time_chunks = int(os.environ.get("LOOKONCE_TIME_CHUNKS", "0"))
        self.time_chunks_small = int(os.environ.get("LOOKONCE_TIME_CHUNKS_SMALL", "0"))
        self.chunk_min_frames = 64              # >= the 49 frames of attention history: a window then depends on ONE predecessor
        self._chunk_streams: Dict[str, list] = {}
        self._pack_key = None
        self._packed = None
        self._ws: Dict[tuple, dict] = {}
        self._zeros: Dict[tuple, dict] = {}
        self._blob = None                  # (device, packed tree) when built by `from_packed`
        self._debug_taps: Optional[dict] = None
        self._prof: Optional[list] = None  # bench.py: list of (kernel tag, start event, end event) per launch
        self._prof_only: Optional[set] = None
        self._prof_stride: int = 1              # bench.py: bracket every n-th call of the selected names only (see bench.py)
        self._prof_count: dict = {}
        # asteroid-filterbanks' STFTFB may register a second buffer (`torch_window`) next to `_filters` (un-vendored,
        # version unpinned: could not be checked here).  The kernels only need `_filters`, so such keys of a reference
        # checkpoint are dropped before the strict key check instead of failing it.
        self._register_load_state_dict_pre_hook(self._drop_foreign_filterbank_keys)

    @staticmethod
    def _drop_foreign_filterbank_keys(state_dict, prefix, *_):
        for k in [k for k in state_dict if k.startswith(prefix) and ".filterbank." in k and not k.endswith("._filters")]:
            state_dict.pop(k)

    # ------------------------------------------------------------------------------------------------
    # reference API
    # ------------------------------------------------------------------------------------------------
    def init_buffers(self, batch_size, device):
        """Zero streaming state, reference shapes (tfgridnet_causal.py:173-186, 408-427)."""
        z = lambda *s: torch.zeros(*s, device=device)
        F_, C_, L_ = self.n_freqs, self.emb_dim, self.local_atten_len
        bufs = {}
        for i in range(self.n_blocks):
            bufs[f"buf{i}"] = dict(K_buf=z(batch_size * self.n_head, L_ - 1, self.E * F_),
                                   V_buf=z(batch_size * self.n_head, L_ - 1, self.V_dim * F_),
                                   c0=z(1, batch_size * F_, self.hidden),
                                   h0=z(1, batch_size * F_, self.hidden))
        return dict(conv_buf=z(batch_size, self.num_ch * 2, 2, F_), deconv_buf=z(batch_size, C_, 2, F_),
                    istft_buf=z(batch_size, self.n_srcs, F_ * 2, 1), gridnet_bufs=bufs)

    def predict(self, x, embed, input_state, pad=True, want_state=True):
        """Reference signature (net.py:54-66) plus `want_state`: False (only used by `forward` when it starts from the
        zero state) skips materialising the next state, which `forward` discards anyway."""
        mod = 0
        if pad:
            pad_size = (0, self.stft_pad_size) if self.lookahead else (0, 0)
            x, mod = mod_pad(x, chunk_size=self.stft_chunk_size, pad=pad_size)
        x, next_state = self._separate(x, embed, input_state, want_state)
        # the kernels already leave out the stft_pad_size look-ahead tail the reference trims at net.py:61
        if mod != 0:
            x = x[:, :, :-mod]
        return x, next_state

    def forward(self, x, embeds, input_state=None, pad=True):
        embeds = embeds[:, 0]  # [B, E]
        # zero initial state and the next state is dropped (reference net.py:68-76): no per-call state tensors
        x, next_state = self.predict(x, embeds, input_state, pad, want_state=input_state is not None)
        return x

    def make_streamer(self, batch_size: int, device, use_graph: bool = True):
        """Chunked real-time front end (BASELINE configs[1]): see `Streamer`."""
        return Streamer(self, batch_size, device, use_graph)

    # ------------------------------------------------------------------------------------------------
    # host-side plumbing
    # ------------------------------------------------------------------------------------------------
    @classmethod
    def from_packed(cls, path: str, device) -> "Net":
        """A separator whose weights come ONLY from a packed blob (`checkpoint.export_packed`, format in
        include/lookonce_weights.h): the blob's payload is uploaded in one copy and its tensors are handed to the C ABI
        as they are.  The parameter tree is dropped (nothing to pack, nothing to load) — this is the host a Python-free
        caller of the C ABI would write, with the launch sequence of `_separate`."""
        from . import checkpoint
        meta, flat = checkpoint.import_packed(path, device)
        if meta["model"] != "separator":
            raise ValueError(f"{path}: holds a {meta['model']!r} blob")
        if meta["abi_version"] != _cabi.ABI_VERSION:
            raise ValueError(f"{path}: packed for ABI v{meta['abi_version']}, library is v{_cabi.ABI_VERSION}")
        net = cls(**meta["params"]).eval()
        net.tfgridnet = None                                     # no parameters: the blob is the only weight source
        tree: dict = {"blocks": [dict() for _ in range(net.n_blocks)]}
        for name, t in flat.items():
            if name.startswith("blocks."):
                _, i, key = name.split(".", 2)
                tree["blocks"][int(i)][key] = t
            else:
                tree[name] = t
        net._blob = (next(iter(flat.values())).device, tree)
        return net

    def _weights(self, device) -> dict:
        if self._blob is not None:
            if torch.device(device) != self._blob[0]:
                raise RuntimeError(f"packed weights live on {self._blob[0]}, input on {device}")
            return self._blob[1]
        tensors = list(self.parameters()) + list(self.buffers())
        key = (str(device),) + tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self._pack_key:
            sd = {k: v for k, v in self.state_dict(keep_vars=True).items()}
            for k, v in sd.items():
                if v.device != device:
                    raise RuntimeError(f"parameter {k} lives on {v.device}, input on {device}")
            with torch.no_grad():
                self._packed = pack_all(sd, self.n_blocks)
            self._pack_key = key
        return self._packed

    def _workspace(self, B, T, device) -> dict:
        key = (B, T, str(device))
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) > 4:
                self._ws.clear()
            F_, C_, nh = self.n_freqs, self.emb_dim, self.n_head
            e = lambda *s: torch.empty(*s, device=device, dtype=torch.float32)
            hist = self.local_atten_len - 1
            # q / kx / vx: split-precision fp16 rows (include/lookonce_hip.h); kx / vx carry KV_PAD_ROWS zero rows that
            # only this allocation ever writes
            z16 = lambda *s: torch.zeros(*s, device=device, dtype=torch.float16)
            ws = dict(xa=e(B, T, F_, C_), xb=e(B, T, F_, C_), xc=e(B, T, F_, C_), hbuf=e(B * T * F_, 2 * self.hidden),
                      q=z16(B * nh, T, 2 * QK_PAD), kx=z16(B * nh, T + hist + KV_PAD_ROWS, 2 * QK_PAD),
                      vx=z16(B * nh, T + hist + KV_PAD_ROWS, 2 * self.V_dim * F_), gain=e(B, F_, C_),
                      gain_raw=e(B, F_ * C_), hist_dirty=False)
            self._ws[key] = ws
        return ws

    @staticmethod
    def _dev_key(device) -> str:
        """'cuda' / torch.device('cuda') and 'cuda:N' of the current device are ONE key (`range_status('cuda')`
        must find the flag the forward keyed by `x.device` = 'cuda:0')."""
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        return str(dev)

    def _range_flag(self, device) -> torch.Tensor:
        """This Net's two-word range flag for `device` (include/lookonce_hip.h: [0] sticky, [1] unused here): pinned host
        memory — device-accessible, and readable by the host without a copy or a launch."""
        key = self._dev_key(device)
        if key not in self._range_flags:
            self._range_flags[key] = self._flag_words(torch.device(key))
        return self._range_flags[key]

    def range_status(self, device=None) -> bool:
        """Waits for the current stream of `device`, then reads and clears this Net's range flag: True when a forward since
        the last look produced non-finite samples.  Call it after the last forward of a loop (a set flag otherwise raises at
        the start of the next forward).  `device=None`: EVERY device this Net has run on (it used to look at the
        first one only)."""
        if device is None:
            keys = list(self._range_flags)
        else:
            keys = [self._dev_key(device)]
        bad = False
        for key in keys:
            dev = torch.device(key)
            flag = self._range_flag(dev)
            self._sync(dev)
            if int(flag[0]) != 0:
                flag.zero_()
                bad = True
        return bad

    def _sync(self, dev):
        torch.cuda.current_stream(dev).synchronize()

    _RANGE_MSG = ("LH_ERR_RANGE: {} produced non-finite samples. The frame kernels scale every row by a power of "
                  "two before the fp16 hi + lo split, so this means inf / NaN in the input or the state, or an activation "
                  "beyond the fp32 range itself (include/lookonce_hip.h, range contract).")

    def _zero_state(self, B, device) -> dict:
        key = (B, str(device))
        if key not in self._zeros:
            if len(self._zeros) > 4:
                self._zeros.clear()
            self._zeros[key] = self.init_buffers(B, device)        # read-only: the kernels write state to separate tensors
        return self._zeros[key]

    def _separate(self, x: torch.Tensor, embed: torch.Tensor, state: Optional[dict], want_state: bool = True):
        """TFGridNet.forward (reference tfgridnet_causal.py:188-283) on the HIP kernels."""
        if self.training and torch.is_grad_enabled():
            raise RuntimeError("lookoncetohear_amd.Net is an inference-only drop-in (forward kernels, no autograd): call "
                               ".eval() and/or run under torch.no_grad(); training stays on the reference model")
        self._check_gemm_mode()                 # before the first launch, not mid-forward
        if self.gemm_mode == "f32all":
            return self._separate_ref32(x, embed, state, want_state)
        lib = self._lib(x)
        dev = x.device
        if self.range_check and int(self._range_flag(dev)[0]) != 0:       # deferred look at the previous forwards' flag: a host read
            self._range_flag(dev).zero_()
            raise RuntimeError(self._RANGE_MSG.format("an earlier forward of this Net"))
        hop, nfft = self.stft_chunk_size, self.nfft
        assert x.dim() == 3 and x.shape[1] == self.num_ch, "input must be [B, num_ch, N]"
        Bn, _, n = x.shape
        from_zero = state is None and not want_state
        if state is None:
            state = self._zero_state(Bn, dev) if from_zero else self.init_buffers(Bn, dev)
        T = (n - nfft) // hop + 1
        if T < 1:
            raise ValueError(f"need at least {nfft} samples, got {n}")
        ns = (T - 1) * hop + nfft
        x = x[..., :ns].contiguous().float()
        embed = embed.contiguous().float()
        F_, C_, nh, H_ = self.n_freqs, self.emb_dim, self.n_head, self.hidden
        hist = self.local_atten_len - 1
        # raw launches go to the CURRENT HIP device: make it the tensors' device (the reference eval driver builds
        # `cuda:N` tensors without torch.cuda.set_device, src/ts_hear_test.py:175)
        with torch.no_grad(), self._device_ctx(x):
            pk = self._weights(dev)
            ws = self._workspace(Bn, T, dev)
            st = self._stream(dev)
            P = lambda t: t.data_ptr()
            new = lambda t: torch.empty_like(t, dtype=torch.float32)
            c32 = lambda t: t.contiguous().float()
            taps = self._debug_taps
            xa, xb, xc, hbuf = ws["xa"], ws["xb"], ws["xc"], ws["hbuf"]
            prof = self._prof
            only = self._prof_only                   # bench.py: restrict the event pairs to these C-ABI calls
            stride, count = self._prof_stride, self._prof_count

            class _Timed:                       # HIP events on the launch stream around each C-ABI call
                @staticmethod
                def call(fn, *args):
                    name = fn[:-4] if fn.endswith("_win") else fn       # a windowed call counts as its stage
                    if prof is None or (only is not None and name not in only):
                        return lib_.call(fn, *args)
                    if stride > 1:                  # sampled bracket: an event record costs ~6 us of stream time
                        n = count.get(name, 0)
                        count[name] = n + 1
                        if n % stride:
                            return lib_.call(fn, *args)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    lib_.call(fn, *args)
                    e1.record()
                    prof.append((name, e0, e1))

            lib_, lib = lib, _Timed

            conv_in = c32(state["conv_buf"]); conv_out = new(conv_in)
            lib.call("lh_stft_conv_in", P(x), P(conv_in), P(conv_out), P(pk["wfb_t"]), P(pk["conv_w"]),
                     P(pk["conv_b"]), P(xa), Bn, T, ns, st)
            lib.call("lh_embed_proj_ln", P(embed), P(pk["emb_w"]), P(pk["emb_b"]), P(pk["emb_ln_w"]),
                     P(pk["emb_ln_b"]), P(ws["gain_raw"]), P(ws["gain"]), Bn, st)
            if taps is not None:
                taps["Z0"], taps["G"] = xa.clone(), ws["gain"].clone()

            rows = Bn * T * F_
            mode = 1 if self.gemm_mode == "f16x3" else 0
            wkey, bkey = ("_w16", "_b16") if mode else ("_w", "_b")
            K = self._n_time_chunks(Bn, T, mode)
            if K > 1:
                self._blocks_chunked(lib, pk, ws, state, Bn, T, dev, from_zero, want_state, K)
            for i in range(self.n_blocks if K == 1 else 0):
                bp = pk["blocks"][i]
                bs = state["gridnet_bufs"][f"buf{i}"]
                h0, c0 = c32(bs["h0"]), c32(bs["c0"])
                hN, cN = new(h0), new(c0)
                fuse = mode == 1 and self.fuse_linear
                # the fused intra path runs the two directions as consecutive launches (the reverse one accumulates into
                # the forward one's rows): worth it once a single direction fills the GPU, otherwise the unfused kernel
                # (both directions concurrently) has half the latency (streaming, batch 1)
                if fuse and Bn * T >= self.fuse_intra_min_frames:
                    lib.call("lh_intra_block", P(xa), P(bp["intra_w16"]), P(bp["intra_b16"]), P(bp["intra_lin_w2"]),
                             P(bp["intra_lin_b"]), P(xb), Bn * T, st)
                elif mode == 1 and Bn * T <= self.stream_intra_max_frames:
                    # a handful of frames (streaming): one workgroup per (frame, direction), mat-vec recurrence
                    lib.call("lh_intra_stream", P(xa), P(bp["intra_s_wih"]), P(bp["intra_s_b"]), P(bp["intra_s_whh"]),
                             P(hbuf), Bn * T, st)
                    lib.call("lh_linear_res", P(hbuf), P(bp["intra_lin_w"]), P(bp["intra_lin_b"]), P(xa), P(xb), rows,
                             2 * H_, st)
                else:
                    # intra: LN + BiLSTM over frequency -> Linear(128->64) + residual
                    lib.call("lh_ln_lstm_intra", P(xa), P(bp["intra_ln_w"]), P(bp["intra_ln_b"]), P(bp["intra" + wkey]),
                             P(bp["intra" + bkey]), P(hbuf), Bn * T, mode, st)
                    lib.call("lh_linear_res", P(hbuf), P(bp["intra_lin_w"]), P(bp["intra_lin_b"]), P(xa), P(xb), rows,
                             2 * H_, st)
                if fuse and Bn * F_ <= self.inter_matvec_max_seqs and T >= 32:
                    # few sequences, many steps (batch 1 offline): one workgroup per sequence, mat-vec recurrence
                    lib.call("lh_inter_matvec", P(xb), P(bp["inter_s_wih"]), P(bp["inter_s_b"]), P(bp["inter_s_whh"]),
                             P(bp["inter_lin_w"]), P(bp["inter_lin_b"]), P(h0), P(c0), P(hN), P(cN), P(xc), Bn, T, st)
                elif fuse:
                    lib.call("lh_inter_block", P(xb), P(bp["inter_w8"]), P(bp["inter_b16"]), P(bp["inter_lin_wu"]),
                             P(bp["inter_lin_b"]), P(h0), P(c0), P(hN), P(cN), P(xc), Bn, T, st)
                else:
                    # inter: LN + causal LSTM over time with carried state -> Linear(64->64) + residual
                    lib.call("lh_ln_lstm_inter", P(xb), P(bp["inter_ln_w"]), P(bp["inter_ln_b"]), P(bp["inter" + wkey]),
                             P(bp["inter" + bkey]), P(h0), P(c0), P(hN), P(cN), P(hbuf), Bn, T, mode, st)
                    lib.call("lh_linear_res", P(hbuf), P(bp["inter_lin_w"]), P(bp["inter_lin_b"]), P(xb), P(xc), rows,
                             H_, st)
                # attention: history rows in, Q/K/V, local attention (head merge fused), projection + LN + residual
                if not from_zero:
                    lib.call("lh_ring_pack", P(c32(bs["K_buf"])), P(c32(bs["V_buf"])), P(ws["kx"]), P(ws["vx"]), Bn, T, st)
                    ws["hist_dirty"] = True
                elif ws["hist_dirty"]:                      # history rows of the window-extended buffers back to zero
                    ws["kx"][:, :hist].zero_()
                    ws["vx"][:, :hist].zero_()
                    ws["hist_dirty"] = False
                lib.call("lh_qkv_proj_ln", P(xc), P(bp["qkv_w"]), P(bp["qkv_b"]), P(bp["qkv_slopes"]), P(bp["lnq_w"]),
                         P(bp["lnq_b"]), P(bp["lnk_w"]), P(bp["lnk_b"]), P(bp["lnv_w"]), P(bp["lnv_b"]), P(ws["q"]),
                         P(ws["kx"]), P(ws["vx"]), None, Bn, T, st)
                lib.call("lh_local_attn", P(ws["q"]), P(ws["kx"]), P(ws["vx"]), P(xb), Bn, T, st)
                if taps is not None:
                    taps[f"blocks.{i}.attn"] = xb.clone()          # head-major [B][T][4][97][16]
                gain = ws["gain"] if (i == 0 and self.n_blocks > 1) else None   # `batch * embed` before block 1
                lib.call("lh_proj_ln_res", P(xb), P(bp["proj_w"]), P(bp["proj_b"]), P(bp["proj_slope"]),
                         P(bp["proj_ln_w"]), P(bp["proj_ln_b"]), P(xc), P(gain) if gain is not None else None, P(xa),
                         Bn, T, st)
                if want_state:
                    bs["h0"], bs["c0"] = hN, cN
                    bs["K_buf"] = torch.empty(Bn * nh, hist, self.E * F_, device=dev, dtype=torch.float32)
                    bs["V_buf"] = torch.empty(Bn * nh, hist, self.V_dim * F_, device=dev, dtype=torch.float32)
                    lib.call("lh_ring_unpack", P(ws["kx"]), P(ws["vx"]), P(bs["K_buf"]), P(bs["V_buf"]), Bn, T, st)
                if taps is not None:
                    taps[f"blocks.{i}.Y2"] = xc.clone()
                    taps[f"blocks.{i}.Q"] = unsplit_qk(ws["q"], self.E * F_)
                    taps[f"blocks.{i}.K"] = unsplit_qk(ws["kx"][:, hist:hist + T], self.E * F_)
                    taps[f"blocks.{i}.V"] = unsplit_v(ws["vx"][:, hist:hist + T])
                    taps[f"blocks.{i}.out"] = xa.clone()

            dec_in, ist_in = c32(state["deconv_buf"]), c32(state["istft_buf"])
            dec_out, ist_out = new(dec_in), new(ist_in)
            y = torch.empty(Bn, self.n_srcs, hop * T, device=dev, dtype=torch.float32)
            flag = self._range_flag(dev)
            lib.call("lh_deconv_istft", P(xa), P(dec_in), P(dec_out), P(ist_in), P(ist_out), P(pk["deconv_w"]),
                     P(pk["deconv_b"]), P(pk["wfb_dec"]), P(y), P(flag), 1, Bn, T, st)
            if want_state:
                state["conv_buf"], state["deconv_buf"], state["istft_buf"] = conv_out, dec_out, ist_out
            if self.range_check == "sync" and not self._capturing():
                if self.range_status(dev):
                    raise RuntimeError(self._RANGE_MSG.format("this forward"))
        return y, (state if want_state else None)

    def _n_time_chunks(self, Bn: int, T: int, mode: int) -> int:
        """Windows the block loop is cut into (see `time_chunks` / `time_chunks_small` in __init__); 1 for streaming-size
        inputs, stage taps and the exact-fp32 modes.  Every window keeps at least `chunk_min_frames` (>= the 49 frames of
        attention history: a window then depends on ONE predecessor)."""
        if mode != 1 or not self.fuse_linear or self._debug_taps is not None:
            return 1
        if Bn * self.n_freqs <= self.inter_matvec_max_seqs and T < 32:
            return 1                                             # (the per-sequence inter kernel serves T >= 32, `_separate`)
        if Bn * T < self.fuse_intra_min_frames:                  # small batches: unfused intra pair
            if Bn * T <= self.stream_intra_max_frames:
                return 1                                         # a handful of frames: the streaming intra kernel
            K = int(self.time_chunks_small)
            if K == 0:                                           # automatic (profiles/r06h_batch_sweep_windows.txt)
                nseq = Bn * self.n_freqs
                K = 3
                if nseq <= self.inter_matvec_max_seqs:
                    # per-sequence inter kernel (one workgroup per sequence, B <= 2): windows only pay while its workgroups leave
                    # CUs dark (B = 1: 97 of 256 -> -14 %; B = 2: 194 -> +5 %); one utterance is best in two (three: 1.18 against
                    # 1.12 ms)
                    tail = nseq % self.n_cus or self.n_cus
                    K = 1 if tail >= 0.6 * self.n_cus else (2 if nseq <= self.n_cus else 3)
        else:
            K = int(self.time_chunks)
            if K == 0:                                           # automatic
                tiles = (Bn * self.n_freqs + 15) // 16           # workgroups (= CUs) of the inter launch
                K = 2 if tiles < 0.7 * self.n_cus else 1
            while K > 1 and Bn * (T // K) < self.fuse_intra_min_frames // 2:
                K -= 1
        return max(min(K, T // max(self.chunk_min_frames, 1)), 1)

    def _window_bounds(self, Bn: int, T: int, K: int) -> list:
        """Window boundaries [0, ..., T]: even cuts, moved to multiples of the attention kernel's query-tile length (40 frames
        once the whole-clip launch exceeds 512 workgroups, else 32: lh_attn.hip `lh_local_attn_win`) when the windows are long
        enough — the windows then cut the time axis into exactly the tiles of the whole-clip launch and the forward is
        bit-identical to the unchunked one (every other stage is per frame or per sequence)."""
        bh8 = (Bn * self.n_head + 7) // 8 * 8
        align = 40 if bh8 * ((T + 31) // 32) > 512 else 32
        if Bn * self.n_freqs <= self.inter_matvec_max_seqs:
            align = 64                                        # lh_inter_matvec's chunk of steps (a multiple of the 32-frame tile)
        cuts = [(k * T) // K for k in range(K + 1)]
        if T // K >= 2 * align or (align == 64 and T // K >= 64):
            cuts = [0] + [int(round(k * T / K / align)) * align for k in range(1, K)] + [T]
        return cuts

    def _lanes(self, dev, K) -> _Lanes:
        # ONE pool of side streams per device, grown on demand and shared by every window count: the process has four hardware
        # queues, and every further stream shares one with somebody (a bench run that had left six window streams behind slowed
        # the EMBEDDER's two-stream forward from 59.7 to 68.4 ms: its side stream had landed on a busy queue)
        pool = self._chunk_streams.setdefault(str(dev), [])
        while len(pool) < K - 1:
            pool.append(torch.cuda.Stream(device=dev))
        if self.chunk_min_frames < self.local_atten_len - 1:
            raise ValueError("chunk_min_frames must cover the attention history (49 frames) on concurrent streams")
        return _Lanes(dev, pool[:K - 1])

    def _blocks_chunked(self, lib, pk, ws, state, Bn, T, dev, from_zero, want_state, K):
        """The three GridNet blocks (tfgridnet_causal.py:489-590) with the time axis cut into K windows, window k on stream k:
        for block i and window k the five stage launches of `_separate` through the `_win` entry points, with two cross-stream
        dependences per (block, window) — the inter LSTM's (h, c) and the K / V rows of the previous window."""
        F_, nh, hist = self.n_freqs, self.n_head, self.local_atten_len - 1
        P = lambda t: t.data_ptr()
        c32 = lambda t: t.contiguous().float()
        xa, xb, xc = ws["xa"], ws["xb"], ws["xc"]
        lanes = self._lanes(dev, K)
        # the whole-clip rules of `_separate`, per batch: which intra / inter kernels the windows run
        unfused_intra = Bn * T < self.fuse_intra_min_frames
        matvec_inter = Bn * F_ <= self.inter_matvec_max_seqs
        if "kxb" not in ws:                                   # one history-extended K / V pair per block
            z16 = lambda *s_: torch.zeros(*s_, device=dev, dtype=torch.float16)
            ws["kxb"] = [ws["kx"]] + [z16(*ws["kx"].shape) for _ in range(self.n_blocks - 1)]
            ws["vxb"] = [ws["vx"]] + [z16(*ws["vx"].shape) for _ in range(self.n_blocks - 1)]
            ws["hist_dirty_b"] = [False] * self.n_blocks
        bounds = self._window_bounds(Bn, T, K)
        # per block: (h, c) of every window boundary; history rows in place (on the current stream, before the fork)
        hs, cs = [], []
        for i in range(self.n_blocks):
            bs = state["gridnet_bufs"][f"buf{i}"]
            h0, c0 = c32(bs["h0"]).reshape(-1, self.hidden), c32(bs["c0"]).reshape(-1, self.hidden)
            hs.append([h0] + [torch.empty_like(h0) for _ in range(K)])
            cs.append([c0] + [torch.empty_like(c0) for _ in range(K)])
            kx, vx = ws["kxb"][i], ws["vxb"][i]
            if not from_zero:
                lib.call("lh_ring_pack", P(c32(bs["K_buf"])), P(c32(bs["V_buf"])), P(kx), P(vx), Bn, T, self._stream(dev))
                ws["hist_dirty_b"][i] = True
                if i == 0:
                    ws["hist_dirty"] = True                   # block 0's pair is the unchunked path's one pair
            elif ws["hist_dirty_b"][i] or (i == 0 and ws["hist_dirty"]):
                kx[:, :hist].zero_()
                vx[:, :hist].zero_()
                ws["hist_dirty_b"][i] = False
                if i == 0:
                    ws["hist_dirty"] = False
        lanes.fork()
        for i in range(self.n_blocks):
            bp = pk["blocks"][i]
            kx, vx = ws["kxb"][i], ws["vxb"][i]
            gain = ws["gain"] if (i == 0 and self.n_blocks > 1) else None
            ev_r = ev_q = None                                # the previous window's (h, c) / K, V rows of THIS block
            for k in range(K):
                t0, Tc = bounds[k], bounds[k + 1] - bounds[k]
                carry = (1 if k > 0 else 0) | (2 if k + 1 < K else 0)     # inner boundaries: internal cell-state form
                with lanes.on(k):
                    st = self._stream(dev)
                    if unfused_intra:
                        lib.call("lh_ln_lstm_intra_win", P(xa), P(bp["intra_ln_w"]), P(bp["intra_ln_b"]), P(bp["intra_w16"]),
                                 P(bp["intra_b16"]), P(ws["hbuf"]), Bn, T, t0, Tc, st)
                        lib.call("lh_linear_res_win", P(ws["hbuf"]), P(bp["intra_lin_w"]), P(bp["intra_lin_b"]), P(xa), P(xb),
                                 Bn, T, t0, Tc, 2 * self.hidden, st)
                    else:
                        lib.call("lh_intra_block_win", P(xa), P(bp["intra_w16"]), P(bp["intra_b16"]), P(bp["intra_lin_w2"]),
                                 P(bp["intra_lin_b"]), P(xb), Bn, T, t0, Tc, st)
                    if k > 0:
                        lanes.wait(k, ev_r)
                    if matvec_inter:
                        lib.call("lh_inter_matvec_win", P(xb), P(bp["inter_s_wih"]), P(bp["inter_s_b"]), P(bp["inter_s_whh"]),
                                 P(bp["inter_lin_w"]), P(bp["inter_lin_b"]), P(hs[i][k]), P(cs[i][k]), P(hs[i][k + 1]),
                                 P(cs[i][k + 1]), P(xc), Bn, T, t0, Tc, carry, st)
                    else:
                        lib.call("lh_inter_block_win", P(xb), P(bp["inter_w8"]), P(bp["inter_b16"]), P(bp["inter_lin_wu"]),
                                 P(bp["inter_lin_b"]), P(hs[i][k]), P(cs[i][k]), P(hs[i][k + 1]), P(cs[i][k + 1]), P(xc), Bn, T,
                                 t0, Tc, carry, st)
                    if k + 1 < K:
                        ev_r = lanes.signal(k)
                    lib.call("lh_qkv_proj_ln_win", P(xc), P(bp["qkv_w"]), P(bp["qkv_b"]), P(bp["qkv_slopes"]), P(bp["lnq_w"]),
                             P(bp["lnq_b"]), P(bp["lnk_w"]), P(bp["lnk_b"]), P(bp["lnv_w"]), P(bp["lnv_b"]), P(ws["q"]),
                             P(kx), P(vx), None, Bn, T, t0, Tc, st)
                    ev_prev = ev_q
                    if k + 1 < K:
                        ev_q = lanes.signal(k)
                    if k > 0:
                        lanes.wait(k, ev_prev)
                    lib.call("lh_local_attn_win", P(ws["q"]), P(kx), P(vx), P(xb), Bn, T, t0, Tc, st)
                    lib.call("lh_proj_ln_res_win", P(xb), P(bp["proj_w"]), P(bp["proj_b"]), P(bp["proj_slope"]),
                             P(bp["proj_ln_w"]), P(bp["proj_ln_b"]), P(xc), P(gain) if gain is not None else None, P(xa),
                             Bn, T, t0, Tc, st)
        lanes.join()
        if want_state:
            for i in range(self.n_blocks):
                bs = state["gridnet_bufs"][f"buf{i}"]
                bs["h0"], bs["c0"] = hs[i][K].reshape(1, -1, self.hidden), cs[i][K].reshape(1, -1, self.hidden)
                bs["K_buf"] = torch.empty(Bn * nh, hist, self.E * F_, device=dev, dtype=torch.float32)
                bs["V_buf"] = torch.empty(Bn * nh, hist, self.V_dim * F_, device=dev, dtype=torch.float32)
                lib.call("lh_ring_unpack", P(ws["kxb"][i]), P(ws["vxb"][i]), P(bs["K_buf"]), P(bs["V_buf"]), Bn, T,
                         self._stream(dev))

    def _check_gemm_mode(self):
        if self.gemm_mode not in ("f32rec", "f32all", "f16x3"):
            raise ValueError(f"gemm_mode / LOOKONCE_GEMM must be 'f16x3', 'f32rec' or 'f32all', got {self.gemm_mode!r}"
                             + (" ('f32' was the pre-round-5 spelling of 'f32rec')" if self.gemm_mode == "f32" else ""))

    def _capturing(self) -> bool:
        return torch.cuda.is_current_stream_capturing()

    def _separate_ref32(self, x: torch.Tensor, embed: torch.Tensor, state: Optional[dict], want_state: bool = True):
        """`gemm_mode = "f32all"`: the same forward with EVERY contraction in plain fp32 — the exact fp32-MFMA recurrences
        (`lh_ln_lstm_intra / _inter`, LH_GEMM_F32) between the plain-fp32 reference kernels of lh_ref32.hip for the frame
        stages (weights straight from the state dict, Q / K / V as fp32 rows).  Slow (~50x), allocation-happy, test /
        diagnosis only: the in-tree A/B that tells split-precision error from a kernel bug (reference arithmetic:
        tfgridnet_causal.py:188-283 is fp32 end to end)."""
        if self._blob is not None:
            raise RuntimeError("gemm_mode='f32all' reads the parameter tree; a `from_packed` Net has none")
        lib = self._lib(x)
        dev = x.device
        hop, nfft = self.stft_chunk_size, self.nfft
        assert x.dim() == 3 and x.shape[1] == self.num_ch, "input must be [B, num_ch, N]"
        Bn, _, n = x.shape
        if state is None:
            state = self.init_buffers(Bn, dev)
        T = (n - nfft) // hop + 1
        if T < 1:
            raise ValueError(f"need at least {nfft} samples, got {n}")
        ns = (T - 1) * hop + nfft
        x = x[..., :ns].contiguous().float()
        embed = embed.contiguous().float()
        F_, C_, nh, H_ = self.n_freqs, self.emb_dim, self.n_head, self.hidden
        hist = self.local_atten_len - 1
        rows = Bn * T * F_
        with torch.no_grad(), self._device_ctx(x):
            pk = self._weights(dev)                    # only the fp32-MFMA LSTM images and the speaker-gain tensors
            sd = {k: v.detach().float().contiguous() for k, v in self.tfgridnet.state_dict().items()}
            st = self._stream(dev)
            P = lambda t: t.data_ptr()
            e = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
            c32 = lambda t: t.contiguous().float()
            xa, xb, xc, hbuf = e(Bn, T, F_, C_), e(Bn, T, F_, C_), e(Bn, T, F_, C_), e(rows, 2 * H_)
            gain_raw, gain = e(Bn, F_ * C_), e(Bn, F_, C_)
            conv_in = c32(state["conv_buf"]); conv_out = torch.empty_like(conv_in)
            spec_s, sx_s = e(Bn, 2 * self.num_ch, T + 2, F_), e(Bn, self.n_srcs, T + 1, 2 * F_)   # scratch: named, so they outlive the calls
            lib.call("lh_ref32_stft_conv_in", P(x), P(conv_in), P(conv_out), P(sd["enc.filterbank._filters"]),
                     P(sd["conv.0.weight"]), P(sd["conv.0.bias"]), P(spec_s), P(xa), Bn, T, ns, st)
            lib.call("lh_embed_proj_ln", P(embed), P(pk["emb_w"]), P(pk["emb_b"]), P(pk["emb_ln_w"]), P(pk["emb_ln_b"]),
                     P(gain_raw), P(gain), Bn, st)                                   # plain fp32 in every mode
            taps = self._debug_taps
            if taps is not None:
                taps["Z0"], taps["G"] = xa.clone(), gain.clone()
            pre, rows_s, pre_s = e(rows, 112), e(rows, C_), e(rows, C_)
            q = e(Bn * nh, T, self.E * F_)
            for i in range(self.n_blocks):
                bp, bs, g = pk["blocks"][i], state["gridnet_bufs"][f"buf{i}"], f"blocks.{i}."
                h0, c0 = c32(bs["h0"]), c32(bs["c0"])
                hN, cN = torch.empty_like(h0), torch.empty_like(c0)
                lib.call("lh_ln_lstm_intra", P(xa), P(bp["intra_ln_w"]), P(bp["intra_ln_b"]), P(bp["intra_w"]), P(bp["intra_b"]),
                         P(hbuf), Bn * T, 0, st)
                lib.call("lh_ref32_linear", P(hbuf), P(sd[g + "intra_linear.weight"]), P(sd[g + "intra_linear.bias"]), None,
                         P(xa), P(xb), rows, 2 * H_, C_, st)
                lib.call("lh_ln_lstm_inter", P(xb), P(bp["inter_ln_w"]), P(bp["inter_ln_b"]), P(bp["inter_w"]), P(bp["inter_b"]),
                         P(h0), P(c0), P(hN), P(cN), P(hbuf), Bn, T, 0, st)
                lib.call("lh_ref32_linear", P(hbuf), P(sd[g + "inter_linear.weight"]), P(sd[g + "inter_linear.bias"]), None,
                         P(xb), P(xc), rows, H_, C_, st)
                # Q | K | V: Linear + PReLU (one launch per projection: each has its own slope), head split + LayerNorm
                kx, vx = e(Bn * nh, T + hist, self.E * F_), e(Bn * nh, T + hist, self.V_dim * F_)
                kx[:, :hist] = bs["K_buf"].float()
                vx[:, :hist] = bs["V_buf"].float()
                for nm, D, dst, row0, rpb in (("Q", self.E, q, 0, T), ("K", self.E, kx, hist, T + hist),
                                              ("V", self.V_dim, vx, hist, T + hist)):
                    a = g + f"attn_conv_{nm}."
                    ncol = nh * D
                    lib.call("lh_ref32_linear", P(xc), P(sd[a + "0.weight"]), P(sd[a + "0.bias"]), P(sd[a + "1.weight"]), None,
                             P(pre), rows, C_, ncol, st)
                    lib.call("lh_ref32_head_ln", P(pre), ncol, 0, D, P(sd[a + "3.norm.weight"]), P(sd[a + "3.norm.bias"]),
                             P(dst), row0, rpb, Bn, T, st)
                lib.call("lh_ref32_local_attn", P(q), P(kx), P(vx), P(xb), Bn, T, st)
                a = g + "attn_concat_proj."
                gn = gain if (i == 0 and self.n_blocks > 1) else None
                lib.call("lh_ref32_proj_ln_res", P(xb), P(sd[a + "0.weight"]), P(sd[a + "0.bias"]), P(sd[a + "1.weight"]),
                         P(sd[a + "3.norm.weight"]), P(sd[a + "3.norm.bias"]), P(xc), P(gn) if gn is not None else None,
                         P(rows_s), P(pre_s), P(xa), Bn, T, st)
                if want_state:
                    bs["h0"], bs["c0"] = hN, cN
                    bs["K_buf"], bs["V_buf"] = kx[:, T:T + hist].clone(), vx[:, T:T + hist].clone()
                if taps is not None:
                    taps[f"blocks.{i}.Y2"], taps[f"blocks.{i}.Q"] = xc.clone(), q.clone()
                    taps[f"blocks.{i}.K"], taps[f"blocks.{i}.V"] = kx[:, hist:].clone(), vx[:, hist:].clone()
                    taps[f"blocks.{i}.out"] = xa.clone()
            dec_in, ist_in = c32(state["deconv_buf"]), c32(state["istft_buf"])
            dec_out, ist_out = torch.empty_like(dec_in), torch.empty_like(ist_in)
            y = e(Bn, self.n_srcs, hop * T)
            lib.call("lh_ref32_deconv_istft", P(xa), P(dec_in), P(dec_out), P(ist_in), P(ist_out), P(sd["deconv.weight"]),
                     P(sd["deconv.bias"]), P(sd["dec.filterbank._filters"]), P(sx_s), P(y), Bn, T, st)
            if want_state:
                state["conv_buf"], state["deconv_buf"], state["istft_buf"] = conv_out, dec_out, ist_out
        return y, (state if want_state else None)

    # ------------------------------------------------------------------------------------------------
    # streaming fast path (Streamer)
    # ------------------------------------------------------------------------------------------------
    def _speaker_gain(self, embed: torch.Tensor, gain_raw: torch.Tensor, gain: torch.Tensor):
        """gain[b][f][c] = LayerNorm(Linear(embed)) (reference tfgridnet_causal.py:247-248) into preallocated tensors."""
        lib = self._lib(embed)
        pk = self._weights(embed.device)
        st = self._stream(embed.device)
        P = lambda t: t.data_ptr()
        with self._device_ctx(embed):
            lib.call("lh_embed_proj_ln", P(embed), P(pk["emb_w"]), P(pk["emb_b"]), P(pk["emb_ln_w"]), P(pk["emb_ln_b"]),
                     P(gain_raw), P(gain), embed.shape[0], st)

    def _stream_chunk(self, x, gain, sin: dict, sout: dict, rings, pos, y, pk: dict, ws: dict, flag=None):
        """One chunk of ONE frame for `Streamer`: the launches of `_separate` with every state tensor read from `sin` and
        written to `sout` (preallocated), the K / V history in per-block persistent rings, the speaker gain given.
        `pk` / `ws`: the packed weights and the T=1 workspace, OWNED by the caller — a captured graph holds raw pointers
        into them, so they must not be the entries `_weights` / `_workspace` may replace or evict later."""
        lib = self._lib(x)
        dev = x.device
        hop, nfft = self.stft_chunk_size, self.nfft
        Bn, _, n = x.shape
        T = (n - nfft) // hop + 1
        assert T == 1 and n == nfft, "the streaming path takes chunks of stft_chunk_size + stft_pad_size samples"
        F_, H_ = self.n_freqs, self.hidden
        st = self._stream(dev)
        P = lambda t: t.data_ptr()
        xa, xb, xc, hbuf = ws["xa"], ws["xb"], ws["xc"], ws["hbuf"]
        lib.call("lh_stft_conv_in", P(x), P(sin["conv_buf"]), P(sout["conv_buf"]), P(pk["wfb_t"]), P(pk["conv_w"]),
                 P(pk["conv_b"]), P(xa), Bn, T, n, st)
        for i in range(self.n_blocks):
            bp = pk["blocks"][i]
            kx, vx = rings[i]
            lib.call("lh_intra_stream", P(xa), P(bp["intra_s_wih"]), P(bp["intra_s_b"]), P(bp["intra_s_whh"]), P(hbuf),
                     Bn * T, st)
            lib.call("lh_linear_res", P(hbuf), P(bp["intra_lin_w"]), P(bp["intra_lin_b"]), P(xa), P(xb), Bn * T * F_,
                     2 * H_, st)
            lib.call("lh_inter_block", P(xb), P(bp["inter_w8"]), P(bp["inter_b16"]), P(bp["inter_lin_wu"]),
                     P(bp["inter_lin_b"]), P(sin["h"][i]), P(sin["c"][i]), P(sout["h"][i]), P(sout["c"][i]), P(xc), Bn, T, st)
            lib.call("lh_qkv_proj_ln", P(xc), P(bp["qkv_w"]), P(bp["qkv_b"]), P(bp["qkv_slopes"]), P(bp["lnq_w"]),
                     P(bp["lnq_b"]), P(bp["lnk_w"]), P(bp["lnk_b"]), P(bp["lnv_w"]), P(bp["lnv_b"]), P(ws["q"]),
                     P(kx), P(vx), P(pos), Bn, T, st)
            lib.call("lh_local_attn", P(ws["q"]), P(kx), P(vx), P(xb), Bn, T, st)
            g = gain if (i == 0 and self.n_blocks > 1) else None
            lib.call("lh_proj_ln_res", P(xb), P(bp["proj_w"]), P(bp["proj_b"]), P(bp["proj_slope"]),
                     P(bp["proj_ln_w"]), P(bp["proj_ln_b"]), P(xc), P(g) if g is not None else None, P(xa), Bn, T, st)
        lib.call("lh_deconv_istft", P(xa), P(sin["deconv_buf"]), P(sout["deconv_buf"]), P(sin["istft_buf"]),
                 P(sout["istft_buf"]), P(pk["deconv_w"]), P(pk["deconv_b"]), P(pk["wfb_dec"]), P(y),
                 P(flag) if flag is not None else None, 0, Bn, T, st)



class Streamer:
    """8 ms-chunk streaming driver (reference usage: `Net.predict(chunk[B,2,192], embed[B,256], state, pad=False)`
    in a loop) with the per-chunk launch sequence captured once into HIP graphs.

    One chunk = 128 new samples + 64 look-ahead samples -> 128 output samples.  Everything the chunk loop touches is
    static device memory, and nothing is copied that a kernel can write in place:
      * conv / deconv / iSTFT tails and the inter-LSTM (h, c) exist twice; even chunks read set 0 and write set 1, odd
        chunks the other way round (two captured graphs, replayed alternately);
      * the K / V history of each block is a persistent 50-row ring in the attention kernel's own split-precision
        layout: the new row goes to slot (chunk mod 50) (`lh_qkv_proj_ln` ring_pos, a device counter the graph
        increments), and as the 50 rows are exactly the window of the chunk's one frame, nothing is ever moved;
      * the speaker gain LayerNorm(Linear(embedding)) is computed once in `set_embedding`.
    `step()` is: copy the chunk in, replay a graph, hand back the output buffer (a view that the next `step`
    overwrites).  `use_graph=False` runs the same launches eagerly (also what the capture warm-up does).
    Measured (MI355X, batch 1): 0.68 ms per chunk with the generic `predict` path captured in one graph, see DESIGN.md.
    """

    def __init__(self, net: Net, batch_size: int, device, use_graph: bool = True):
        if net.gemm_mode != "f16x3" or not net.fuse_linear:
            raise ValueError("Streamer runs the split-precision fused kernels (LOOKONCE_GEMM=f16x3, LOOKONCE_FUSE=1)")
        self.net = net
        self.B = B = batch_size
        self.device = dev = torch.device(device)
        n = net.stft_chunk_size + net.stft_pad_size
        z = lambda *s, **k: torch.zeros(*s, device=dev, **k)
        self.chunk = z(B, net.num_ch, n)
        self.embed = z(B, net.spk_emb_dim)
        self.out = z(B, net.n_srcs, net.stft_chunk_size)
        F_, C_, nh, H_ = net.n_freqs, net.emb_dim, net.n_head, net.hidden
        mk = lambda: dict(conv_buf=z(B, net.num_ch * 2, 2, F_), deconv_buf=z(B, C_, 2, F_),
                          istft_buf=z(B, net.n_srcs, F_ * 2, 1),
                          h=[z(1, B * F_, H_) for _ in range(net.n_blocks)], c=[z(1, B * F_, H_) for _ in range(net.n_blocks)])
        self.sets = [mk(), mk()]
        rows = 1 + net.local_atten_len - 1 + KV_PAD_ROWS
        self.rings = [(z(B * nh, rows, 2 * QK_PAD, dtype=torch.float16), z(B * nh, rows, 2 * net.V_dim * F_, dtype=torch.float16))
                      for _ in range(net.n_blocks)]
        self.pos = z(1, dtype=torch.int32)
        self.gain = z(B, F_, C_)
        self.gain_raw = z(B, F_ * C_)
        # range guard: THIS streamer's own two-word flag (never shared with the Net's offline forwards or another streamer);
        # a non-finite chunk (output: zeros) raises at the first `step` after that chunk has finished on the device.
        # On the GPU the flag lives in PINNED HOST memory (device-accessible under unified addressing): the back end stores
        # to it directly in the rare chunk that needs it and `step` just reads the word — no exchange kernel, no copy, no
        # extra launches in the chunk loop (round 3 polled with a kernel + copy every 64th chunk: +0.15 ms on that chunk,
        # which was the p99 of the latency distribution).
        self.range_flag = net._flag_words(dev)
        self.parity = 0
        self.graphs = None
        self.graph = None
        # strong references: the captured graphs bake in pointers into the packed weights and the T=1 workspace.
        # `Net._workspace` evicts its cache after a few shapes and `Net._weights` re-packs after any parameter change;
        # holding the objects here keeps the memory alive, and `step` refuses to replay once the weights were re-packed.
        with torch.no_grad(), net._device_ctx(self.chunk):
            self._pk = net._weights(dev)
            self._pack_key = net._pack_key
            self._n_steps = 0
            # (tensor, version at build time) of every parameter / buffer: `step` re-checks a few per chunk, round robin
            self._versions = [] if net._blob is not None else [(t, t._version) for t in list(net.parameters()) + list(net.buffers())]
            self._vpos = 0
            self._ws = net._workspace(B, 1, dev)
            net._ws.pop((B, 1, str(dev)), None)          # private to this streamer from now on
        if use_graph:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for k in (0, 1):                # warm-up: packs weights, allocates the T=1 workspace
                    self._body(k)
            torch.cuda.current_stream(dev).wait_stream(side)
            self.graphs = []
            for k in (0, 1):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._body(k)
                self.graphs.append(g)
            self.graph = self.graphs[0]
            self.reset()

    def _body(self, k: int):
        with self.net._device_ctx(self.chunk):
            self.net._stream_chunk(self.chunk, self.gain, self.sets[k], self.sets[k ^ 1], self.rings, self.pos, self.out,
                                   self._pk, self._ws, self.range_flag)
            # ring slot counter, kept in [0, window): an ever-growing int32 would go negative after 2^31 chunks and C's
            # `%` would then index before the ring.  One 1-thread kernel (two torch elementwise launches cost 9 us of the chunk)
            st = self.net._stream(self.device)
            self.net._lib(self.pos).call("lh_ring_advance", self.pos.data_ptr(), self.net.local_atten_len, st)

    def reset(self):
        for st in self.sets:
            for t in [st["conv_buf"], st["deconv_buf"], st["istft_buf"]] + st["h"] + st["c"]:
                t.zero_()
        for kx, vx in self.rings:
            kx.zero_()
            vx.zero_()
        self.pos.zero_()
        self.range_flag.zero_()
        self.parity = 0

    def set_embedding(self, embed: torch.Tensor):
        self.embed.copy_(embed.reshape(self.B, -1))
        with torch.no_grad():
            self.net._speaker_gain(self.embed, self.gain_raw, self.gain)

    def step(self, chunk: torch.Tensor) -> torch.Tensor:
        """chunk [B, 2, 192] (128 new + 64 look-ahead samples) -> [B, 2, 128]."""
        # O(1) staleness check (re-deriving the pack key walks all 130 parameters: ~0.1 ms of host time per 8 ms chunk):
        # any `Net` call after a parameter change re-packs and replaces `net._packed`.  The streamer owns references to
        # the images its graphs point into, so a stale streamer is never unsafe, only out of date.
        net = self.net
        cur = net._blob[1] if net._blob is not None else net._packed      # blob-only hosts (`from_packed`) never re-pack
        if cur is not self._pk:
            raise RuntimeError("the Net's parameters changed after this Streamer was built (its HIP graphs hold pointers "
                               "into the old packed weights): create a new streamer with net.make_streamer(...)")
        if net.range_check and int(self.range_flag[0]) != 0:     # host read of the pinned word: free
            self.range_flag.zero_()
            raise RuntimeError("LH_ERR_RANGE: an earlier chunk produced non-finite samples, emitted as zeros (inf / NaN in "
                               "the input or in the carried state); reset() the streamer")
        # ... and a parameter updated IN PLACE (optimizer step, load_state_dict) without any other `Net` call in between
        # would replay silently on the old images: the tensors' version counters are re-checked three per chunk, round
        # robin (all ~130 within 0.4 s of audio).  Round 3 summed all of them every 64th chunk: ~0.13 ms of host time on
        # that chunk — exactly the p99 of the chunk latency (0.40 ms against a p50 of 0.26).
        self._n_steps += 1
        for _ in range(3 if self._versions else 0):
            t, v = self._versions[self._vpos]
            self._vpos = (self._vpos + 1) % len(self._versions)
            if t._version != v:
                raise RuntimeError("a parameter of the Net was modified in place after this Streamer was built: create a new "
                                   "streamer with net.make_streamer(...)")
        self.chunk.copy_(chunk)
        with torch.no_grad():
            if self.graphs is not None:
                self.graphs[self.parity].replay()
            else:
                self._body(self.parity)
        self.parity ^= 1
        return self.out
