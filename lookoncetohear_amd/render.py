"""Binaural rendering on the MI355X — the data-pipeline step in front of the separator.

Host-side mirror of the arithmetic in the reference's simulators and dataset:
`SOFASimulator._convolve(src, hrtf, idx)` / `ASHSimulator._convolve(src, hrtf_file)` (reference
src/datasets/multi_ch_simulator.py:40-61, :166-174) and `MixLibriSpeechNoisyEnrollNorm.__getitem__` lines 176-202,
batched over utterances.  Choosing WHICH impulse responses (SOFA / BRIR files by `random.Random(seed)`) stays with
the caller — that is data selection, and the `sofa` / `scaper` packages it needs are not part of this path.
No CPU fallback: CUDA tensors only.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _cabi


class BinauralRenderer(_cabi.HipHost):
    _host_name = "BinauralRenderer"

    def convolve(self, src: torch.Tensor, rir: torch.Tensor) -> torch.Tensor:
        """`_convolve`, batched: src [R, N] mono rows, rir [R, 2, Lh] -> [R, 2, N] (`convolve(...)[:len(src)]` per ear)."""
        R = src.shape[0]
        ones = torch.ones(1, R, device=src.device)
        tgt = torch.zeros(1, dtype=torch.int32, device=src.device)
        ev = self.render(src[None], rir[None], ones, tgt)[3]
        return ev[0]

    def render(self, srcs: torch.Tensor, rirs: torch.Tensor, gains: torch.Tensor, tgt_idx: torch.Tensor
               ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """srcs [B, S1, N] (noise bed LAST), rirs [B, S1, 2, Lh], gains [B, S1] (1 for sources, noise_scale for the
        noise row), tgt_idx [B] -> mixture [B, 2, N], target [B, 2, N], norm_factor [B], events [B, S1, 2, N]."""
        lib = self._lib(srcs)
        if srcs.dim() != 3 or rirs.dim() != 4 or rirs.shape[:2] != srcs.shape[:2] or rirs.shape[2] != 2:
            raise ValueError(f"expected srcs [B,S1,N] and rirs [B,S1,2,Lh]; got {tuple(srcs.shape)} and {tuple(rirs.shape)}")
        B, S1, N = srcs.shape
        Lh = rirs.shape[3]
        if tuple(gains.shape) != (B, S1) or tuple(tgt_idx.shape) != (B,):
            raise ValueError("gains must be [B,S1] and tgt_idx [B]")
        if int(tgt_idx.max()) >= S1 or int(tgt_idx.min()) < 0:
            raise IndexError("tgt_idx out of range")
        dev = srcs.device
        srcs, rirs, gains = srcs.contiguous().float(), rirs.contiguous().float(), gains.contiguous().float()
        tgt_idx = tgt_idx.to(torch.int32).contiguous()
        events = torch.empty(B, S1, 2, N, device=dev)
        mixture, target = torch.empty(B, 2, N, device=dev), torch.empty(B, 2, N, device=dev)
        peak = torch.empty(B, dtype=torch.int32, device=dev)
        st = self._stream(dev)
        P = lambda t: t.data_ptr()
        with self._device_ctx(srcs):
            lib.call("lh_render_binaural", P(srcs), P(rirs), P(gains), P(tgt_idx), P(events), P(peak), P(mixture),
                     P(target), B, S1, N, Lh, st)
        return mixture, target, peak.view(torch.float32), events
