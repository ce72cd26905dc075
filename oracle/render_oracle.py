"""CPU oracle of the binaural rendering step — TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline).

Restates, with scipy/numpy exactly as the reference calls them:
  * `SOFASimulator._convolve` / `ASHSimulator._convolve` (reference src/datasets/multi_ch_simulator.py:40-61, :166-174):
        src_l = convolve(src, rir[0])[:len(src)];  src_r = convolve(src, rir[1])[:len(src)]
    (`scipy.signal.convolve`, float32 in -> float32 out; scipy picks direct or FFT by size, so the reference's own
    result carries ~1e-6 relative rounding; `render(..., exact=True)` evaluates the same sum in float64 instead)
  * the mixing of `MixLibriSpeechNoisyEnrollNorm.__getitem__` (reference
    src/datasets/MixLibriSpeechNoisyEnrollNorm.py:176-202): noise * noise_scale, norm_factor = |sum(events) + noise|.max(),
    divide everything by it when > 1, mixture = sum(events) + noise, target = events[tgt_idx]
    (the train-only white/pink/brown augmentation of :186-195 draws from unseeded RNGs and is not part of the path).
The choice of impulse responses (SOFA / BRIR files picked by `random.Random(seed)`) is data selection, not
arithmetic: callers pass the chosen responses in.  Pinned against the reference's own call: scipy is importable here,
so tests compare this file with `scipy.signal.convolve` directly (tests/test_render.py).
"""
from __future__ import annotations

import numpy as np
import torch


def convolve_trunc(src: np.ndarray, rir2: np.ndarray, exact: bool = False) -> np.ndarray:
    """[N], [2, Lh] -> [2, N]: multi_ch_simulator.py:56-58."""
    if exact:
        out = [np.convolve(src.astype(np.float64), rir2[e].astype(np.float64))[:len(src)] for e in range(2)]
        return np.stack(out, 0)
    from scipy.signal import convolve
    return np.stack([convolve(src, rir2[0])[:len(src)], convolve(src, rir2[1])[:len(src)]], axis=0)


def mix(events, noise, noise_scale: float, tgt_idx: int):
    """events: list of float32 tensors [2, N]; noise [2, N].  MixLibriSpeechNoisyEnrollNorm.py:176-202 (eval path)."""
    events = [e.clone().float() for e in events]
    noise = noise.clone().float() * noise_scale
    norm_factor = torch.abs(sum(events) + noise).max()
    if norm_factor > 1.0:
        for i in range(len(events)):
            events[i] /= norm_factor
        noise /= norm_factor
    mixture = sum(events) + noise
    return mixture, events[tgt_idx], norm_factor


def render(srcs: np.ndarray, rirs: np.ndarray, noise_scale: float, tgt_idx: int, exact: bool = False):
    """srcs [S1, N] (noise LAST), rirs [S1, 2, Lh] -> mixture [2, N], target [2, N], norm_factor, events [S1, 2, N]
    (events before normalisation, noise row already scaled)."""
    ev = [convolve_trunc(srcs[i], rirs[i], exact) for i in range(srcs.shape[0])]
    evt = [torch.from_numpy(np.ascontiguousarray(e)).float() for e in ev]
    mixture, target, nf = mix(evt[:-1], evt[-1], noise_scale, tgt_idx)
    events = torch.stack(evt[:-1] + [evt[-1] * noise_scale])
    return mixture, target, nf, events


def synthetic_scene(idx: int, n: int = 80000, n_src: int = 3, lh: int = 256, reverb: bool = False):
    """Deterministic stand-in for (scaper sources, HRIR/BRIR pick): S1 = n_src + 1 mono rows and 2-ear responses with
    an inter-aural delay <= 16 samples and level difference; `reverb` adds an exponentially decaying tail (BRIR-like)."""
    g = np.random.default_rng(1000 + idx)
    t = np.arange(n) / 16000.0
    rows = []
    for s in range(n_src):
        f0 = g.uniform(100, 250)
        env = (np.sin(2 * np.pi * g.uniform(1.5, 4.0) * t + g.uniform(0, 6.28)) > -0.2).astype(np.float64)
        sig = sum(np.sin(2 * np.pi * f0 * (k + 1) * t + g.uniform(0, 6.28)) / (k + 1) for k in range(8)) * env
        rows.append(0.2 * sig / (np.abs(sig).max() + 1e-9))
    rows.append(0.02 * g.standard_normal(n))
    srcs = np.stack(rows).astype(np.float32)
    rirs = np.zeros((n_src + 1, 2, lh), np.float64)
    for s in range(n_src + 1):
        itd = int(g.integers(0, 17))
        for e in range(2):
            d = min(8 + (itd if e == 1 else 0), lh - 1)
            taps = g.standard_normal(lh) * np.exp(-np.arange(lh) / (lh / 6.0 if reverb else 12.0))
            taps[:d] = 0.0
            taps[d] = 1.0 * (0.6 if e == 1 else 1.0)
            rirs[s, e] = taps * (0.3 if not reverb else 0.2)
            rirs[s, e, d] = (0.6 if e == 1 else 1.0)
    rs = np.random.RandomState(idx)
    return srcs, rirs.astype(np.float32), float(rs.uniform(3.0, 10.0)), int(rs.randint(n_src))
