"""Deterministic synthetic binaural mixtures with the reference dataset's I/O contract.

The reference eval loop (reference `src/ts_hear_test.py:124-146`) consumes, per utterance,
`mixture [2, 80000]` (16 kHz, 5 s, peak <= 1), `target [2, 80000]` and a unit-L2 non-negative
d-vector `embedding_gt [1, 256]`, produced by `MixLibriSpeechNoisyEnroll.__getitem__`
(reference `src/datasets/MixLibriSpeechNoisyEnrollNorm.py:152-376`).  Neither the dataset nor a
checkpoint exists in this environment, so `bench.py`, the eval driver and the tests use this
generator instead: per-utterance seed = utterance index (the reference seeds
val/test samples by index too, `MixLibriSpeechNoisyEnrollNorm.py:164-168`), three pseudo-speech
sources (AM harmonic stacks, f0 100-250 Hz, random on/off bursts) each rendered through a random
2-channel 64-tap FIR (pseudo-HRIR, interaural delay <= 16 samples, cf. `max_shift: 16`
`configs/tsh.json:87`), plus 2-channel coloured noise, peak-normalised when above 1.
"""
from __future__ import annotations

import numpy as np
import torch

SR = 16000


def _source(rs: np.random.RandomState, n: int) -> np.ndarray:
    t = np.arange(n) / SR
    f0 = rs.uniform(100.0, 250.0) * (1.0 + 0.03 * np.sin(2 * np.pi * rs.uniform(2, 6) * t + rs.uniform(0, 6.28)))
    phase = 2 * np.pi * np.cumsum(f0) / SR
    sig = np.zeros(n)
    for h in range(1, 13):
        sig += rs.uniform(0.2, 1.0) / h * np.sin(h * phase + rs.uniform(0, 6.28))
    # syllable-rate envelope with random pauses
    env = 0.5 * (1 + np.sin(2 * np.pi * rs.uniform(3, 5) * t + rs.uniform(0, 6.28)))
    gate = (rs.rand(int(np.ceil(n / 4000)) + 1) > 0.25).astype(np.float64)
    gate = np.repeat(gate, 4000)[:n]
    gate = np.convolve(gate, np.ones(400) / 400, mode="same")[:n]
    return sig * env * gate


def _hrir(rs: np.random.RandomState) -> np.ndarray:
    h = np.zeros((2, 64))
    itd = rs.randint(0, 17)
    near = rs.randint(0, 2)
    decay = np.exp(-np.arange(64) / rs.uniform(3, 9))
    for ch in range(2):
        d = itd if ch != near else 0
        taps = rs.randn(64 - d) * decay[: 64 - d]
        taps[0] += 1.0
        h[ch, d:] = taps * (rs.uniform(0.4, 0.9) if ch != near else 1.0)
    return h


def utterance(idx: int, n: int = 80000):
    """Returns (mixture [2,n], target [2,n], embedding [1,256]) float32 numpy arrays for utterance `idx`."""
    rs = np.random.RandomState(idx)
    srcs = []
    for _ in range(3):
        s = _source(rs, n)
        h = _hrir(rs)
        srcs.append(np.stack([np.convolve(s, h[c])[:n] for c in range(2)]))
    srcs = np.stack(srcs)                                   # [3,2,n]
    srcs *= 0.12 / (srcs.std() + 1e-9)
    noise = rs.randn(2, n + 8)
    noise = np.stack([np.convolve(noise[c], [0.4, 0.3, 0.15, 0.08, 0.04, 0.02, 0.01, 0.005], mode="valid")[:n]
                      for c in range(2)])
    noise *= 0.004 * rs.uniform(3.0, 10.0)                  # noise_scale U(3,10), configs/tsh.json:86
    mix = srcs.sum(0) + noise
    peak = np.abs(mix).max()
    if peak > 1.0:                                          # peak-normalise (MixLibriSpeechNoisyEnrollNorm.py:196-202)
        mix, srcs = mix / peak, srcs / peak
    tgt = srcs[rs.randint(0, 3)]
    emb = np.abs(rs.randn(256))
    emb = emb / np.linalg.norm(emb)
    return mix.astype(np.float32), tgt.astype(np.float32), emb[None].astype(np.float32)


ENROLL_SEED_OFFSET = 1_000_003          # the enrollment recording of utterance i is synthetic utterance i + this


def enrollment(idx: int, n: int = 80000) -> np.ndarray:
    """[1, 2, n] float32: the noisy binaural enrollment recording the reference dataset returns as `inputs['enrollments']`
    (`num_enroll` = 1 recordings of `enroll_len` = 5 s: MixLibriSpeechNoisyEnrollNorm.py:118, 259-303 — another scene
    with the target speaker in it, rendered like a mixture).  Synthetic stand-in: the mixture of an unrelated synthetic
    utterance, seeded by the utterance index like everything else here."""
    return utterance(ENROLL_SEED_OFFSET + int(idx), n)[0][None]


def batch(indices, n: int = 80000, enroll_n: int = 0):
    """Stack utterances -> dict of torch CPU tensors: mixture [B,2,n], target [B,2,n], embedding_gt [B,1,256] and, with
    `enroll_n` > 0, enrollments [B,1,2,enroll_n] (the eval loop squeezes dim 1, reference src/ts_hear_test.py:133)."""
    m, t, e = zip(*(utterance(int(i), n) for i in indices))
    out = dict(mixture=torch.from_numpy(np.stack(m)), target=torch.from_numpy(np.stack(t)),
               embedding_gt=torch.from_numpy(np.stack(e)))
    if enroll_n:
        out["enrollments"] = torch.from_numpy(np.stack([enrollment(int(i), enroll_n) for i in indices]))
    return out
