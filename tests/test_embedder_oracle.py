"""CPU: the enrollment-embedder oracle (PARITY UNPINNED — espnet2 trunk restated, see oracle/embedder_oracle.py)
is at least self-consistent: parameter count of SURVEY.md §6 (~2.37 M), output contract of reference
tfgridnet_orig/tfgridnet.py:100-127 ([B, M, N] -> [B, 256]), per-utterance independence, fp32 ~ fp64."""
import torch

from lookoncetohear_amd import synth
from oracle import embedder_oracle as E


def test_embedder_oracle_contract():
    cfg = E.ECfg(**E.EMBED_PARAMS)
    sd = E.synthetic_state_dict(cfg, 0)
    assert sum(v.numel() for v in sd.values()) == 2368681
    d = synth.batch([0, 1], 16000)
    e = E.forward(cfg, sd, d["mixture"])
    assert e.shape == (2, 256) and torch.isfinite(e).all()
    e0 = E.forward(cfg, sd, d["mixture"][:1])
    assert (e0 - e[:1]).abs().max() < 1e-5
    e64 = E.forward(cfg, sd, d["mixture"], dtype=torch.float64)
    assert (e64.float() - e).abs().max() < 1e-4
    # scale invariance from the std normalisation (tfgridnet.py:109-110)
    e2 = E.forward(cfg, sd, 3.0 * d["mixture"])
    assert (e2 - e).abs().max() < 1e-4


def test_embedder_oracle_front_end_and_head_pinned():
    """tests/golden/embedder_pinned_golden.npz was written by oracle/check_embedder_against_reference.py in the build
    container from reference code: `spec_*` = the reference's own `Stft` module (tfgridnet_orig/stft.py:32-233,
    imported unmodified) on the std-normalised seeded inputs; `embed_*` = the reference's `EmbedTFGridNet.forward`
    (tfgridnet_orig/tfgridnet.py:100-127, imported unmodified) executed around a stub trunk whose blocks delegate to
    the oracle's restated block.  Pins the front end (window, centring, reflect pad, scaling, re/im channel order) and
    the head (flatten order, Linear + LayerNorm, frame mean); the inside of the trunk blocks stays a restatement."""
    import os
    import numpy as np
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "embedder_pinned_golden.npz")))
    cfg = E.ECfg(**E.EMBED_PARAMS)
    sd = E.synthetic_state_dict(cfg, 0)
    for tag in ("a", "b"):
        *idx, n = [int(v) for v in g[f"spec_{tag}_idx"]]
        x = synth.batch(idx, n)["mixture"]
        taps = {}
        e = E.forward(cfg, sd, x, dtype=torch.float64, taps=taps)
        assert taps["spec"].shape == g[f"spec_{tag}"].shape
        assert (taps["spec"].float() - torch.from_numpy(g[f"spec_{tag}"])).abs().max() < 1e-5     # fp32 storage of fp64
        assert (e.float() - torch.from_numpy(g[f"embed_{tag}"])).abs().max() < 1e-6
