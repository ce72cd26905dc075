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
