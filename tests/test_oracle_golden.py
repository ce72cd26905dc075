"""CPU: the oracle restatement reproduces the committed reference outputs (tests/golden, written by
oracle/gen_golden.py from the unmodified reference model).  Tolerances: vs the fp64 reference output the
fp32 oracle must sit at the fp32 noise floor (reference fp32-vs-fp64 itself is ~1e-5)."""
import numpy as np
import torch

from oracle import tfgridnet_oracle as O
from lookoncetohear_amd import synth

TOL32 = 5e-5      # fp32 oracle vs fp64 reference truth
TOL64 = 2e-6      # fp64 oracle vs fp64 reference truth stored as fp32 (rounding of the stored value)


def _maxabs(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def test_offline_cases(golden, oracle_cfg_sd):
    cfg, sd = oracle_cfg_sd
    for name, idx, n in (("off_b2_n8000", [0, 1], 8000), ("off_b1_n8100", [2], 8100)):
        d = synth.batch(idx, n)
        y32 = O.forward(cfg, sd, d["mixture"], d["embedding_gt"])
        y64 = O.forward(cfg, sd, d["mixture"], d["embedding_gt"], dtype=torch.float64)
        assert y32.shape == golden[name + "_y64"].shape
        assert _maxabs(y32, golden[name + "_y64"]) < TOL32
        assert _maxabs(y32, golden[name + "_y32"]) < TOL32
        assert _maxabs(y64, golden[name + "_y64"]) < TOL64


def test_nonzero_state_in_out(golden, oracle_cfg_sd):
    cfg, sd = oracle_cfg_sd
    d = synth.batch([3, 4], 128 * 12 + 64)
    st = O.random_state(cfg, 2, 3)
    y, st2 = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], st, pad=False)
    assert _maxabs(y, golden["state_b2_y64"]) < TOL32
    for k, v in O.flat_state(st2).items():
        assert _maxabs(O.subsample(v, 256), golden["state_b2_s64." + k]) < TOL32, k


def test_streaming_matches_reference_and_offline(golden, oracle_cfg_sd):
    cfg, sd = oracle_cfg_sd
    nchunk = 60
    d = synth.batch([5], 128 * nchunk + 64)
    st, outs = None, []
    for i in range(nchunk):
        y, st = O.predict(cfg, sd, d["mixture"][:, :, i * 128:i * 128 + 192], d["embedding_gt"][:, 0], st, pad=False)
        outs.append(y)
    ys = torch.cat(outs, -1)
    assert _maxabs(ys, golden["stream_b1_y64"]) < TOL32
    y_off, _ = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], None, pad=False)
    assert _maxabs(ys, y_off) < TOL32          # streaming == offline invariant


def test_full_clip_and_intermediates(golden, oracle_cfg_sd):
    cfg, sd = oracle_cfg_sd
    d = synth.batch([6], 80000)
    taps = {}
    y = O.forward(cfg, sd, d["mixture"], d["embedding_gt"], fast_lstm=True, taps=taps)
    assert tuple(y.shape) == (1, 2, 80000)
    assert _maxabs(y[:, :, ::8], golden["full_b1_y64"]) < TOL32
    for k in ("Z0", "G", "blocks.0.Q", "blocks.1.K", "blocks.2.V", "blocks.0.out", "blocks.1.out", "blocks.2.out"):
        assert _maxabs(O.subsample(taps[k]), golden["full_b1_t64." + k]) < TOL32, k


def test_si_snr_definition():
    g = torch.Generator().manual_seed(0)
    t = torch.randn(3, 2, 4000, generator=g)
    p = 0.7 * t + 0.1 * torch.randn(3, 2, 4000, generator=g) + 0.3
    # scale / offset invariance and the closed form 10log10(|a t|^2/|a t - p|^2)
    v = O.si_snr(p, t)
    assert torch.allclose(v, O.si_snr(3.0 * p + 1.0, t), atol=1e-3)
    assert v.shape == (3, 2) and (v > 10).all() and (v < 30).all()
    assert O.si_snr_i(p, p, t).abs().max() < 1e-5


def test_package_random_init_weights_equal_the_oracles(oracle_cfg_sd):
    """bench.py / smoke() draw their random-init weights from lookoncetohear_amd.config (the product side must not
    import oracle/); the oracle keeps its own copy of the rules.  Both must be the same tensors, bit for bit — the
    committed goldens were generated from the oracle's."""
    from lookoncetohear_amd import config
    from oracle import embedder_oracle as E
    _, sd = oracle_cfg_sd
    assert config.TSH_PARAMS == O.TSH_PARAMS and config.EMBED_PARAMS == E.EMBED_PARAMS
    mine = config.separator_weights(0)
    assert mine.keys() == sd.keys()
    for k in sd:
        assert torch.equal(mine[k], sd[k]), k
    esd = E.synthetic_state_dict(E.ECfg(**E.EMBED_PARAMS), 3)
    emine = config.embedder_weights(3)
    assert emine.keys() == esd.keys()
    for k in esd:
        assert torch.equal(emine[k], esd[k]), k


def test_foreign_filterbank_keys_are_dropped_and_training_forward_is_refused(oracle_cfg_sd):
    """A reference checkpoint may carry asteroid's second filterbank buffer (`torch_window`); it must not break the
    strict load.  And the drop-in refuses a grad-enabled training-mode forward instead of returning a graph-less
    tensor that `loss.backward()` would choke on."""
    import pytest
    from lookoncetohear_amd.net import Net
    _, sd = oracle_cfg_sd
    sd2 = dict(sd)
    sd2["tfgridnet.enc.filterbank.torch_window"] = torch.zeros(192)
    sd2["tfgridnet.dec.filterbank.torch_window"] = torch.zeros(192)
    net = Net(**O.TSH_PARAMS)
    net.load_state_dict(sd2, strict=True)
    with pytest.raises(RuntimeError, match="inference-only"):
        net(torch.zeros(1, 2, 1000), torch.zeros(1, 1, 256))


def test_aten_port_reproduces_reference_fp32_goldens(golden, oracle_cfg_sd):
    """oracle/aten_port.py issues the reference's own ATen operator sequence (it is what bench.py's cpu_baseline times):
    against the reference's committed fp32 outputs it agrees to the last bits (bit-identical on the machine that wrote
    the goldens: `python -m oracle.aten_port`; other CPUs may pick other MKL / oneDNN kernels, hence 1e-6)."""
    from oracle import aten_port as P
    _, sd = oracle_cfg_sd
    d = P.Dims()
    for name, idx, n in (("off_b2_n8000", [0, 1], 8000), ("off_b1_n8100", [2], 8100)):
        b = synth.batch(idx, n)
        y = P.forward(d, sd, b["mixture"], b["embedding_gt"])
        assert y.shape == golden[name + "_y32"].shape
        assert _maxabs(y, golden[name + "_y32"]) < 1e-6 * max(1.0, float(np.abs(golden[name + "_y32"]).max()))
        assert _maxabs(y, golden[name + "_y64"]) < TOL32
