"""Binaural rendering.  CPU: the oracle against the reference's own call
(`scipy.signal.convolve(src, rir)[:len(src)]`, multi_ch_simulator.py:56-57) and the emulated HIP kernels against the
oracle; GPU: full-size parity through the C ABI."""
import numpy as np
import pytest
import torch

from lookoncetohear_amd import _cabi
from lookoncetohear_amd.render import BinauralRenderer
from oracle import render_oracle as R


def _batch(idxs, n, lh, loud=1.0, reverb=False):
    sc = [R.synthetic_scene(i, n, 3, lh, reverb) for i in idxs]
    sc = [(s[0] * loud, s[1], s[2], s[3]) for s in sc]
    srcs = torch.from_numpy(np.stack([s[0] for s in sc]))
    rirs = torch.from_numpy(np.stack([s[1] for s in sc]))
    gains = torch.ones(len(sc), 4)
    gains[:, 3] = torch.tensor([s[2] for s in sc])
    return sc, srcs, rirs, gains, torch.tensor([s[3] for s in sc])


def _check(rr, dev, idxs, n, lh, loud=1.0, reverb=False):
    sc, srcs, rirs, gains, tgt = _batch(idxs, n, lh, loud, reverb)
    mix, tg, nf, ev = rr.render(srcs.to(dev), rirs.to(dev), gains.to(dev), tgt.to(dev))
    mix, tg, nf, ev = mix.cpu(), tg.cpu(), nf.cpu(), ev.cpu()
    for b, s in enumerate(sc):
        m64, t64, n64, e64 = R.render(*s, exact=True)
        amp = float(e64.abs().max())
        tol = 4e-6 * amp * max(1.0, (lh / 256) ** 0.5)          # fp32 accumulation over Lh taps
        assert float((ev[b] - e64).abs().max()) < tol, (b, n, lh)
        assert float((mix[b] - m64).abs().max()) < 2 * tol and float((tg[b] - t64).abs().max()) < tol
        assert abs(float(nf[b]) - float(n64)) < 2 * tol
        # the mixing stage is bit-exact in the reference's operation order, given the rendered rows
        mm, tt, nn = R.mix([ev[b, i] for i in range(3)], ev[b, 3], 1.0, s[3])
        assert torch.equal(mm, mix[b]) and torch.equal(tt, tg[b]) and float(nn) == float(nf[b])
    return nf


def test_oracle_matches_reference_call():
    """oracle == scipy.signal.convolve as the reference calls it; exact (float64) sum within fp32 rounding of it."""
    from scipy.signal import convolve
    srcs, rirs, ns, ti = R.synthetic_scene(5, 6000, 3, 300)
    for i in range(4):
        ref = np.stack([convolve(srcs[i], rirs[i, 0])[:6000], convolve(srcs[i], rirs[i, 1])[:6000]])
        assert ref.dtype == np.float32
        assert np.array_equal(R.convolve_trunc(srcs[i], rirs[i]), ref)
        assert np.abs(R.convolve_trunc(srcs[i], rirs[i], exact=True) - ref).max() < 2e-6 * np.abs(ref).max() + 1e-7
    m, t, nf, ev = R.render(srcs * 5, rirs, ns, ti)
    assert float(nf) > 1.0 and abs(float(m.abs().max()) - 1.0) < 1e-6 and m.shape == t.shape == (2, 6000)


def test_emulated_kernels_match_oracle():
    from tests.hipemu.build_emu import build_emu
    from tests.hipemu.hosts import EmuRenderer
    rr = EmuRenderer()
    rr.emu_lib = _cabi.Lib(build_emu())
    nf = _check(rr, "cpu", [0, 1], 5003, 200)                   # ragged N, peak below 1: no normalisation
    assert float(nf.max()) < 1.0
    nf = _check(rr, "cpu", [2], 4500, 3000, loud=4.0, reverb=True)   # room-length response: overlap-save FFT path, peak > 1
    assert float(nf.min()) > 1.0
    _check(rr, "cpu", [4], 9000, 1024)                          # shortest FFT-path response, three output blocks
    _check(rr, "cpu", [5], 2100, 1000)                          # one tap short of it: direct form, several tap stages
    _check(rr, "cpu", [3], 300, 7)                              # shorter than a tile, tiny filter
    with pytest.raises(ValueError):
        rr.render(torch.zeros(1, 4, 100), torch.zeros(1, 3, 2, 8), torch.ones(1, 4), torch.zeros(1, dtype=torch.int64))
    with pytest.raises(IndexError):
        rr.render(torch.zeros(1, 4, 100), torch.zeros(1, 4, 2, 8), torch.ones(1, 4), torch.tensor([4]))


@pytest.mark.gpu
def test_gpu_render_full_size():
    _cabi.load()
    rr = BinauralRenderer()
    _check(rr, "cuda:0", [0, 1, 2, 3], 80000, 256)              # 5 s clips, HRIR-length responses
    _check(rr, "cuda:0", [4, 5], 80000, 4096, loud=3.0, reverb=True)   # BRIR-length responses (FFT path), normalised
    _check(rr, "cuda:0", [9], 80000, 4097, reverb=True)         # longest FFT-path response
    _check(rr, "cuda:0", [10], 50000, 5000, reverb=True)        # longer than that: direct form with tap stages
    _check(rr, "cuda:0", [11], 80000, 1023)                     # just below the FFT path
    _check(rr, "cuda:0", [6], 12345, 33)
    sc, srcs, rirs, gains, tgt = _batch([7, 8], 80000, 256)
    a = rr.render(srcs.cuda(), rirs.cuda(), gains.cuda(), tgt.cuda())
    b = rr.render(srcs.cuda(), rirs.cuda(), gains.cuda(), tgt.cuda())
    assert all(torch.equal(x, y) for x, y in zip(a, b))         # deterministic
    ev = rr.convolve(srcs[0].cuda(), rirs[0].cuda())
    assert torch.equal(ev[:3], a[3][0, :3])
    with pytest.raises(RuntimeError):
        rr.render(srcs, rirs, gains, tgt)                       # CPU tensors: no fallback


@pytest.mark.gpu
def test_render_next_to_lstm_kernels_is_bit_identical():
    """lh_render.hip is the one file still built with packed fp32 (v_pk_fma_f32: the FIR / FFT kernels run at half the rate
    without it).  Packed chains of OTHER kernels lost accumulate steps in lanes 48..63 whenever a workgroup of the LSTM
    kernels (second HIP stream) shared their CU — profiles/r03c_packed_fp32_corruption.txt; the render kernels came through
    that stress bit-exact, and this test keeps it that way: renders on one stream while the fused intra LSTM kernel runs
    back to back on another must equal the quiet render, both response lengths (direct form and FFT path)."""
    from lookoncetohear_amd import config
    from lookoncetohear_amd.net import Net
    lib = _cabi.load()
    dev = torch.device("cuda:0")
    rr = BinauralRenderer()
    torch.manual_seed(0)
    net = Net(**config.TSH_PARAMS).eval().to(dev)
    bp = net._weights(dev)["blocks"][0]
    nx = torch.randn(32, 625, 97, 64, device=dev)
    nout = torch.empty_like(nx)
    P = lambda t: t.data_ptr()
    s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for taps, seeds in ((256, [0, 1, 2, 3]), (4096, [4, 5, 6, 7])):
        sc, srcs, rirs, gains, tgt = _batch(seeds, 80000, taps)
        args = (srcs.to(dev), rirs.to(dev), gains.to(dev), tgt.to(dev))
        quiet = rr.render(*args)
        torch.cuda.synchronize()
        for rep in range(8):
            with torch.cuda.stream(s0):
                for _ in range(3):
                    lib.call("lh_intra_block", P(nx), P(bp["intra_w16"]), P(bp["intra_b16"]), P(bp["intra_lin_w2"]),
                             P(bp["intra_lin_b"]), P(nout), 32 * 625, s0.cuda_stream)
            with torch.cuda.stream(s1):
                noisy = rr.render(*args)
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(quiet, noisy)), (taps, rep)
