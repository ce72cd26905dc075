"""CPU: host-side glue of the drop-in that needs no kernel — `mod_pad` against the reference's two-pad form
(reference src/models/tfgridnet_realtime/net.py:8-18), and the profile tooling that `scripts/gpu.sh` runs on the GPU box
(`scripts/rocpd_summary.py --timeline` on a synthetic rocpd database)."""
import os
import sqlite3
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference_mod_pad(x, chunk_size, pad):
    # the reference's arithmetic, restated: right-pad to a multiple of chunk_size, then F.pad(x, pad)
    mod = 0
    if x.shape[-1] % chunk_size != 0:
        mod = chunk_size - (x.shape[-1] % chunk_size)
    return F.pad(F.pad(x, (0, mod)), pad), mod


@pytest.mark.parametrize("n", [1, 127, 128, 129, 2049, 80000])
@pytest.mark.parametrize("pad", [(0, 0), (0, 64), (3, 64)])
def test_mod_pad_is_the_reference_two_pad_form_in_one_copy(n, pad):
    from lookoncetohear_amd.net import mod_pad
    x = torch.randn(2, 2, n, generator=torch.Generator().manual_seed(n))
    got, mod = mod_pad(x, 128, pad)
    ref, mod_ref = _reference_mod_pad(x, 128, pad)
    assert mod == mod_ref and got.shape == ref.shape and torch.equal(got, ref)
    if mod == 0 and pad == (0, 0):
        assert got.data_ptr() == x.data_ptr()          # nothing to pad: no copy (the kernels only read x)


def test_timeline_of_one_step_from_a_rocpd_database(tmp_path):
    db = str(tmp_path / "r.db")
    c = sqlite3.connect(db)
    c.execute("create table kernels(name text, start int, end int)")
    t = 0
    for _ in range(3):
        for name, dur, gap in [("lh::k_stft_conv_in", 140_000, 0), ("at::pad", 8_000, 1_500), ("lh::k_intra_xp", 400_000, 6_000)]:
            t += gap
            c.execute("insert into kernels values(?,?,?)", (name, t, t + dur))
            t += dur
    c.commit()
    c.close()
    out = str(tmp_path / "tl.txt")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rocpd_summary.py"), db, out, "--timeline"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = open(out).read().splitlines()
    rows = [l for l in lines if not l.startswith("#")]
    assert len(rows) == 3 and rows[0].startswith("lh::k_stft_conv_in") and rows[2].split()[-1] == "6.0"
    assert "kernels 548.0 + gaps 7.5" in lines[-1] and "shorter than 100 us: 8.0 us" in lines[-1]


def test_stale_gemm_mode_spelling_fails_at_construction_not_mid_forward(monkeypatch):
    """`LOOKONCE_GEMM=f32` (the spelling removed in round 5) used to raise only after the front-end kernels of a
    forward had been launched; it now fails when the Net is built, and an attribute set later is checked before the first launch."""
    from lookoncetohear_amd import config
    from lookoncetohear_amd.net import Net
    monkeypatch.setenv("LOOKONCE_GEMM", "f32")
    with pytest.raises(ValueError, match="f32rec"):
        Net(**config.TSH_PARAMS)
    monkeypatch.setenv("LOOKONCE_GEMM", "f32rec")
    net = Net(**config.TSH_PARAMS).eval()
    assert net.gemm_mode == "f32rec"
    net.gemm_mode = "f32"
    with pytest.raises(ValueError, match="f32rec"), torch.no_grad():
        net(torch.zeros(1, 2, 4000), torch.zeros(1, 1, 256))      # refused before any library / device is touched
