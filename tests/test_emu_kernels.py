"""CPU: the UNMODIFIED kernel sources (lookoncetohear_amd/csrc/*.hip) compiled as host C++ against the hipemu
SIMT emulator (tests/hipemu/include/hip/hip_runtime.h: fibers for threads, rendezvous for barriers / shuffles /
MFMA with the documented gfx950 lane maps) and driven through the same C ABI and the same `Net` host code as on
the GPU, checked against the CPU oracle on tiny shapes.  This validates index algebra, weight packing and state
handling without a GPU; it is test tooling, never a fallback (the product loader only opens _lookonce_hip.so)."""
import pytest
import torch

from lookoncetohear_amd import _cabi, synth
from lookoncetohear_amd.net import Net
from tests.hipemu.hosts import EmuEmbed, EmuNet
from oracle import tfgridnet_oracle as O

TOL = 5e-5


@pytest.fixture(scope="module")
def emu_net(oracle_cfg_sd):
    from tests.hipemu.build_emu import build_emu
    lib = _cabi.Lib(build_emu())
    cfg, sd = oracle_cfg_sd
    net = EmuNet(**O.TSH_PARAMS).eval()  # the product host code over the emulated library (tests/hipemu/hosts.py)
    net.load_state_dict(sd, strict=True)
    net.emu_lib = lib
    return net


def test_offline_stages_and_output(emu_net, oracle_cfg_sd):
    cfg, sd = oracle_cfg_sd
    d = synth.batch([0], 128 * 3)
    taps, otaps = {}, {}
    emu_net._debug_taps = taps
    try:
        y = emu_net(d["mixture"], d["embedding_gt"])
    finally:
        emu_net._debug_taps = None
    yo = O.forward(cfg, sd, d["mixture"], d["embedding_gt"], taps=otaps)
    for k in ["Z0", "G"] + [f"blocks.{i}.{n}" for i in range(3) for n in ("Y2", "Q", "K", "V")] + ["blocks.1.out", "blocks.2.out"]:
        assert (taps[k] - otaps[k].reshape(taps[k].shape)).abs().max() < TOL, k
    # block 0 output carries the fused speaker gain (`batch * embed` before block 1, tfgridnet_causal.py:250-251)
    assert (taps["blocks.0.out"] - otaps["blocks.0.out"] * otaps["G"]).abs().max() < TOL
    assert y.shape == yo.shape and (y - yo).abs().max() < TOL


def test_nonzero_state_multi_tile(emu_net, oracle_cfg_sd):
    """B=2, T=19: three front/back-end tiles, two attention tiles, every ring/tail/LSTM state non-zero."""
    cfg, sd = oracle_cfg_sd
    B, T = 2, 19
    d = synth.batch([3, 4], 128 * T + 64)
    st = O.random_state(cfg, B, 3)
    y, s2 = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    yo, so = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    assert (y - yo).abs().max() < TOL
    fo, fm = O.flat_state(so), O.flat_state(s2)
    for k in fo:
        assert fm[k].shape == fo[k].shape and (fm[k] - fo[k]).abs().max() < TOL, k


def test_streaming_chunks_and_mod_pad(emu_net, oracle_cfg_sd):
    cfg, sd = oracle_cfg_sd
    d = synth.batch([5], 128 * 4 + 64)
    st, outs = emu_net.init_buffers(1, "cpu"), []
    for i in range(4):
        y, st = emu_net.predict(d["mixture"][:, :, i * 128:i * 128 + 192], d["embedding_gt"][:, 0], st, pad=False)
        outs.append(y)
    yo, _ = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], None, pad=False)
    assert (torch.cat(outs, -1) - yo).abs().max() < TOL
    d = synth.batch([2], 400)
    mix = d["mixture"][:, :, :300]
    y = emu_net(mix, d["embedding_gt"])
    yo = O.forward(cfg, sd, mix, d["embedding_gt"])
    assert y.shape == yo.shape == (1, 2, 300) and (y - yo).abs().max() < TOL


def test_fused_intra_path_forced_small(emu_net, oracle_cfg_sd):
    """The large-batch intra path (k_intra_xp, lh_recur.hip: LSTM + Linear + residual fused, x half of the gates one step
    ahead, hand-ordered step; forward launch then accumulating reverse launch), which `Net` only selects from 6000 frames (8192 before round 6)
    on, forced at a tiny size: 38 frames = 3 sequence tiles, last one ragged."""
    cfg, sd = oracle_cfg_sd
    B, T = 2, 19
    d = synth.batch([3, 4], 128 * T + 64)
    st = O.random_state(cfg, B, 3)
    yo, so = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    saved = emu_net.fuse_intra_min_frames
    emu_net.fuse_intra_min_frames = 1
    try:
        y, s2 = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    finally:
        emu_net.fuse_intra_min_frames = saved
    assert (y - yo).abs().max() < TOL
    fo, fm = O.flat_state(so), O.flat_state(s2)
    for k in fo:
        assert (fm[k] - fo[k]).abs().max() < TOL, k


def test_previous_fused_intra_kernel_forced_small(emu_net, oracle_cfg_sd):
    """k_ln_lstm_lin<1> (the fused intra kernel before k_intra_xp became the default; kept for A/B runs, selected with
    lh_set_tuning(2, 2)) at the same tiny size: 38 frames = 3 sequence tiles, last one ragged, both directions."""
    cfg, sd = oracle_cfg_sd
    lib = emu_net.emu_lib
    B, T = 2, 19
    d = synth.batch([3, 4], 128 * T + 64)
    st = O.random_state(cfg, B, 3)
    yo, so = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    saved = emu_net.fuse_intra_min_frames
    emu_net.fuse_intra_min_frames = 1
    lib.call("lh_set_tuning", 2, 2)
    try:
        y, s2 = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    finally:
        emu_net.fuse_intra_min_frames = saved
        lib.call("lh_set_tuning", 2, 0)
    assert (y - yo).abs().max() < TOL


@pytest.mark.parametrize("intra", ["fused", "unfused"])
def test_time_windows_equal_the_whole_clip(emu_net, oracle_cfg_sd, intra):
    """`Net.time_chunks` (ABI 14 `_win` entry points): B = 6, T = 7 cut into 3 windows of 2 / 2 / 3 frames — every stage of
    every block through its windowed launch, (h, c) handed from window to window, the attention of a window reading its
    history from the previous windows' rows of the per-block K / V buffers, non-zero state in and the next state out — against
    the whole-clip launches of the same kernels and against the oracle.  (The recurrent stages are bit-identical — the inner
    boundaries carry the cell state in the kernel's internal form; the attention of these 2-frame windows sums its 50 slots in
    another tile alignment than the one 16-frame tile of the whole clip, hence 5e-6 here.  Windows that start on a tile
    boundary reproduce the whole clip bit for bit: tests/test_gpu_modes.py at B = 32.)"""
    cfg, sd = oracle_cfg_sd
    B, T = 6, 7
    d = synth.batch(list(range(20, 20 + B)), 128 * T + 64)
    st = O.random_state(cfg, B, 13)
    yo, so = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    # "fused": the large-batch kernels forced at this size (lh_intra_block_win + lh_inter_block_win); "unfused": the mid-size batch
    # path as it is (6 x 97 sequences: lh_ln_lstm_intra_win + lh_linear_res_win in front of lh_inter_block_win)
    saved = (emu_net.fuse_intra_min_frames, emu_net.stream_intra_max_frames, emu_net.time_chunks, emu_net.time_chunks_small,
             emu_net.chunk_min_frames)
    emu_net.time_chunks = emu_net.time_chunks_small = 1
    if intra == "fused":
        emu_net.fuse_intra_min_frames = 1
    else:
        emu_net.stream_intra_max_frames = 0
    try:
        y1, s1 = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
        emu_net.time_chunks = emu_net.time_chunks_small = 3
        emu_net.chunk_min_frames = 2
        assert emu_net._n_time_chunks(B, T, 1) == 3
        y3, s3 = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
        yz = emu_net(d["mixture"], d["embedding_gt"]) if intra == "fused" else None    # from the zero state: history rows re-zeroed
    finally:
        (emu_net.fuse_intra_min_frames, emu_net.stream_intra_max_frames, emu_net.time_chunks, emu_net.time_chunks_small,
         emu_net.chunk_min_frames) = saved
    assert (y3 - y1).abs().max() < 5e-6 and (y3 - yo).abs().max() < TOL
    f1, f3, fo = O.flat_state(s1), O.flat_state(s3), O.flat_state(so)
    for k in fo:
        assert f3[k].shape == fo[k].shape and (f3[k] - f1[k]).abs().max() < 1e-5 and (f3[k] - fo[k]).abs().max() < TOL, k
    for k in ("h0", "c0", "K_buf", "V_buf"):                  # block 0 up to its Q / K / V stage: recurrences + frame kernels only
        assert torch.equal(f3["gridnet_bufs.buf0." + k], f1["gridnet_bufs.buf0." + k]), k
    if yz is not None:
        assert (yz - O.forward(cfg, sd, d["mixture"], d["embedding_gt"])).abs().max() < TOL


def test_time_windows_of_one_utterance(emu_net, oracle_cfg_sd):
    """`Net.time_chunks_small`: ONE utterance (the latency-bound batch-1 path: unfused intra pair + the per-sequence inter kernel,
    here through `lh_inter_matvec_win`), T = 34 frames cut into windows of 17, non-zero state in, next state out, against the
    whole-clip launches and the oracle.  Block 0's recurrent stages and Q / K / V rows are bit-identical; the attention of the
    second window sums in another tile alignment (see the test above)."""
    cfg, sd = oracle_cfg_sd
    B, T = 1, 34
    d = synth.batch([31], 128 * T + 64)
    st = O.random_state(cfg, B, 17)
    yo, so = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    saved = emu_net.stream_intra_max_frames, emu_net.time_chunks_small, emu_net.chunk_min_frames
    emu_net.stream_intra_max_frames = 0                     # (34 frames would otherwise take the streaming intra kernel)
    try:
        y1, s1 = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
        emu_net.time_chunks_small, emu_net.chunk_min_frames = 2, 2
        assert emu_net._n_time_chunks(B, T, 1) == 2 and emu_net._window_bounds(B, T, 2) == [0, 17, 34]
        y2, s2 = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    finally:
        emu_net.stream_intra_max_frames, emu_net.time_chunks_small, emu_net.chunk_min_frames = saved
    assert (y1 - yo).abs().max() < TOL and (y2 - y1).abs().max() < 5e-6 and (y2 - yo).abs().max() < TOL
    f1, f2, fo = O.flat_state(s1), O.flat_state(s2), O.flat_state(so)
    for k in fo:
        assert (f2[k] - f1[k]).abs().max() < 1e-5 and (f2[k] - fo[k]).abs().max() < TOL, k
    for k in ("h0", "c0", "K_buf", "V_buf"):
        assert torch.equal(f2["gridnet_bufs.buf0." + k], f1["gridnet_bufs.buf0." + k]), k


@pytest.mark.parametrize("tune5", [0, 2, (0, 19)], ids=["k_inter_xp", "k_lstm_lin8p", "k_inter_xp-roles-swapped"])
def test_fused_inter_kernels(emu_net, oracle_cfg_sd, tune5):
    """k_inter_xp (lh_recur.hip, the default) and the previous k_lstm_lin8p (lh_set_tuning(5, 2)): B=6 (582 sequences = 37 tiles, last one ragged; above
    the per-sequence mat-vec kernel's batch limit), T=7, carried (h0, c0) in and (hN, cN) out against the oracle."""
    cfg, sd = oracle_cfg_sd
    lib = emu_net.emu_lib
    B, T = 6, 7
    d = synth.batch(list(range(20, 20 + B)), 128 * T + 64)
    st = O.random_state(cfg, B, 13)
    yo, so = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    tune9 = 3                                              # (default: every wave raised during the on-chain phase)
    if isinstance(tune5, tuple):                           # key 9 + 16: the projection / LayerNorm roles on the other four waves
        tune5, tune9 = tune5
    lib.call("lh_set_tuning", 5, tune5)
    lib.call("lh_set_tuning", 9, tune9)
    try:
        y, s2 = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    finally:
        lib.call("lh_set_tuning", 5, 0)
        lib.call("lh_set_tuning", 9, 3)
    assert (y - yo).abs().max() < TOL
    fo, fm = O.flat_state(so), O.flat_state(s2)
    for k in fo:
        assert fm[k].shape == fo[k].shape and (fm[k] - fo[k]).abs().max() < TOL, k


def test_unfused_inter_path_with_32_sequence_tiles(emu_net, oracle_cfg_sd):
    """The unfused inter pass (lh_ln_lstm_inter + lh_linear_res, LOOKONCE_FUSE=0) in its 32-sequence-tile form
    (k_ln_lstm_h3<2>, only selected with lh_set_tuning(1, 2)): B=2, T=5 = 194 sequences = 6 tiles + a ragged one, carried
    (h0, c0) in and out against the oracle."""
    cfg, sd = oracle_cfg_sd
    lib = emu_net.emu_lib
    B, T = 2, 5
    d = synth.batch([30, 31], 128 * T + 64)
    st = O.random_state(cfg, B, 17)
    yo, so = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    saved = emu_net.fuse_linear
    emu_net.fuse_linear = False
    lib.call("lh_set_tuning", 1, 2)
    try:
        y, s2 = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    finally:
        emu_net.fuse_linear = saved
        lib.call("lh_set_tuning", 1, 0)
    assert (y - yo).abs().max() < TOL
    fo, fm = O.flat_state(so), O.flat_state(s2)
    for k in fo:
        assert (fm[k] - fo[k]).abs().max() < TOL, k


def test_tiled_intra_kernel_forced_small(emu_net, oracle_cfg_sd):
    """The mid-size intra path (lh_ln_lstm_intra: 16-sequence MFMA tiles + lh_linear_res, used between 128 and 6000 (8192 before round 6)
    frames) forced at a size where `Net` would pick the streaming mat-vec kernel."""
    cfg, sd = oracle_cfg_sd
    d = synth.batch([6], 128 * 5 + 64)
    yo = O.forward(cfg, sd, d["mixture"], d["embedding_gt"])
    saved = emu_net.stream_intra_max_frames
    emu_net.stream_intra_max_frames = 0
    try:
        y = emu_net(d["mixture"], d["embedding_gt"])
    finally:
        emu_net.stream_intra_max_frames = saved
    assert (y - yo).abs().max() < TOL


def test_inter_matvec_two_chunks(emu_net, oracle_cfg_sd):
    """Per-sequence inter LSTM (lh_inter_matvec, batch <= 2 and T >= 32): T = 70 = one full 64-step chunk + a ragged one,
    carried (h, c) in and out, against the oracle; the tiled kernel (lh_inter_block) on the same input must agree."""
    cfg, sd = oracle_cfg_sd
    B, T = 1, 70
    d = synth.batch([9], 128 * T + 64)
    st = O.random_state(cfg, B, 11)
    yo, so = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    y, s2 = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    assert (y - yo).abs().max() < TOL
    fo, fm = O.flat_state(so), O.flat_state(s2)
    for k in fo:
        assert (fm[k] - fo[k]).abs().max() < TOL, k
    saved = emu_net.inter_matvec_max_seqs
    emu_net.inter_matvec_max_seqs = 0
    try:
        y2, _ = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    finally:
        emu_net.inter_matvec_max_seqs = saved
    assert (y2 - y).abs().max() < 2e-5


def test_attention_tile_modes_multi_tile(emu_net, oracle_cfg_sd):
    """T = 37 with non-zero state: three 16-frame tiles (one query tile per workgroup) and two 32-frame tiles (two query
    tiles sharing their K / V rows, the default for T > 16), last tile ragged in both; outputs and the next state
    (K_buf / V_buf through lh_ring_unpack) against the oracle."""
    cfg, sd = oracle_cfg_sd
    lib = emu_net.emu_lib
    B, T = 1, 37
    d = synth.batch([7], 128 * T + 64)
    st = O.random_state(cfg, B, 5)
    yo, so = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    fo = O.flat_state(so)
    try:
        for mode in (1, 2, 3):                               # 3 = 40-frame tiles in three MFMA row tiles (one ragged tile here)
            lib.call("lh_set_tuning", 4, mode)
            y, s2 = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
            assert (y - yo).abs().max() < TOL, mode
            fm = O.flat_state(s2)
            for k in fo:
                assert fm[k].shape == fo[k].shape and (fm[k] - fo[k]).abs().max() < TOL, (mode, k)
        assert lib.raw("lh_set_tuning")(4, 4) == 1           # LH_ERR_ARG
        # T = 85: three 40-frame tiles (t0 = 0, 40, 80; the last one 5 frames), dead rows 40..47 of every third row tile
        T2 = 85
        d2 = synth.batch([11], 128 * T2 + 64)
        st2 = O.random_state(cfg, B, 7)
        yo2, _ = O.predict(cfg, sd, d2["mixture"], d2["embedding_gt"][:, 0], O.clone_state(st2), pad=False)
        lib.call("lh_set_tuning", 4, 3)
        y2, _ = emu_net.predict(d2["mixture"], d2["embedding_gt"][:, 0], O.clone_state(st2), pad=False)
        assert (y2 - yo2).abs().max() < TOL
    finally:
        lib.call("lh_set_tuning", 4, 0)


def test_backend_runs_of_tiles(emu_net, oracle_cfg_sd):
    """Back end, T = 37 (three 15-frame tiles) with non-zero state: one run of three consecutive tiles per utterance (the
    second and third tile continue from the first one's partial-product ring and last spectrum), two runs (2 + 1 tiles),
    and the automatic choice (one run per tile here) must all reproduce the oracle's waveform and next state."""
    cfg, sd = oracle_cfg_sd
    lib = emu_net.emu_lib
    B, T = 1, 37
    d = synth.batch([8], 128 * T + 64)
    st = O.random_state(cfg, B, 6)
    yo, so = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
    fo = O.flat_state(so)
    try:
        for runs in (1, 2, 0):
            lib.call("lh_set_tuning", 6, runs)
            y, s2 = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
            assert (y - yo).abs().max() < TOL, runs
            fm = O.flat_state(s2)
            for k in ("deconv_buf", "istft_buf"):
                assert (fm[k] - fo[k]).abs().max() < TOL, (runs, k)
    finally:
        lib.call("lh_set_tuning", 6, 0)


def test_ring_pack_unpack_roundtrip(emu_net):
    """fp32 state -> split-precision history rows -> fp32: hi + 2^-11 lo keeps 22 bits (|err| <= 2^-22 |v| + tiny)."""
    from lookoncetohear_amd.weights import KV_PAD_ROWS, QK_PAD, unsplit_qk, unsplit_v
    lib = emu_net.emu_lib
    B, T = 1, 3
    g = torch.Generator().manual_seed(9)
    kb = torch.randn(4 * B, 49, 582, generator=g) * 3
    vb = torch.randn(4 * B, 49, 1552, generator=g) * 3
    kx = torch.zeros(4 * B, T + 49 + KV_PAD_ROWS, 2 * QK_PAD, dtype=torch.float16)
    vx = torch.zeros(4 * B, T + 49 + KV_PAD_ROWS, 2 * 1552, dtype=torch.float16)
    lib.call("lh_ring_pack", kb.data_ptr(), vb.data_ptr(), kx.data_ptr(), vx.data_ptr(), B, T, None)
    assert (unsplit_qk(kx[:, :49]) - kb).abs().max() < 3 * 2.0 ** -21 and (unsplit_v(vx[:, :49]) - vb).abs().max() < 3 * 2.0 ** -21
    assert kx[:, 49:].abs().max() == 0 and vx[:, 49:].abs().max() == 0            # nothing else touched
    # rows T .. T+48 are what lh_ring_unpack reads: place the packed rows there and read them back
    krows, vrows = kx[:, :49].clone(), vx[:, :49].clone()
    kx[:, T:T + 49], vx[:, T:T + 49] = krows, vrows
    k2, v2 = torch.empty_like(kb), torch.empty_like(vb)
    lib.call("lh_ring_unpack", kx.data_ptr(), vx.data_ptr(), k2.data_ptr(), v2.data_ptr(), B, T, None)
    assert torch.equal(k2, unsplit_qk(krows)) and torch.equal(v2, unsplit_v(vrows))


def test_streamer_ring_and_pingpong(emu_net, oracle_cfg_sd):
    """`Streamer` (eager on the emulator): ping-pong state sets, persistent K / V rings with rotating write slot,
    cached speaker gain.  53 chunks wrap the 50-slot ring; compared with the oracle's chunk-by-chunk `predict`, and a
    second pass after `reset()` must reproduce the first bit for bit."""
    cfg, sd = oracle_cfg_sd
    nchunk = 53
    d = synth.batch([8], 128 * nchunk + 64)
    mix, emb = d["mixture"], d["embedding_gt"][:, 0]
    st = emu_net.make_streamer(1, "cpu", use_graph=False)
    st.set_embedding(emb)
    outs = [st.step(mix[:, :, i * 128:i * 128 + 192]).clone() for i in range(nchunk)]
    ost, oouts = None, []
    for i in range(nchunk):
        yo, ost = O.predict(cfg, sd, mix[:, :, i * 128:i * 128 + 192], emb, ost, pad=False)
        oouts.append(yo)
    assert (torch.cat(outs, -1) - torch.cat(oouts, -1)).abs().max() < TOL
    st.reset()
    outs2 = [st.step(mix[:, :, i * 128:i * 128 + 192]).clone() for i in range(3)]
    assert torch.equal(torch.cat(outs2, -1), torch.cat(outs[:3], -1))


def test_any_finite_input_scale_matches_the_oracle(emu_net, oracle_cfg_sd):
    """Range-safe splits (pow2_scale, lh_common.h): the kernels that split un-normalised data scale
    each row / tile by a power of two first, so a mixture 1e-4 x or 1e6 x the nominal level — far outside what an
    unscaled fp16 hi half can hold — is separated like the plain-fp32 reference does it (tfgridnet_causal.py:188-283):
    finite, and within the tolerance relative to the output amplitude against the fp64 oracle."""
    cfg, sd = oracle_cfg_sd
    d = synth.batch([1], 128 * 5 + 64)
    sd64 = {k: v.double() for k, v in sd.items()}
    for scale in (1e-4, 1e-2, 1.0, 1e2, 1e6):
        mix = d["mixture"] * scale
        y = emu_net(mix, d["embedding_gt"])
        yo = O.forward(cfg, sd64, mix.double(), d["embedding_gt"].double())
        amp = float(yo.abs().max())
        assert torch.isfinite(y).all()
        assert float((y.double() - yo).abs().max()) < TOL * max(amp, 1.0), (scale, amp)


def test_range_guard_is_per_caller_and_emits_silence(emu_net, oracle_cfg_sd):
    """Range guard (include/lookonce_hip.h): a NaN / inf that reaches the back end is stored as 0 and raises the CALLER's
    flag word.  `Net` looks at its own word when its NEXT forward starts (the forward itself stays asynchronous) and in
    `range_status()`; `range_check = "sync"` raises from the offending forward.  A second Net or a Streamer on the same
    device keeps its own word: it neither sees nor clears the first one's flag."""
    cfg, sd = oracle_cfg_sd
    d = synth.batch([1], 128 * 3)
    bad = d["mixture"].clone()
    bad[0, 0, 100] = float("nan")
    other = EmuNet(**O.TSH_PARAMS).eval()
    other.load_state_dict(sd, strict=True)
    other.emu_lib = emu_net.emu_lib
    y = emu_net(bad, d["embedding_gt"])                    # flag raised, not consumed
    assert torch.isnan(y).any()                            # and the NaN reaches the caller, like from the reference
    yo = other(d["mixture"], d["embedding_gt"])            # the other Net must not see (or clear) it
    assert torch.isfinite(yo).all() and other.range_status("cpu") is False
    assert emu_net.range_status("cpu") is True             # still pending for its owner ...
    assert emu_net.range_status("cpu") is False            # ... and cleared by the look
    emu_net(bad, d["embedding_gt"])
    with pytest.raises(RuntimeError, match="LH_ERR_RANGE"):    # deferred: the next forward of the same Net raises
        emu_net(d["mixture"], d["embedding_gt"])
    assert torch.equal(emu_net(d["mixture"], d["embedding_gt"]), yo)
    emu_net.range_check = "sync"
    try:
        with pytest.raises(RuntimeError, match="LH_ERR_RANGE"):
            emu_net(bad, d["embedding_gt"])
    finally:
        emu_net.range_check = True
    assert torch.equal(emu_net(d["mixture"], d["embedding_gt"]), yo)
    st = emu_net.make_streamer(1, "cpu", use_graph=False)
    st.set_embedding(d["embedding_gt"][:, 0])
    st.step(d["mixture"][:, :, :192])
    out = st.step(bad[:, :, :192]).clone()                 # non-finite chunk: the next step sees its flag word
    assert torch.isfinite(out).all()
    assert emu_net.range_status("cpu") is False            # the streamer's flag is not the Net's
    with pytest.raises(RuntimeError, match="LH_ERR_RANGE"):
        st.step(d["mixture"][:, :, :192])
    st.reset()
    assert torch.isfinite(st.step(d["mixture"][:, :, :192])).all()
    assert emu_net.emu_lib.raw("lh_selftest_fp16_subnormal")(None) == 0
    # the C-ABI's own fetch-and-clear (hosts without Python): one exchange, then clean
    flag = torch.ones(2, dtype=torch.int32)
    assert emu_net.emu_lib.raw("lh_range_status")(flag.data_ptr(), None) == 4
    assert emu_net.emu_lib.raw("lh_range_status")(flag.data_ptr(), None) == 0 and int(flag[0]) == 0


def test_all_fp32_mode_reference_kernels(emu_net, oracle_cfg_sd):
    """`gemm_mode = "f32all"`: exact fp32-MFMA recurrences + the plain-fp32 reference kernels of
    lh_ref32.hip for every frame stage (raw state-dict weights, fp32 Q / K / V) — offline from the zero state with every
    stage tap, and one call with a carried non-zero state (history rows, LSTM state, conv / deconv / iSTFT tails in and out)
    against the oracle.  Independent of weights.py's packed frame-kernel images and of the split-precision kernels."""
    cfg, sd = oracle_cfg_sd
    d = synth.batch([0, 1], 128 * 4)
    taps, otaps = {}, {}
    emu_net.gemm_mode, emu_net._debug_taps = "f32all", taps
    try:
        y = emu_net(d["mixture"], d["embedding_gt"])
        emu_net._debug_taps = None
        yo = O.forward(cfg, sd, d["mixture"], d["embedding_gt"], dtype=torch.float64, taps=otaps)
        for k in ["Z0", "G"] + [f"blocks.{i}.{n}" for i in range(3) for n in ("Y2", "Q", "K", "V")] + ["blocks.1.out", "blocks.2.out"]:
            assert (taps[k].double() - otaps[k].reshape(taps[k].shape)).abs().max() < TOL, k
        assert (y.double() - yo).abs().max() < TOL
        B, T = 2, 6
        d = synth.batch([3, 4], 128 * T + 64)
        st = O.random_state(cfg, B, 3)
        yo, so = O.predict(cfg, sd, d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
        y, s2 = emu_net.predict(d["mixture"], d["embedding_gt"][:, 0], O.clone_state(st), pad=False)
        assert (y - yo).abs().max() < TOL
        fo, f2 = O.flat_state(so), O.flat_state(s2)
        for k in fo:
            assert (fo[k] - f2[k]).abs().max() < TOL, k
    finally:
        emu_net.gemm_mode, emu_net._debug_taps = "f16x3", None
    with pytest.raises(ValueError):
        emu_net.gemm_mode = "f32"                          # the old alias of f32rec is gone
        try:
            emu_net(d["mixture"], d["embedding_gt"])
        finally:
            emu_net.gemm_mode = "f16x3"


def test_cabi_argument_errors(emu_net):
    lib = emu_net.emu_lib
    assert lib.raw("lh_check_config")(192, 128, 2, 64, 3, 64, 4, 50, 2, 256) == 0
    assert lib.raw("lh_check_config")(256, 128, 2, 64, 3, 64, 4, 50, 2, 256) == 2      # LH_ERR_UNSUPPORTED
    assert lib.raw("lh_local_attn")(None, None, None, None, 1, 1, None) == 1             # LH_ERR_ARG
    assert lib.raw("lh_linear_res")(None, None, None, None, None, 0, 64, None) == 1


@pytest.mark.parametrize("fused_axis, mv", [(True, True), (True, False), (False, False)],
                         ids=["k_emb_rec+k_emb_inter_mv", "k_emb_rec", "legacy-axis-kernels"])
def test_embedder_stages_and_embedding(fused_axis, mv):
    """Enrollment embedder: every stage tap and the final embedding against oracle/embedder_oracle.py
    (fp64) on 2 utterances x 21 frames — front end, both axis paths, Q/K/V + full attention + projection, head.  Second
    parameter: the round-1 axis kernels (k_emb_gx / k_emb_lstm / k_emb_convt_res behind lh_emb_axis) — only in -DLH_LEGACY
    builds like the emulator's; the product library does not contain them (tests/test_cabi_symbols.py).  First parameter set = the
    product's choice at this size: the inter axis of a small batch on one workgroup per (sequence, direction) (k_emb_inter_mv, round 6;
    18 steps = one ragged chunk, both directions), the intra axis on k_emb_rec; second: k_emb_rec on both axes (larger batches)."""
    from tests.hipemu.build_emu import build_emu
    from lookoncetohear_amd.embed_net import EmbedTFGridNet
    from oracle import embedder_oracle as E
    cfg = E.ECfg(**E.EMBED_PARAMS)
    sd = E.synthetic_state_dict(cfg, 0)
    net = EmuEmbed(**E.EMBED_PARAMS).eval()
    net.load_state_dict(sd, strict=True)
    net.emu_lib = _cabi.Lib(build_emu())
    net.fused_axis = fused_axis
    if not mv:
        net.inter_mv_max_wgs = 0
    x = synth.batch([0, 1], 1280)["mixture"]
    taps, otaps = {}, {}
    net._debug_taps = taps
    emb = net(x)
    net._debug_taps = None
    ref = E.forward(cfg, sd, x, dtype=torch.float64, taps=otaps)
    assert emb.shape == (2, 256)
    for k, v in taps.items():
        o = otaps[k].reshape(v.shape)
        # (2e-5 of the tap's amplitude: the third block's attention output sits at 1.0-1.2e-5 with either recurrent kernel — the
        # logits of random-init weights amplify the fp32 rounding of the rows in front of it)
        assert float((v.double() - o).abs().max()) < 2e-5 * float(o.abs().max()) + 1e-5, k
    assert float((emb.double() - ref).abs().max()) < 2e-5
    assert float(torch.nn.functional.cosine_similarity(emb.double(), ref).min()) > 1 - 1e-9
    with pytest.raises(ValueError):
        net(x[:, :, :100])                 # < 4 STFT frames
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 1280))       # wrong mic count


def test_embedder_attention_gemms_longer_clip():
    """80 frames (Tp = 128: 4 k-steps in the P.V product; the 21-frame test above has 2).
    One utterance,
    embedding against the fp64 oracle, with k_gemm_nt3 (default) and the superseded k_gemm_nt (lh_set_tuning(17, 0))."""
    from tests.hipemu.build_emu import build_emu
    from oracle import embedder_oracle as E
    cfg = E.ECfg(**E.EMBED_PARAMS)
    sd = E.synthetic_state_dict(cfg, 0)
    net = EmuEmbed(**E.EMBED_PARAMS).eval()
    net.load_state_dict(sd, strict=True)
    net.emu_lib = _cabi.Lib(build_emu())
    x = synth.batch([7], 64 * 79)["mixture"]
    ref = E.forward(cfg, sd, x, dtype=torch.float64)
    emb = net(x)
    assert float((emb.double() - ref).abs().max()) < 2e-5
    try:
        for variant in (0, 2):         # 0: the rounds 1-4 GEMM (-DLH_LEGACY builds like the emulator's); 2: k_gemm_nt3 staged two k-steps ahead
            net.emu_lib.call("lh_set_tuning", 17, variant)
            other = net(x)
            assert float((other.double() - ref).abs().max()) < 2e-5 and float((other - emb).abs().max()) < 1e-5, variant
    finally:
        net.emu_lib.call("lh_set_tuning", 17, 3)


def test_packed_blob_is_a_sufficient_weight_source(emu_net, tmp_path):
    """`Net.from_packed`: no parameter tree, every C-ABI weight argument comes from the LHWPACK1 blob
    (include/lookonce_weights.h) — bit-equal to the module-backed run (emulated library, CPU tensors)."""
    from lookoncetohear_amd import checkpoint
    path = str(tmp_path / "sep.lhw")
    checkpoint.export_packed(emu_net, path)
    blob_net = EmuNet.from_packed(path, "cpu")
    assert blob_net.tfgridnet is None and len(list(blob_net.parameters())) == 0
    blob_net.emu_lib = emu_net.emu_lib
    d = synth.batch([2], 128 * 3)
    with torch.no_grad():
        assert torch.equal(blob_net(d["mixture"], d["embedding_gt"]), emu_net(d["mixture"], d["embedding_gt"]))
    # the blob-only host must also stream (real-time use): same chunks through both streamers, bit-equal
    d = synth.batch([4], 128 * 3 + 64)
    outs = []
    for net in (emu_net, blob_net):
        st = net.make_streamer(1, "cpu", use_graph=False)
        st.set_embedding(d["embedding_gt"][:, 0])
        outs.append(torch.cat([st.step(d["mixture"][:, :, i * 128:i * 128 + 192]).clone() for i in range(3)], -1))
    assert torch.equal(outs[0], outs[1])


def test_streamer_refuses_stale_weights(emu_net):
    """A `Streamer` holds pointers into the packed weights it was built with: after a re-pack (any `Net` call following a
    parameter change) the next chunk raises; after an in-place update with no other call in between, the rolling check of
    the tensors' version counters (three per chunk) does."""
    d = synth.batch([4], 128 * 2 + 64)
    st = emu_net.make_streamer(1, "cpu", use_graph=False)
    st.set_embedding(d["embedding_gt"][:, 0])
    st.step(d["mixture"][:, :, :192])
    p = next(emu_net.parameters())
    keep = p.detach().clone()
    try:
        with torch.no_grad():
            p.add_(0.0)                                   # in place: version counter moves, `_packed` does not
        st._vpos = 0                                      # the first parameter is among the next three checked
        with pytest.raises(RuntimeError, match="modified in place"):
            st.step(d["mixture"][:, :, :192])
        emu_net(d["mixture"][:, :, :320], d["embedding_gt"])      # any Net call re-packs
        with pytest.raises(RuntimeError, match="parameters changed"):
            st.step(d["mixture"][:, :, :192])
    finally:
        with torch.no_grad():
            p.copy_(keep)


def test_inter_xp_shortest_sequences(emu_net):
    """k_inter_xp peels its first two steps and has a two-step unrolled loop with a tail: T = 2, 3, 4, 5 (no loop / tail only /
    one loop pass / loop + tail) with carried state and a ragged last tile, against the previous kernel (lh_set_tuning(5, 2));
    T = 1 must route to the previous kernel (the hand-ordered one needs two steps)."""
    lib = emu_net.emu_lib
    bp = emu_net._weights(torch.device("cpu"))["blocks"][1]
    P = lambda t: t.data_ptr()
    B = 1                                            # 97 sequences = 7 tiles, the last one with a single live row
    g = torch.Generator().manual_seed(21)
    for T in (1, 2, 3, 4, 5):
        x = torch.randn(B, T, 97, 64, generator=g)
        h0 = torch.randn(B * 97, 64, generator=g) * 0.3
        c0 = torch.randn(B * 97, 64, generator=g) * 0.3
        res = []
        for tune in (0, 2):
            out, hN, cN = torch.zeros_like(x), torch.zeros_like(h0), torch.zeros_like(c0)
            lib.call("lh_set_tuning", 5, tune)
            try:
                lib.call("lh_inter_block", P(x), P(bp["inter_w8"]), P(bp["inter_b16"]), P(bp["inter_lin_wu"]),
                         P(bp["inter_lin_b"]), P(h0), P(c0), P(hN), P(cN), P(out), B, T, 0)
            finally:
                lib.call("lh_set_tuning", 5, 0)
            res.append((out, hN, cN))
        for a, b in zip(*res):
            assert torch.isfinite(a).all() and (a - b).abs().max() < 2e-6, T
        if T == 1:
            assert all(torch.equal(a, b) for a, b in zip(*res))


def test_ring_advance_wraps(emu_net):
    """lh_ring_advance: the streaming ring slot counter stays in [0, modulo); bad arguments are refused."""
    import ctypes
    lib = emu_net.emu_lib
    pos = torch.tensor([48], dtype=torch.int32)
    for want in (49, 0, 1):
        lib.call("lh_ring_advance", pos.data_ptr(), 50, 0)
        assert pos.item() == want
    assert lib.raw("lh_ring_advance")(ctypes.c_void_p(pos.data_ptr()), 0, None) == 1          # LH_ERR_ARG
    assert lib.raw("lh_ring_advance")(None, 50, None) == 1


def test_enroll_then_separate_chain_on_the_emulator(emu_net, oracle_cfg_sd):
    """The eval loop's hot sequence (reference src/ts_hear_test.py:132-138) on the emulated kernels:
    `eval.evaluate(net, ..., enroll_model=embedder)` = enrollments.squeeze(1) -> embedder -> unsqueeze(1) -> separator ->
    metric rows, against the oracle chain.  Tiny shapes (2 utterances, 3 mixture frames, 21 enrollment frames); the GPU
    version is tests/test_gpu_chain.py."""
    from tests.hipemu.build_emu import build_emu
    from lookoncetohear_amd.embed_net import EmbedTFGridNet
    from lookoncetohear_amd.eval import evaluate
    from lookoncetohear_amd.metrics import per_utterance
    from oracle import embedder_oracle as E
    cfg, sd = oracle_cfg_sd
    ecfg = E.ECfg(**E.EMBED_PARAMS)
    esd = E.synthetic_state_dict(ecfg, 0)
    enet = EmuEmbed(**E.EMBED_PARAMS).eval()
    enet.load_state_dict(esd, strict=True)
    enet.emu_lib = _cabi.Lib(build_emu())
    data_fn = lambda idx: synth.batch(idx, 128 * 3, enroll_n=1280)
    kept = []

    def model(mixture, embedding):
        y = emu_net(mixture, embedding)
        kept.append((y, embedding))
        return y

    res, rows = evaluate(model, data_fn, 2, batch_size=2, device="cpu", enroll_model=enet)
    d = data_fn([0, 1])
    assert d["enrollments"].shape == (2, 1, 2, 1280)
    emb_o = E.forward(ecfg, esd, d["enrollments"].squeeze(1), dtype=torch.float64)
    y_o = O.forward(cfg, {k: v.double() for k, v in sd.items()}, d["mixture"].double(), emb_o.unsqueeze(1))
    y, emb = kept[0]
    assert float((emb[:, 0].double() - emb_o).abs().max()) < TOL
    assert float((y.double() - y_o).abs().max()) < TOL
    o_sisnr, o_snri, o_cos = per_utterance(y_o.float(), d["mixture"], d["target"], emb_o.float(), d["embedding_gt"][:, 0])
    for r in rows:
        assert abs(r["output_sisnr"] - float(o_sisnr[r["idx"]])) < 0.05 and abs(r["si_snr_i"] - float(o_snri[r["idx"]])) < 0.05
        assert abs(r["embedding_sim"] - float(o_cos[r["idx"]])) < 1e-5
    assert res["n"] == 2
