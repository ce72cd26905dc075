"""Drives scripts/ubench/libpk_race*.so (the packed-fp32 variance chain of k_proj_ln_res, verbatim, in a loop): quiet run vs
runs next to kernels of the product library on a second HIP stream; counts threads whose checksum differs and which lanes.
    python scripts/ubench/pk_race.py [libpk_race.so ...]"""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from lookoncetohear_amd import _cabi, config  # noqa: E402
from lookoncetohear_amd.net import Net  # noqa: E402

dev = torch.device("cuda:0")
lib = _cabi.load()
torch.manual_seed(0)
net = Net(**config.TSH_PARAMS).eval().to(dev)
bp = net._weights(dev)["blocks"][1]
nx = torch.randn(32, 625, 97, 64, device=dev)
nout = torch.empty_like(nx)
ws = net._workspace(32, 625, dev)
P = lambda t: t.data_ptr()
s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def noise(kind, st):
    if kind == "intra":
        lib.call("lh_intra_block", P(nx), P(bp["intra_w16"]), P(bp["intra_b16"]), P(bp["intra_lin_w2"]), P(bp["intra_lin_b"]), P(nout), 32 * 625, st)
    elif kind == "attn":
        lib.call("lh_local_attn", P(ws["q"]), P(ws["kx"]), P(ws["vx"]), P(nout), 32, 625, st)


BLOCKS, ITERS = 2048, 400
g = torch.Generator().manual_seed(1)
inp = torch.randn(BLOCKS * 256, 16, generator=g).to(dev)
for name in sys.argv[1:] or ["libpk_race.so", "libpk_race_lds.so"]:
    v = ctypes.CDLL(os.path.join(HERE, name))
    v.pk_victim.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    out = torch.zeros(BLOCKS * 256, 2, dtype=torch.int32, device=dev)
    v.pk_victim(P(inp), P(out), BLOCKS, ITERS, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    ref = out.clone()
    for kind in ("none", "intra", "attn"):
        bad_runs, bad_threads, lanes = 0, 0, set()
        for rep in range(10):
            with torch.cuda.stream(s0):
                if kind != "none":
                    for _ in range(4):
                        noise(kind, s0.cuda_stream)
            with torch.cuda.stream(s1):
                v.pk_victim(P(inp), P(out), BLOCKS, ITERS, s1.cuda_stream)
            torch.cuda.synchronize()
            diff = (out != ref).any(-1)
            n = int(diff.sum())
            if n:
                bad_runs += 1
                bad_threads += n
                lanes |= set((diff.nonzero().flatten() % 64).tolist())
        print(f"{name}: next to {kind:5s}: {bad_runs} of 10 runs differ, {bad_threads} threads; lanes {sorted(lanes)[:20]}{'...' if len(lanes) > 20 else ''}", flush=True)
