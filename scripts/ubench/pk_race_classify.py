"""What does the corrupted v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0] return in lanes 48..63?  One execution per thread
(ITERS = 1, so the checksums are the raw result bits), run next to the intra LSTM kernel; every wrong result is matched
against the candidate sums of its inputs."""
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
P = lambda t: t.data_ptr()
s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
v = ctypes.CDLL(os.path.join(HERE, sys.argv[1] if len(sys.argv) > 1 else "libpk_race_t7_add_opsel.so"))
v.pk_victim.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
BLOCKS = 8192
g = torch.Generator().manual_seed(1)
inp = torch.randn(BLOCKS * 256, 16, generator=g).to(dev)
inp[:, 13] = inp[:, 12] + 1.0        # (unused slot) ; the harness copies v134 -> v135, so src1.lo == src1.hi == mean
out = torch.zeros(BLOCKS * 256, 2, dtype=torch.int32, device=dev)
v.pk_victim(P(inp), P(out), BLOCKS, 1, torch.cuda.current_stream(dev).cuda_stream)
torch.cuda.synchronize()
ref = out.clone()
d0, d1, m = inp[:, 0], inp[:, 1], inp[:, 12]
cands = {"src0.lo + src1": d0 + m, "src0.hi + src1": d1 + m, "src0.lo": d0, "src0.hi": d1, "src1": m, "2 src1": m + m,
         "src0.lo + src0.hi": d0 + d1, "0": torch.zeros_like(m)}
assert torch.equal(ref[:, 0].view(torch.float32), cands["src0.lo + src1"]) and torch.equal(ref[:, 1].view(torch.float32), cands["src0.hi + src1"])
tally = {}
for rep in range(30):
    with torch.cuda.stream(s0):
        for _ in range(3):
            lib.call("lh_intra_block", P(nx), P(bp["intra_w16"]), P(bp["intra_b16"]), P(bp["intra_lin_w2"]), P(bp["intra_lin_b"]), P(nout), 32 * 625, s0.cuda_stream)
    with torch.cuda.stream(s1):
        v.pk_victim(P(inp), P(out), BLOCKS, 1, s1.cuda_stream)
    torch.cuda.synchronize()
    for half, name in ((0, "lo result"), (1, "hi result")):
        bad = (out[:, half] != ref[:, half]).nonzero().flatten()
        if len(bad) == 0:
            continue
        got = out[bad, half].view(torch.float32)
        for cn, cv in cands.items():
            n = int((got == cv[bad]).sum())
            if n:
                tally[(name, cn)] = tally.get((name, cn), 0) + n
        tally[(name, "total wrong")] = tally.get((name, "total wrong"), 0) + len(bad)
        tally[(name, "lanes")] = sorted(set(tally.get((name, "lanes"), [])) | set((bad % 64).tolist()))
for k in sorted(tally):
    print(k, tally[k])
