"""Timing probe (not product code): s_memtime stamps of k_inter_xp's workgroup 3 (-DXP_TRACE build), wave 0 (projection
role) and wave 4 (LayerNorm role).
    LOOKONCE_HIP_LIB=$PWD/lookoncetohear_amd/_lookonce_hip_xptrace.so python scripts/probe_xp_trace_inter.py
Stamps per step: 0 = top of step, 1 = phase H issued, 2 = phase C issued, 3 = behind the barrier."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lookoncetohear_amd import _cabi, config  # noqa: E402
from lookoncetohear_amd.net import Net  # noqa: E402

dev = torch.device("cuda:0")
lib = _cabi.load()
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    lib.call("lh_set_tuning", int(k), int(v))
torch.manual_seed(0)
net = Net(**config.TSH_PARAMS).eval().to(dev)
bp = net._weights(dev)["blocks"][0]
B, T = 32, 625
x = torch.randn(B, T, 97, 64, device=dev)
out = torch.zeros_like(x)
h0 = torch.zeros(B * 97, 64, device=dev)
c0 = torch.zeros_like(h0)
hN, cN = torch.zeros_like(h0), torch.zeros_like(h0)
P = lambda t: t.data_ptr()
st = torch.cuda.current_stream(dev).cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(4):
    if i == 3:
        e0.record()
    lib.call("lh_inter_block", P(x), P(bp["inter_w8"]), P(bp["inter_b16"]), P(bp["inter_lin_wu"]), P(bp["inter_lin_b"]),
             P(h0), P(c0), P(hN), P(cN), P(out), B, T, st)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
buf = np.zeros(2 * 128 * 4, dtype=np.uint64)
assert lib.raw("lh_probe_xp_trace_inter_read")(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(2, 128, 4).astype(np.int64)
print(sys.argv[1:], "launch %.4f ms = %.1f ns per step" % (ms, ms * 1e6 / T))
for w, name in enumerate(["wave 0 (projection role)", "wave 4 (LayerNorm role)"]):
    tt = t[w]
    order = np.argsort(tt[:, 0])
    tt = tt[order][4:-4]                # ring of the last 128 steps, sorted by time; the wrap aside
    step = np.diff(tt[:, 0])
    tick_ns = ms * 1e6 / T / np.mean(step)
    print("%-26s ticks per step mean %.2f (1 tick = %.2f ns) | phase H %.2f  phase C %.2f  barrier wait %.2f  barrier->top %.2f" %
          (name, np.mean(step), tick_ns, np.mean(tt[:, 1] - tt[:, 0]), np.mean(tt[:, 2] - tt[:, 1]),
           np.mean(tt[:, 3] - tt[:, 2]), np.mean(tt[1:, 0] - tt[:-1, 3])))
