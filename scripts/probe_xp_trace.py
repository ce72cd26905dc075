"""Timing probe (not product code): s_memtime stamps of wave 0 of two workgroups of k_intra_xp (-DXP_TRACE build).
    LOOKONCE_HIP_LIB=$PWD/lookoncetohear_amd/_lookonce_hip_xptrace.so python scripts/probe_xp_trace.py [key=value ...]
Stamps per step: 0 = top of step (behind the barrier), 1 = phase H issued, 2 = phase C issued (in front of the barrier)."""
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
lib.call("lh_set_tuning", 2, 1)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    lib.call("lh_set_tuning", int(k), int(v))
torch.manual_seed(0)
net = Net(**config.TSH_PARAMS).eval().to(dev)
bp = net._weights(dev)["blocks"][0]
B, T = 32, 625
x = torch.randn(B, T, 97, 64, device=dev)
out = torch.zeros_like(x)
P = lambda t: t.data_ptr()
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    lib.call("lh_intra_block", P(x), P(bp["intra_w16"]), P(bp["intra_b16"]), P(bp["intra_lin_w2"]), P(bp["intra_lin_b"]), P(out), B * T, st)
torch.cuda.synchronize()
buf = np.zeros(2 * 128 * 4, dtype=np.uint64)
assert lib.raw("lh_probe_xp_trace_read")(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(2, 128, 4)[:, :97].astype(np.int64)
for w, name in enumerate(["workgroup 7", "workgroup n/2+3"]):
    tt = t[w][2:]          # the two peeled steps aside
    step = np.diff(tt[:, 0])
    print(name, sys.argv[1:], "ticks per step median %.0f p10 %.0f p90 %.0f | phase H %.0f  phase C %.0f  barrier %.0f" %
          (np.median(step), np.percentile(step, 10), np.percentile(step, 90), np.median(tt[:, 1] - tt[:, 0]),
           np.median(tt[:, 2] - tt[:, 1]), np.median(tt[1:, 0] - tt[:-1, 2])))
