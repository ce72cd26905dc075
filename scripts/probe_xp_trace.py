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

# workgroup log of the last forward-direction launch: residency per CU over time
log = np.zeros(4096 * 4, dtype=np.uint64)
if lib.raw("lh_probe_xp_wglog_read")(log.ctypes.data_as(ctypes.c_void_p)) == 0:
    n = (B * T + 15) // 16
    L = log.reshape(4096, 4)[:n].astype(np.int64)
    t0, t1, hw, xcc = L[:, 0], L[:, 1], L[:, 2], L[:, 3] & 0xF
    # HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3
    key = xcc * 4096 + se * 64 + sh * 32 + cu
    print("workgroups", n, "distinct (xcc, se, sh, cu):", len(np.unique(key)), " distinct xcc:", len(np.unique(xcc)))
    span = t1.max() - t0.min()
    dur = t1 - t0
    print("launch span ticks %d; workgroup duration ticks median %d p10 %d p90 %d min %d max %d" %
          (span, np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), dur.min(), dur.max()))
    # residency: average number of workgroups alive per CU over the launch span (sampled)
    ts = np.linspace(t0.min(), t1.max(), 200)
    alive = [(np.sum((t0 <= t) & (t1 > t))) for t in ts]
    print("alive workgroups over time (20 samples):", [int(a) for a in alive[5::10]])
    # co-residency of the per-CU neighbours
    per = {}
    for k, a, b in zip(key, t0, t1):
        per.setdefault(int(k), []).append((int(a), int(b)))
    mx = []
    for k, iv in per.items():
        ev = sorted([(a, 1) for a, b in iv] + [(b, -1) for a, b in iv])
        c = m = 0
        for _, dlt in ev:
            c += dlt; m = max(m, c)
        mx.append(m)
    print("max concurrent workgroups per CU: histogram", np.bincount(mx).tolist(), " workgroups per CU: min %d max %d" % (min(len(v) for v in per.values()), max(len(v) for v in per.values())))
