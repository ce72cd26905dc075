"""Timing probe (not product code): per-step s_memtime stamps of two workgroups of the fused intra kernel.

Run on the GPU box with the -DLH_PROBE_TRACE build:
    LOOKONCE_HIP_LIB=$PWD/lookoncetohear_amd/_lookonce_hip_probe_TRACE.so python scripts/probe_trace.py
Stamps per step: 0 = top of step (behind the barrier), 1 = row-wise work + global traffic issued, 2 = all MFMAs issued,
3 = cell update issued (in front of the barrier).  s_memtime ticks at a constant 100 MHz.
"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from lookoncetohear_amd import _cabi, config, synth  # noqa: E402
from lookoncetohear_amd.net import Net  # noqa: E402

tune = [kv.split("=") for kv in sys.argv[1:]]
net = Net(**config.TSH_PARAMS).eval()
net.load_state_dict(config.separator_weights(0), strict=True)
net = net.to("cuda:0")
lib = _cabi.load()
for k, v in tune:
    lib.raw("lh_set_tuning")(int(k), int(v))
d = synth.batch(list(range(32)), 80000)
x, e = d["mixture"].to("cuda:0"), d["embedding_gt"].to("cuda:0")
with torch.no_grad():
    for _ in range(3):
        net(x, e)
torch.cuda.synchronize()
buf = np.zeros(2 * 128 * 4, dtype=np.uint64)
rc = lib.raw("lh_probe_trace_read")(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
t = buf.reshape(2, 128, 4)[:, :97].astype(np.int64)
for w, name in enumerate(["workgroup 7 (first round)", "workgroup n-9 (last round)"]):
    tt = t[w]
    step = np.diff(tt[:, 0])
    seg = np.stack([tt[:, 1] - tt[:, 0], tt[:, 2] - tt[:, 1], tt[:, 3] - tt[:, 2]], 1)
    bar = tt[1:, 0] - tt[:-1, 3]
    print(name, "ticks per step (median / p10 / p90):", np.median(step), np.percentile(step, 10), np.percentile(step, 90))
    print("   segments median: rows %.1f  mfma %.1f  cells %.1f  barrier %.1f" %
          (np.median(seg[:, 0]), np.median(seg[:, 1]), np.median(seg[:, 2]), np.median(bar)))
    print("   first 12 steps:", step[:12].tolist())

# back end (k_deconv_istft): stamps of workgroup 3, third tile of its run
bb = np.zeros(32, dtype=np.uint64)
if lib.raw("lh_probe_be_trace_read")(bb.ctypes.data_as(ctypes.c_void_p)) == 0:
    b = bb.astype(np.int64)
    names = ["tile start -> ring primed", "primed -> loop", "frames 0-3", "frames 4-7", "frames 8-11", "frames 12-14",
             "loop end -> barrier", "", "halo + synthesis", "overlap-add + re-zero"]
    seq = [0, 1, 2, 3, 4, 5, 8, 9, 10]
    print("back end tile (cycles):", {f"{seq[i]}->{seq[i+1]}": int(b[seq[i + 1]] - b[seq[i]]) for i in range(len(seq) - 1)},
          "total", int(b[10] - b[0]))
    print("one pair of frames: stage + barrier", int(b[12] - b[11]), " loads issued + 42 tile products + barrier", int(b[13] - b[12]),
          " gather + spectrum split", int(b[14] - b[13]), "| inside the product phase: load issue", int(b[15] - b[12]),
          "wave 0's products", int(b[16] - b[15]), "barrier wait", int(b[13] - b[16]))
    print("tile start: Sx copy + barrier", int(b[17] - b[0]), " setup", int(b[18] - b[17]), " ring priming (16 loads issued)", int(b[1] - b[18]))

# QKV frame kernel: one frame of workgroup 5
qq = np.zeros(16, dtype=np.uint64)
if lib.raw("lh_probe_qkv_trace_read")(qq.ctypes.data_as(ctypes.c_void_p)) == 0:
    q = qq.astype(np.int64)
    print("QKV frame (cycles): stage + barrier", int(q[1] - q[0]), " prefetch issue + 49 products", int(q[2] - q[1]), " barrier", int(q[3] - q[2]),
          " Q/K LayerNorm + stores", int(q[4] - q[3]), " V LayerNorm + stores", int(q[5] - q[4]), " total", int(q[5] - q[0]))

# attention: workgroup 803 (wave 0)
aa = np.zeros(16, dtype=np.uint64)
if lib.raw("lh_probe_attn_trace_read")(aa.ctypes.data_as(ctypes.c_void_p)) == 0:
    a = aa.astype(np.int64)
    print("attention workgroup (cycles): scores", int(a[1] - a[0]), " barrier", int(a[2] - a[1]), " softmax", int(a[3] - a[2]),
          " barrier", int(a[4] - a[3]), " P.V + store", int(a[5] - a[4]), " total", int(a[5] - a[0]))
