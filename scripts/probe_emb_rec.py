"""GPU probe (test tooling): where a step of k_emb_rec goes.  Needs the probe build
    python -m lookoncetohear_amd.build --variant ertrace -DER_TRACE
and prints, for waves 0 and 4 of workgroup 5 (forward direction), the mean cycles between the stamps of 64 steps:
0 step start -> 1 x group 0 issued -> 2 chain MFMAs issued -> 3 zip (42 MFMAs + cell update) done -> 4 before barrier -> 5 after barrier.
    LOOKONCE_HIP_LIB=$PWD/lookoncetohear_amd/_lookonce_hip_ertrace.so python scripts/probe_emb_rec.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lookoncetohear_amd import _cabi, config, synth  # noqa: E402
from lookoncetohear_amd.embed_net import EmbedTFGridNet  # noqa: E402

dev = torch.device("cuda:0")
lib = _cabi.load()
if os.environ.get("PROBE_TUNE"):
    k, v = os.environ["PROBE_TUNE"].split("=")
    lib.call("lh_set_tuning", int(k), int(v))
net = EmbedTFGridNet(**config.EMBED_PARAMS).eval()
net.load_state_dict(config.embedder_weights(0), strict=True)
net = net.to(dev)
net.n_streams = 1
B = int(os.environ.get("PROBE_B", "64"))
x = synth.batch(list(range(8)), 80000)["mixture"].repeat((B + 7) // 8, 1, 1)[:B].contiguous().to(dev)
rd = lib._dll.lh_probe_er_trace_read
rd.argtypes = [ctypes.c_void_p]
rd.restype = ctypes.c_int
with torch.no_grad():
    for _ in range(2):
        net(x)
torch.cuda.synchronize()
buf = np.zeros(2 * 64 * 8, dtype=np.uint64)
assert rd(buf.ctypes.data) == 0
t = buf.reshape(2, 64, 8).astype(np.float64)       # the LAST k_emb_rec launch of the forward (inter axis, block 2)
names = ["start->xg0", "xg0->chain", "chain->zip", "zip->pre-barrier", "barrier"]
for w in range(2):
    d = t[w, :, 1:6] - t[w, :, 0:5]
    step = t[w, 1:, 0] - t[w, :-1, 0]
    print(f"wave {4 * w}: cycles per step {step.mean():.0f} (min {step.min():.0f}, max {step.max():.0f}); " +
          ", ".join(f"{n} {v:.0f}" for n, v in zip(names, d.mean(0))))
    if t[w, :, 6].any():          # finer stamps inside the zip: after slot 23 and after slot 35 (of 42)
        z = np.stack([t[w, :, 6] - t[w, :, 2], t[w, :, 7] - t[w, :, 6], t[w, :, 3] - t[w, :, 7]], 1).mean(0)
        print(f"         zip: slots 0..23 {z[0]:.0f} ({z[0] / 24:.1f} per slot), 24..35 {z[1]:.0f} ({z[1] / 12:.1f}), 36..41 {z[2]:.0f} ({z[2] / 6:.1f})")
