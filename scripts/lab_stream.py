"""GPU lab for the two latency-bound single-sequence LSTM kernels (lh_stream.hip): times `lh_inter_matvec` (B = 1, T = 625:
97 workgroups x 625 dependent steps) and `lh_intra_stream` (one frame: 2 workgroups x 97 steps) in isolation.
    LOOKONCE_HIP_LIB=lookoncetohear_amd/_lookonce_hip_qs1.so python scripts/lab_stream.py
Used with the -DQS_PROBE=n builds to price the parts of the step (profiles/r03k_*)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lookoncetohear_amd import _cabi, config  # noqa: E402
from lookoncetohear_amd.net import Net  # noqa: E402

dev = torch.device("cuda:0")
lib = _cabi.load()
torch.manual_seed(0)
net = Net(**config.TSH_PARAMS).eval().to(dev)
bp = net._weights(dev)["blocks"][0]
P = lambda t: t.data_ptr()
st = torch.cuda.current_stream(dev).cuda_stream


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B, T = 1, 625
x = torch.randn(B, T, 97, 64, device=dev)
out = torch.zeros_like(x)
h0 = torch.randn(B * 97, 64, device=dev) * 0.3
c0 = torch.randn(B * 97, 64, device=dev) * 0.3
hN, cN = torch.zeros_like(h0), torch.zeros_like(c0)
ms = timed(lambda: lib.call("lh_inter_matvec", P(x), P(bp["inter_s_wih"]), P(bp["inter_s_b"]), P(bp["inter_s_whh"]),
                            P(bp["inter_lin_w"]), P(bp["inter_lin_b"]), P(h0), P(c0), P(hN), P(cN), P(out), B, T, st))
x1 = torch.randn(1, 1, 97, 64, device=dev)
hb = torch.zeros(97, 128, device=dev)
ms2 = timed(lambda: lib.call("lh_intra_stream", P(x1), P(bp["intra_s_wih"]), P(bp["intra_s_b"]), P(bp["intra_s_whh"]), P(hb), 1, st), 200)
print("%-30s inter_matvec %.4f ms = %.1f ns/step   intra_stream %.2f us = %.1f ns/step (incl. its prologue)" %
      (os.path.basename(lib.path), ms, ms * 1e6 / T, ms2 * 1e3, ms2 * 1e6 / 97), flush=True)
