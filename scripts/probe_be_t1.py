"""Timing probe (not product code): s_memtime stamps of k_deconv_istft in the streaming shape (B = 1, T = 1: one workgroup,
one tile).  Build with -DLH_PROBE_TRACE -DLH_PROBE_TRACE_T1:
    LOOKONCE_HIP_LIB=$PWD/lookoncetohear_amd/_lookonce_hip_bet1.so python scripts/probe_be_t1.py"""
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
torch.manual_seed(0)
net = Net(**config.TSH_PARAMS).eval().to(dev)
pk = net._weights(dev)
P = lambda t: t.data_ptr()
st = torch.cuda.current_stream(dev).cuda_stream
B, T = 1, 1
xa = torch.randn(B, T, 97, 64, device=dev)
state = net.init_buffers(B, dev)
dec_in, ist_in = state["deconv_buf"].clone().normal_(), state["istft_buf"].clone().normal_()
dec_out, ist_out = torch.zeros_like(dec_in), torch.zeros_like(ist_in)
y = torch.zeros(B, 2, 128, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(20):
    if i == 10:
        e0.record()
    lib.call("lh_deconv_istft", P(xa), P(dec_in), P(dec_out), P(ist_in), P(ist_out), P(pk["deconv_w"]), P(pk["deconv_b"]),
             P(pk["wfb_dec"]), P(y), None, 0, B, T, st)
e1.record()
torch.cuda.synchronize()
print("lh_deconv_istft B=1 T=1: %.2f us per call (back to back)" % (e0.elapsed_time(e1) * 100))
bb = np.zeros(32, dtype=np.uint64)
if lib.raw("lh_probe_be_trace_read")(bb.ctypes.data_as(ctypes.c_void_p)) == 0:
    b = bb.astype(np.int64)
    order = [(20, "kernel entry"), (21, "prologue (taps to LDS, zero fill) done"), (0, "tile start"), (17, "Sx[0] from the carried spectrum"),
             (18, "setup"), (1, "ring primed (halo + frame loads issued)"), (2, "loop entry"), (8, "frame loop done (3 frames: 2 iterations)"),
             (9, "halo state + synthesis done"), (10, "overlap-add + re-zero done")]
    prev = None
    for k, name in order:
        if prev is not None:
            print("  %-52s +%6d ticks" % (name, b[k] - prev))
        prev = b[k]
    print("  total %d ticks" % (b[10] - b[20]))
