"""GPU probe (test tooling): runs one C-ABI call in a loop for a few seconds per setting while sampling rocm-smi power /
clocks from a second thread; prints ms per call, average power, average sclk.   python scripts/power_probe.py "_" "2=1" ..."""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lookoncetohear_amd import _cabi, config  # noqa: E402
from lookoncetohear_amd.net import Net  # noqa: E402

dev = torch.device("cuda:0")
lib = _cabi.load()
torch.manual_seed(0)
net = Net(**config.TSH_PARAMS).eval().to(dev)
bp = net._weights(dev)["blocks"][0]
B, T = 32, 625
x = torch.randn(B, T, 97, 64, device=dev)
out = torch.zeros_like(x)
h0 = torch.randn(B * 97, 64, device=dev) * 0.3
c0 = torch.randn(B * 97, 64, device=dev) * 0.3
hN, cN = torch.zeros_like(h0), torch.zeros_like(c0)
P = lambda t: t.data_ptr()
st = torch.cuda.current_stream(dev).cuda_stream
which = os.environ.get("PROBE_WHICH", "intra")


def call():
    if which == "intra":
        lib.call("lh_intra_block", P(x), P(bp["intra_w16"]), P(bp["intra_b16"]), P(bp["intra_lin_w2"]), P(bp["intra_lin_b"]), P(out), B * T, st)
    else:
        lib.call("lh_inter_block", P(x), P(bp["inter_w8"]), P(bp["inter_b16"]), P(bp["inter_lin_wu"]), P(bp["inter_lin_b"]), P(h0), P(c0), P(hN), P(cN), P(out), B, T, st)


samples = []
stop = False


def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = re.search(r"Power \(W\):\s*([\d.]+)", o)
            sc = re.search(r"sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)", o)
            samples.append((float(pw.group(1)) if pw else None, int(sc.group(1)) if sc else None, o if not pw else ""))
        except Exception as e:  # noqa: BLE001
            samples.append((None, None, repr(e)))


for tune in sys.argv[1:] or ["_"]:
    keys = []
    if tune != "_":
        for kv in tune.split(","):
            k, v = kv.split("=")
            lib.call("lh_set_tuning", int(k), int(v))
            keys.append(int(k))
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    samples.clear()
    stop = False
    th = threading.Thread(target=sampler)
    th.start()
    n = 0
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < float(os.environ.get("PROBE_SECS", "4")):
        for _ in range(50):
            call()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop = True
    th.join()
    ms = e0.elapsed_time(e1) / n
    pw = [s[0] for s in samples if s[0] is not None]
    sc = [s[1] for s in samples if s[1] is not None]
    print(f"{which} tune {tune:12s} {ms:.4f} ms/call  power W avg {sum(pw) / max(len(pw), 1):.0f} max {max(pw) if pw else 0:.0f} ({len(pw)} samples)"
          f"  sclk MHz avg {sum(sc) / max(len(sc), 1):.0f} min {min(sc) if sc else 0}", flush=True)
    if not pw and samples:
        print(samples[0][2][:1500])
    for k in keys:
        lib.call("lh_set_tuning", k, 1 if k == 3 else 0)
