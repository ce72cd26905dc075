"""GPU probe (test tooling): energy per C-ABI call of the B=32 forward.  Records the argument lists of one `Net.forward`,
then replays each distinct call (its block-0 instance) in a loop for a few seconds while a second thread samples rocm-smi;
prints ms per call, average package power, sclk and joules per call, and the same for the whole forward.
    python scripts/power_by_call.py [key=value,...]"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lookoncetohear_amd import _cabi, config, synth  # noqa: E402
from lookoncetohear_amd.net import Net  # noqa: E402

dev = torch.device("cuda:0")
lib = _cabi.load()
for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else "").split(",")):
    k, v = kv.split("=")
    lib.call("lh_set_tuning", int(k), int(v))
EMBED = os.environ.get("PROBE_MODEL", "separator") == "embed"      # PROBE_MODEL=embed: the enrollment embedder (B = 64, one stream)
if EMBED:
    from lookoncetohear_amd.embed_net import EmbedTFGridNet  # noqa: E402
    net = EmbedTFGridNet(**config.EMBED_PARAMS).eval()
    net.load_state_dict(config.embedder_weights(0), strict=True)
    net = net.to(dev)
    net.n_streams = 1
    B = int(os.environ.get("PROBE_B", "64"))
    x = synth.batch(list(range(8)), 80000)["mixture"].repeat((B + 7) // 8, 1, 1)[:B].contiguous().to(dev)
    e = None
    _fwd = lambda: net(x)
else:
    net = Net(**config.TSH_PARAMS).eval()
    net.load_state_dict(config.separator_weights(0), strict=True)
    net = net.to(dev)
    B = int(os.environ.get("PROBE_B", "32"))
    d = synth.batch(list(range(min(B, 8))), 80000)
    x = d["mixture"].repeat((B + 7) // 8, 1, 1)[:B].contiguous().to(dev)
    e = d["embedding_gt"].repeat((B + 7) // 8, 1, 1)[:B].contiguous().to(dev)
    _fwd = lambda: net(x, e)
with torch.no_grad():
    _fwd()
torch.cuda.synchronize()


class Rec:
    def __init__(self, lib):
        self.lib, self.calls, self.path = lib, [], lib.path

    def call(self, name, *args):
        self.calls.append((name, args))
        self.lib.call(name, *args)

    def raw(self, name):
        return self.lib.raw(name)


rec = Rec(lib)
net._lib = lambda t: rec          # instance-level override of HipHost._lib: record the calls of one forward
with torch.no_grad():
    _keep = _fwd()          # (the embedder's scratch tensors are freed after the call: keep the allocator from re-using them
torch.cuda.synchronize()   #  is not possible, so the replayed calls below write into freed-but-still-mapped blocks — timing only)
del net._lib
SECS = float(os.environ.get("PROBE_SECS", "2.5"))
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = re.search(r"Power \(W\):\s*([\d.]+)", o)
            sc = re.search(r"sclk clock level:\s*\S+\s*\((\d+)Mhz\)", o)
            if pw:
                samples.append((float(pw.group(1)), int(sc.group(1)) if sc else 0))
        except Exception:  # noqa: BLE001
            pass


def measure(fn, label, per=1):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    samples.clear()
    stop[0] = False
    th = threading.Thread(target=sampler)
    th.start()
    n, t0 = 0, time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < SECS:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop[0] = True
    th.join()
    ms = e0.elapsed_time(e1) / n
    s = samples[1:] or samples
    pw = sum(a for a, _ in s) / max(len(s), 1)
    sc = sum(b for _, b in s) / max(len(s), 1)
    print(f"{label:22s} x{per}  {ms:8.4f} ms  {pw:6.0f} W  sclk {sc:5.0f} MHz  {ms * pw / 1000:7.3f} J/call  -> {per * ms * pw / 1000:7.3f} J/step", flush=True)
    return per * ms, per * ms * pw / 1000


seen, tot_ms, tot_j = {}, 0.0, 0.0
count = {}
for name, _ in rec.calls:
    count[name] = count.get(name, 0) + 1
ONLY = set(filter(None, os.environ.get("PROBE_CALLS", "").split(",")))      # e.g. lh_intra_block,lh_inter_block
for name, args in rec.calls:
    key = name + (".inter" if name == "lh_emb_axis_fused" and args[10] else "")      # the two axis calls are different kernels
    if key in seen or (ONLY and name not in ONLY):
        continue
    seen[key] = True
    a, b = measure(lambda: lib.call(name, *args), key, count[name] // (2 if name == "lh_emb_axis_fused" else 1))
    tot_ms += a
    tot_j += b
print(f"sum over calls: {tot_ms:.3f} ms, {tot_j:.3f} J per step")


def fwd():
    with torch.no_grad():
        _fwd()


measure(fwd, "whole forward")
