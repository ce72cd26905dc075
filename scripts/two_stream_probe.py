"""GPU probe (test tooling): VERDICT r2 item 3 — does running two half-batches on two HIP streams (so that one half's inter
pass, 97 of 256 CUs per half, overlaps the other half's frame kernels) beat one batch-32 forward?
    python scripts/two_stream_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lookoncetohear_amd import config, synth  # noqa: E402
from lookoncetohear_amd.net import Net  # noqa: E402

dev = torch.device("cuda:0")
sd = config.separator_weights(0)
nets = []
for _ in range(2):
    n = Net(**config.TSH_PARAMS).eval()
    n.load_state_dict(sd, strict=True)
    n.range_check = False                 # no host wait inside the forward: the two streams must be fed back to back
    nets.append(n.to(dev))
d = synth.batch(list(range(8)), 80000)
mix = d["mixture"].repeat(4, 1, 1).to(dev)
emb = d["embedding_gt"].repeat(4, 1, 1).to(dev)
halves = [(mix[:16].contiguous(), emb[:16].contiguous()), (mix[16:].contiguous(), emb[16:].contiguous())]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def one():
    return nets[0](mix, emb)


def two(offset_launch=False):
    cur = torch.cuda.current_stream(dev)
    for s in streams:
        s.wait_stream(cur)
    outs = []
    for (x, e), n, s in zip(halves, nets, streams):
        with torch.cuda.stream(s):
            outs.append(n(x, e))
    for s in streams:
        cur.wait_stream(s)
    return outs


def timeit(fn, reps=20):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for rnd in range(3):
    a = timeit(one)
    b = timeit(two)
    print(f"round {rnd}: one batch-32 forward {a:.3f} ms;  two batch-16 forwards on two streams {b:.3f} ms", flush=True)
with torch.no_grad():
    y1 = one()
    y2 = torch.cat(two(), 0)
    torch.cuda.synchronize()
print("max |one - two|", float((y1 - y2).abs().max()))
# where does a difference come from: the half-batch shape, the second instance / side stream, or the concurrency?
with torch.no_grad():
    ya = nets[0](*halves[0]); yb = nets[0](*halves[1])
    torch.cuda.synchronize()
    print("same net, default stream, sequential halves vs batch-32:", float((torch.cat([ya, yb]) - y1).abs().max()))
    with torch.cuda.stream(streams[1]):
        yc = nets[1](*halves[1])
    torch.cuda.synchronize()
    print("second net on a side stream, alone:", float((yc - y1[16:]).abs().max()))
    for k in range(3):
        o = two()
        torch.cuda.synchronize()
        print("concurrent pass", k, float((torch.cat(o, 0) - y1).abs().max()), "half0", float((o[0] - y1[:16]).abs().max()),
              "half1", float((o[1] - y1[16:]).abs().max()))
