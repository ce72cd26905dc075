"""Developer tool (GPU): ms per forward of ONE 5 s utterance over `Net.time_chunks_small` = K windows, each held bit for bit
against the whole-clip forward.   python scripts/time_b1.py [K ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lookoncetohear_amd import config, synth  # noqa: E402
from lookoncetohear_amd.net import Net  # noqa: E402

dev = torch.device("cuda", 0)
net = Net(**config.TSH_PARAMS).eval()
net.load_state_dict(config.separator_weights(0), strict=True)
net = net.to(dev)
d = synth.batch([0], 80000)
mix, emb = d["mixture"].to(dev), d["embedding_gt"].to(dev)


def ms(fn, steps=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


with torch.no_grad():
    net.time_chunks_small = 1
    y1 = net(mix, emb).clone()
    print("whole clip: %.3f ms" % ms(lambda: net(mix, emb)))
    for K in [int(a) for a in sys.argv[1:]] or [2, 3, 4, 5, 9]:
        net.time_chunks_small = K
        same = bool(torch.equal(net(mix, emb), y1))
        print("K = %d  %.3f ms   bit-identical %s   windows %s" % (K, ms(lambda: net(mix, emb)), same, net._window_bounds(1, 625, K)))
    net.time_chunks_small = 1
    print("whole clip: %.3f ms" % ms(lambda: net(mix, emb)))
