"""Developer tool (GPU): ms per forward of the enrollment embedder over the batch size (5 s clips).  python scripts/embed_sweep.py [B ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lookoncetohear_amd import config, synth  # noqa: E402
from lookoncetohear_amd.embed_net import EmbedTFGridNet  # noqa: E402

dev = torch.device("cuda", 0)
net = EmbedTFGridNet(**config.EMBED_PARAMS).eval()
net.load_state_dict(config.embedder_weights(0), strict=True)
net = net.to(dev)
d = synth.batch(list(range(8)), 80000)["mixture"]
with torch.no_grad():
    for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64]:
        x = d.repeat((B + 7) // 8, 1, 1)[:B].contiguous().to(dev)
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
        n = 10 if B <= 16 else 4
        t0 = time.perf_counter()
        for _ in range(n):
            net(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print("B = %3d   %8.3f ms per forward   %7.3f ms per clip   %8.1f clips/s" % (B, ms, ms / B, B / ms * 1e3))
        torch.cuda.empty_cache()
