"""Developer tool (GPU): ms per forward and frames/s of the separator over the batch size (5 s clips) — where the path turns
from latency-bound (one utterance: dependent LSTM steps on a fraction of the CUs) into throughput-bound.
    python scripts/batch_sweep.py [B ...]"""
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
d = synth.batch(list(range(8)), 80000)
with torch.no_grad():
    for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 5, 6, 8, 13, 14, 16, 32, 64]:
        mix = d["mixture"].repeat((B + 7) // 8, 1, 1)[:B].contiguous().to(dev)
        emb = d["embedding_gt"].repeat((B + 7) // 8, 1, 1)[:B].contiguous().to(dev)
        for _ in range(5):
            net(mix, emb)
        torch.cuda.synchronize()
        n = 40 if B <= 16 else 15
        t0 = time.perf_counter()
        for _ in range(n):
            net(mix, emb)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print("B = %3d   %7.3f ms per forward   %6.3f ms per clip   %9.0f frames/s   windows %d" %
              (B, ms, ms / B, B * 625 / ms * 1e3, net._n_time_chunks(B, 625, 1)))
        net._ws.clear()
        torch.cuda.empty_cache()
