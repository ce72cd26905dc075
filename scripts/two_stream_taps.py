"""GPU probe (test tooling): which stage of the forward goes wrong when ANOTHER forward runs concurrently on a second
HIP stream?  Stream 0 is kept busy with queued forwards of net 0 while net 1 runs ONE tapped forward on stream 1; its
stage taps are compared with the taps of the same forward run alone."""
import os
import sys

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
    n.range_check = False
    nets.append(n.to(dev))
d = synth.batch(list(range(8)), 80000)
mix = d["mixture"].repeat(4, 1, 1).to(dev)
emb = d["embedding_gt"].repeat(4, 1, 1).to(dev)
halves = [(mix[:16].contiguous(), emb[:16].contiguous()), (mix[16:].contiguous(), emb[16:].contiguous())]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
with torch.no_grad():
    ref = {}
    nets[1]._debug_taps = ref
    with torch.cuda.stream(streams[1]):
        ref["y"] = nets[1](*halves[1])
    nets[0](*halves[0])
    torch.cuda.synchronize()
    for rep in range(3):
        taps = {}
        nets[1]._debug_taps = taps
        cur = torch.cuda.current_stream(dev)
        for s in streams:
            s.wait_stream(cur)
        with torch.cuda.stream(streams[0]):
            for _ in range(4):
                nets[0](*halves[0])
        with torch.cuda.stream(streams[1]):
            taps["y"] = nets[1](*halves[1])
        torch.cuda.synchronize()
        bad = [(k, float((taps[k].float() - ref[k].float()).abs().max())) for k in ref]
        print(rep, [(k, f"{v:.2e}") for k, v in bad if v > 0] or "all stages identical")
        # are the wrong values of the first bad `.out` tap the PREVIOUS content of that buffer (xa: Z0 -> blocks.0.out ->
        # blocks.1.out -> ...), i.e. stale lines?
        outs = ["Z0"] + [f"blocks.{i}.out" for i in range(3)]
        for i in range(1, 4):
            k, prev = outs[i], outs[i - 1]
            m = taps[k] != ref[k]
            if m.any():
                same_prev = (taps[k][m] == ref[prev][m]).float().mean().item()
                rows = m.reshape(16, 625, 97, 64).any(-1)          # [b][t][f]
                frames = rows.any(-1)
                print(f"   first bad tap {k}: {int(m.sum())} wrong values; fraction equal to the buffer's previous content ({prev}): {same_prev:.3f};"
                      f" frames hit {int(frames.sum())} (full frames: {int(rows.all(-1).sum())}); utterances {frames.any(-1).nonzero().flatten().tolist()}"
                      f" first frames {frames.nonzero()[:6].tolist()}")
                b0, t0_ = frames.nonzero()[0].tolist()
                dfr = (taps[k].reshape(16, 625, 97, 64)[b0, t0_] - ref[k].reshape(16, 625, 97, 64)[b0, t0_]).double()
                rfr = ref[k].reshape(16, 625, 97, 64)[b0, t0_].double()
                print(f"   frame ({b0},{t0_}): diff min {dfr.min():.3e} max {dfr.max():.3e} mean {dfr.mean():.3e} std {dfr.std():.3e};"
                      f" |diff|>1e-4: {int((dfr.abs() > 1e-4).sum())} of {dfr.numel()}; per-bin max|diff| first 8 bins {dfr.abs().amax(-1)[:8].tolist()}")
                break
