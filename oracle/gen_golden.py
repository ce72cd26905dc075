"""TEST INFRASTRUCTURE. Writes tests/golden/separator_golden.npz from the *unmodified reference model*
(imported from /root/reference under oracle/ref_stubs.py; only possible in the build container):

    python -m oracle.gen_golden

Inputs are regenerated at test time from seeds (lookoncetohear_amd.synth + oracle.synthetic_state_dict),
so only reference OUTPUTS (and strided subsamples of hooked intermediates / states) are stored.
Both the fp32 reference output (`*_y32`) and the fp64 reference output (`*_y64`, the error-floor-free truth,
stored rounded to fp32) are kept.
"""
import os
import sys

import numpy as np
import torch

from oracle import ref_stubs
from oracle import tfgridnet_oracle as O
from lookoncetohear_amd import synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "separator_golden.npz")
STREAM_CHUNKS = 60
STATE_SEED = 3


def hooked_forward(net, x, e, st):
    """Run the reference with forward hooks; returns (y, taps) with taps in the oracle's naming/layout."""
    taps = {}
    tg = net.tfgridnet
    hs = []

    def add(mod, name, fn):
        hs.append(mod.register_forward_hook(lambda m, i, o: taps.__setitem__(name, fn(o))))

    add(tg.conv, "Z0", lambda o: o.permute(0, 2, 3, 1))
    add(tg.embed_to_feats_proj, "G", lambda o: o.reshape(o.shape[0], 64, 97).permute(0, 2, 1).unsqueeze(1))
    for i, blk in enumerate(tg.blocks):
        add(blk.attn_conv_Q, f"blocks.{i}.Q", lambda o: o)
        add(blk.attn_conv_K, f"blocks.{i}.K", lambda o: o)
        add(blk.attn_conv_V, f"blocks.{i}.V", lambda o: o)
        add(blk, f"blocks.{i}.out", lambda o: o[0].permute(0, 2, 3, 1))
    with torch.no_grad():
        y = net(x, e, input_state=st)
    for h in hs:
        h.remove()
    return y, taps


def main():
    torch.set_num_threads(8)
    Net = ref_stubs.reference_net_class()
    cfg = O.Cfg(**O.TSH_PARAMS)
    sd = O.synthetic_state_dict(cfg, seed=0)
    nets = {}
    for dt in (torch.float32, torch.float64):
        n = Net(**O.TSH_PARAMS).eval()
        n.load_state_dict(sd, strict=True)
        nets[dt] = n.to(dt)
    g = {}

    def run(dt, fn):
        with torch.no_grad():
            return fn(nets[dt], dt)

    # 1/2: offline, zero state (second one exercises mod_pad, net.py:8-18)
    for name, idx, n in (("off_b2_n8000", [0, 1], 8000), ("off_b1_n8100", [2], 8100)):
        d = synth.batch(idx, n)
        for dt, tag in ((torch.float32, "y32"), (torch.float64, "y64")):
            y = run(dt, lambda net, dt: net(d["mixture"].to(dt), d["embedding_gt"].to(dt),
                                            input_state=O.init_state(cfg, len(idx), dt)))
            g[f"{name}_{tag}"] = y.float().numpy()

    # 3: non-zero state in, state out (predict, pad=False)
    d = synth.batch([3, 4], 128 * 12 + 64)
    for dt, tag in ((torch.float32, "32"), (torch.float64, "64")):
        st = O.random_state(cfg, 2, STATE_SEED, dt)
        y, st2 = run(dt, lambda net, dt: net.predict(d["mixture"].to(dt), d["embedding_gt"][:, 0].to(dt), st, pad=False))
        g[f"state_b2_y{tag}"] = y.float().numpy()
        for k, v in O.flat_state(st2).items():
            g[f"state_b2_s{tag}.{k}"] = O.subsample(v.float(), 256).numpy()

    # 4: streaming, 8 ms chunks with 4 ms look-ahead
    d = synth.batch([5], 128 * STREAM_CHUNKS + 64)
    for dt, tag in ((torch.float32, "y32"), (torch.float64, "y64")):
        def stream(net, dt):
            st = O.init_state(cfg, 1, dt)
            outs = []
            for i in range(STREAM_CHUNKS):
                ch = d["mixture"][:, :, i * 128:i * 128 + 192].to(dt)
                y, st = net.predict(ch, d["embedding_gt"][:, 0].to(dt), st, pad=False)
                outs.append(y)
            return torch.cat(outs, -1)
        g[f"stream_b1_{tag}"] = run(dt, stream).float().numpy()

    # 5: full-size 5 s clip with hooked intermediates (strided subsamples only)
    d = synth.batch([6], 80000)
    for dt, tag in ((torch.float32, "32"), (torch.float64, "64")):
        y, taps = hooked_forward(nets[dt], d["mixture"].to(dt), d["embedding_gt"].to(dt), O.init_state(cfg, 1, dt))
        g[f"full_b1_y{tag}"] = y.float().numpy()[:, :, ::8]
        g[f"full_b1_stats{tag}"] = np.array([y.abs().max().item(), y.pow(2).mean().sqrt().item(), y.sum().item()])
        for k, v in taps.items():
            g[f"full_b1_t{tag}.{k}"] = O.subsample(v.float()).numpy()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, "%.1f KB" % (os.path.getsize(OUT) / 1024), len(g), "arrays")
    return 0


if __name__ == "__main__":
    sys.exit(main())
