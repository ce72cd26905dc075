import sys, torch
sys.path.insert(0, "/root/repo")
from lookoncetohear_amd import synth, config
from lookoncetohear_amd.net import Net
from oracle import tfgridnet_oracle as O
cfg = O.Cfg(**config.TSH_PARAMS); sd = config.separator_weights(0)
net = Net(**config.TSH_PARAMS).eval(); net.load_state_dict(sd, strict=True); net = net.to("cuda:0")
d = synth.batch([0, 1], 16000)
taps, otaps = {}, {}
net._debug_taps = taps
with torch.no_grad():
    y = net(d["mixture"].to("cuda:0"), d["embedding_gt"].to("cuda:0"))
net._debug_taps = None
yo = O.forward(cfg, sd, d["mixture"], d["embedding_gt"], dtype=torch.float64, fast_lstm=True, taps=otaps)
print("y", float((y.cpu().double() - yo).abs().max()))
for k in taps:
    if k in otaps:
        o = otaps[k].reshape(taps[k].shape) if k != "blocks.0.out" else None
        if o is not None:
            print(k, f"{float((taps[k].cpu().double() - o).abs().max()):.3e}  amp {float(o.abs().max()):.2f}")
import os
for mode in ("LOOKONCE_FUSE",):
    pass
