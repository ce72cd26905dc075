"""GPU probe (test tooling): B=32 x 5 s forward under a list of lh_set_tuning settings against the fp64 CPU oracle on two
rows, and against each other on all rows.    python scripts/parity_probe.py "_" "2=2,5=2"   (LOOKONCE_HIP_LIB selects the build)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lookoncetohear_amd import _cabi, synth  # noqa: E402
from lookoncetohear_amd.net import Net  # noqa: E402
from oracle import tfgridnet_oracle as O  # noqa: E402

dev = torch.device("cuda:0")
lib = _cabi.load()
cfg = O.Cfg(**O.TSH_PARAMS)
sd = O.synthetic_state_dict(cfg, seed=0)
net = Net(**O.TSH_PARAMS).eval()
net.load_state_dict(sd, strict=True)
net = net.to(dev)
B = int(os.environ.get("PROBE_B", "32"))
d = synth.batch(list(range(100, 100 + B)), 80000)
x, e = d["mixture"].to(dev), d["embedding_gt"].to(dev)
rows = (0, B - 1)
yo = {r: O.forward(cfg, sd, d["mixture"][r:r + 1], d["embedding_gt"][r:r + 1], dtype=torch.float64, fast_lstm=True) for r in rows}
amp = max(float(v.abs().max()) for v in yo.values())
first = None
for tune in sys.argv[1:] or ["_"]:
    keys = []
    if tune != "_":
        for kv in tune.split(","):
            k, v = kv.split("=")
            lib.call("lh_set_tuning", int(k), int(v))
            keys.append(int(k))
    with torch.no_grad():
        y = net(x, e).cpu()
    errs = [float((y[r:r + 1].double() - yo[r]).abs().max()) for r in rows]
    if first is None:
        first = y
    print(f"{os.path.basename(lib.path)} tune {tune:12s} max|hip - oracle fp64| rows {rows}: {errs[0]:.3e} {errs[1]:.3e}  (amp {amp:.2f})"
          f"  vs first setting, all rows: {float((y - first).abs().max()):.3e}", flush=True)
    for k in keys:
        lib.call("lh_set_tuning", k, 1 if k == 3 else 0)
