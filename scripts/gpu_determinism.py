"""Runs the separator repeatedly on the same batch; reports per stage tap the run-to-run differences and, for the
first differing stage, where the differing elements sit and which run agrees with the CPU oracle."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lookoncetohear_amd import synth
from lookoncetohear_amd.net import Net
from oracle import tfgridnet_oracle as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
cfg = O.Cfg(**O.TSH_PARAMS); sd = O.synthetic_state_dict(cfg, 0)
net = Net(**O.TSH_PARAMS).eval(); net.load_state_dict(sd, strict=True); net = net.to("cuda:0")
d = synth.batch(list(range(B)), N)
x, e = d["mixture"].cuda(), d["embedding_gt"].cuda()
runs = []
for r in range(3):
    taps = {}
    net._debug_taps = taps
    with torch.no_grad():
        y = net(x, e)
    torch.cuda.synchronize()
    taps["y"] = y.clone()
    runs.append({k: v.cpu() for k, v in taps.items()})
net._debug_taps = None
otaps = {}
if B * N <= 40000:
    yo = O.forward(cfg, sd, d["mixture"], d["embedding_gt"], taps=otaps, fast_lstm=True)
    otaps["y"] = yo
first = None
for k in runs[0]:
    d1 = (runs[0][k] - runs[1][k]).abs().max().item()
    d2 = (runs[1][k] - runs[2][k]).abs().max().item()
    eo = [(r[k] - otaps[k].reshape(r[k].shape)).abs().max().item() if k in otaps else float("nan") for r in runs]
    print(f"{k:16s} run0-run1 {d1:.3e}  run1-run2 {d2:.3e}   vs oracle: " + " ".join(f"{v:.2e}" for v in eo))
    if first is None and d1 > 0:
        first = k
if first:
    a, b = runs[0][first], runs[1][first]
    idx = (a != b).nonzero()
    print("first differing stage", first, "shape", tuple(a.shape), "n differing", idx.shape[0])
    print("max diff", (a - b).abs().max().item(), "run1 vs run2 differing", (runs[1][first] != runs[2][first]).sum().item())
    print("distinct dim0", idx[:, 0].unique().tolist()[:20])
    print("distinct dim1", idx[:, 1].unique().tolist()[:40])
    print("distinct last", idx[:, -1].unique().tolist()[:80])
    for i in idx[:8].tolist():
        print(i, a[tuple(i)].item(), b[tuple(i)].item(), otaps[first].reshape(a.shape)[tuple(i)].item() if first in otaps else None)
