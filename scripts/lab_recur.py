"""GPU lab for the recurrent kernels: times `lh_intra_block` / `lh_inter_block` in isolation (HIP events, B x T x 97 x 64
random activations, the random-init weights of block 0) under a list of `lh_set_tuning` settings and compares every
setting's outputs with the first one's.  One process per library build:

    LOOKONCE_HIP_LIB=lookoncetohear_amd/_lookonce_hip_x.so python scripts/lab_recur.py --tunes "_ 5=2 9=0 8=0"

("_" = defaults.)  Prints one line per setting; used through `scripts/gpu.sh lab`.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lookoncetohear_amd import _cabi, config  # noqa: E402
from lookoncetohear_amd.net import Net  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=625)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tunes", default="_")
    ap.add_argument("--which", default="intra,inter")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _cabi.load()
    torch.manual_seed(0)
    net = Net(**config.TSH_PARAMS).eval().to(dev)
    pk = net._weights(dev)
    bp = pk["blocks"][0]
    B, T = args.batch, args.frames
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(B, T, 97, 64, generator=g).to(dev)
    h0 = (torch.randn(B * 97, 64, generator=g) * 0.3).to(dev)
    c0 = (torch.randn(B * 97, 64, generator=g) * 0.3).to(dev)
    P = lambda t: t.data_ptr()
    st = torch.cuda.current_stream(dev).cuda_stream

    def run_intra(out):
        lib.call("lh_intra_block", P(x), P(bp["intra_w16"]), P(bp["intra_b16"]), P(bp["intra_lin_w2"]),
                 P(bp["intra_lin_b"]), P(out), B * T, st)

    def run_inter(out, hN, cN):
        lib.call("lh_inter_block", P(x), P(bp["inter_w8"]), P(bp["inter_b16"]), P(bp["inter_lin_wu"]),
                 P(bp["inter_lin_b"]), P(h0), P(c0), P(hN), P(cN), P(out), B, T, st)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps

    ref = {}
    name = os.path.basename(lib.path)
    which = args.which.split(",")
    for tune in args.tunes.split():
        keys = []
        if tune != "_":
            for kv in tune.split(","):
                k, v = kv.split("=")
                lib.call("lh_set_tuning", int(k), int(v))
                keys.append(int(k))
        line = f"{name:28s} tune {tune:14s}"
        if "intra" in which:
            o1, o2 = torch.zeros_like(x), torch.zeros_like(x)
            ms = timed(lambda: run_intra(o1))
            run_intra(o2)
            torch.cuda.synchronize()
            ref.setdefault("intra", o1.clone())
            line += f"  intra {ms:7.4f} ms  dmax {(o1 - ref['intra']).abs().max().item():.3e}  rerun {'same' if torch.equal(o1, o2) else 'DIFFERS'}"
        if "inter" in which:
            o1, o2 = torch.zeros_like(x), torch.zeros_like(x)
            hN, cN = torch.zeros_like(h0), torch.zeros_like(c0)
            ms = timed(lambda: run_inter(o1, hN, cN))
            run_inter(o2, hN, cN)
            torch.cuda.synchronize()
            ref.setdefault("inter", (o1.clone(), hN.clone(), cN.clone()))
            r = ref["inter"]
            line += (f"  inter {ms:7.4f} ms  dmax {(o1 - r[0]).abs().max().item():.3e} h {(hN - r[1]).abs().max().item():.1e}"
                     f" c {(cN - r[2]).abs().max().item():.1e}  rerun {'same' if torch.equal(o1, o2) else 'DIFFERS'}")
        print(line, flush=True)
        for k in keys:
            lib.call("lh_set_tuning", k, {3: 1, 8: 1, 9: 3}.get(k, 0))       # back to the defaults


if __name__ == "__main__":
    main()
