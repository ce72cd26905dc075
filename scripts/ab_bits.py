"""Bit-identity A/B of two builds of the library on one box: every output of a fixed set of forwards (separator offline at
B = 8 and B = 1, 40 streaming chunks, the embedder) must be equal bit for bit.

    python scripts/ab_bits.py lookoncetohear_amd/_lookonce_hip.so lookoncetohear_amd/_lookonce_hip_x.so

Each library runs in its own process (LOOKONCE_HIP_LIB is read once per process).  Used when a change is supposed to be
arithmetic-neutral (round 5: the split on v_cvt_pk_f16_f32 + v_fma_mix_f32 instead of the compiler's four instructions)."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import hashlib, sys, torch
sys.path.insert(0, %r)
from lookoncetohear_amd import config, synth
from lookoncetohear_amd.net import Net
from lookoncetohear_amd.embed_net import EmbedTFGridNet
dev = "cuda:0"
net = Net(**config.TSH_PARAMS).eval(); net.load_state_dict(config.separator_weights(0), strict=True); net = net.to(dev)
enet = EmbedTFGridNet(**config.EMBED_PARAMS).eval(); enet.load_state_dict(config.embedder_weights(0), strict=True); enet = enet.to(dev)
h = lambda t: hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]
out = {}
with torch.no_grad():
    d = synth.batch(list(range(16)), 80000)
    out["sep_b16_5s"] = h(net(d["mixture"].to(dev), d["embedding_gt"].to(dev)))
    d = synth.batch([20], 80000)
    out["sep_b1_5s"] = h(net(d["mixture"].to(dev), d["embedding_gt"].to(dev)))
    d = synth.batch([21, 22], 128 * 40 + 64)
    st = net.make_streamer(2, dev, use_graph=True); st.set_embedding(d["embedding_gt"].to(dev))
    mix = d["mixture"].to(dev)
    out["stream_b2_40"] = h(torch.cat([st.step(mix[:, :, i * 128:i * 128 + 192]).clone() for i in range(40)], -1))
    x = synth.batch(list(range(30, 34)), 32000)["mixture"].to(dev)
    out["embed_b4_2s"] = h(enet(x))
    for mode in ("f32rec",):
        net.gemm_mode = mode
        d = synth.batch([0, 1], 16000)
        out["sep_" + mode] = h(net(d["mixture"].to(dev), d["embedding_gt"].to(dev)))
torch.cuda.synchronize()
print("HASHES", out)
''' % ROOT


def run(lib):
    env = dict(os.environ, LOOKONCE_HIP_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    for line in r.stdout.splitlines():
        if line.startswith("HASHES "):
            return eval(line[len("HASHES "):])
    raise RuntimeError(lib + ": " + r.stderr[-800:])


if __name__ == "__main__":
    a, b = run(sys.argv[1]), run(sys.argv[2])
    same = True
    for k in a:
        ok = a[k] == b[k]
        same &= ok
        print(f"{k:16s} {a[k]} {b[k]} {'identical' if ok else 'DIFFERENT'}")
    print("BIT-IDENTICAL" if same else "OUTPUTS DIFFER")
    sys.exit(0 if same else 1)
