"""GPU probe (test tooling): the same C-ABI calls of the small-batch path through two library builds, outputs compared.
    python scripts/ab_kernels.py lookoncetohear_amd/_lookonce_hip.so lookoncetohear_amd/_lookonce_hip_slp.so"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lookoncetohear_amd import _cabi, config  # noqa: E402
from lookoncetohear_amd.net import Net  # noqa: E402

dev = torch.device("cuda:0")
libs = [_cabi.Lib(os.path.abspath(p)) for p in sys.argv[1:3]]
torch.manual_seed(0)
net = Net(**config.TSH_PARAMS).eval()
net.load_state_dict(config.separator_weights(0), strict=True)
net = net.to(dev)
bp = net._weights(dev)["blocks"][0]
B, T = 2, 125
g = torch.Generator().manual_seed(5)
x = torch.randn(B, T, 97, 64, generator=g).to(dev)
h0 = (torch.randn(B * 97, 64, generator=g) * 0.3).to(dev)
c0 = (torch.randn(B * 97, 64, generator=g) * 0.3).to(dev)
P = lambda t: t.data_ptr()
res = []
for lib in libs:
    st = torch.cuda.current_stream(dev).cuda_stream
    out = {}
    hbuf = torch.zeros(B * T * 97, 128, device=dev)
    lib.call("lh_ln_lstm_intra", P(x), P(bp["intra_ln_w"]), P(bp["intra_ln_b"]), P(bp["intra_w16"]), P(bp["intra_b16"]), P(hbuf), B * T, 1, st)
    out["ln_lstm_intra(h3)"] = hbuf.clone()
    xb = torch.zeros_like(x)
    lib.call("lh_linear_res", P(hbuf), P(bp["intra_lin_w"]), P(bp["intra_lin_b"]), P(x), P(xb), B * T * 97, 128, st)
    out["linear_res"] = xb.clone()
    xc = torch.zeros_like(x)
    hN, cN = torch.zeros_like(h0), torch.zeros_like(c0)
    lib.call("lh_inter_matvec", P(x), P(bp["inter_s_wih"]), P(bp["inter_s_b"]), P(bp["inter_s_whh"]), P(bp["inter_lin_w"]),
             P(bp["inter_lin_b"]), P(h0), P(c0), P(hN), P(cN), P(xc), B, T, st)
    out["inter_matvec"] = xc.clone()
    hb2 = torch.zeros(B * T * 97, 128, device=dev)
    lib.call("lh_intra_stream", P(x), P(bp["intra_s_wih"]), P(bp["intra_s_b"]), P(bp["intra_s_whh"]), P(hb2), B * T, st)
    out["intra_stream"] = hb2.clone()
    xd = torch.zeros_like(x)
    lib.call("lh_inter_block", P(x), P(bp["inter_w8"]), P(bp["inter_b16"]), P(bp["inter_lin_wu"]), P(bp["inter_lin_b"]),
             P(h0), P(c0), P(hN), P(cN), P(xd), B, T, st)
    out["inter_block"] = xd.clone()
    torch.cuda.synchronize()
    res.append(out)
for k in res[0]:
    print(f"{k:22s} max |A - B| = {float((res[0][k] - res[1][k]).abs().max()):.3e}   (amp {float(res[0][k].abs().max()):.2f})")
print("inter_matvec vs inter_block within A:", float((res[0]["inter_matvec"] - res[0]["inter_block"]).abs().max()),
      " within B:", float((res[1]["inter_matvec"] - res[1]["inter_block"]).abs().max()))
print("intra_stream vs h3 within A:", float((res[0]["intra_stream"] - res[0]["ln_lstm_intra(h3)"]).abs().max()),
      " within B:", float((res[1]["intra_stream"] - res[1]["ln_lstm_intra(h3)"]).abs().max()))

# ---- fp64 reference of the inter path (LayerNorm over C -> LSTM over time with (h0, c0) -> Linear + residual) for the same x
sd = {k: v.double() for k, v in net.state_dict().items()}
pre = "tfgridnet.blocks.0."
xd64 = x.double()                                          # [B, T, F, C]
ln = torch.nn.functional.layer_norm(xd64, (64,), sd[pre + "inter_norm.norm.weight"], sd[pre + "inter_norm.norm.bias"], 1e-5)
seq = ln.permute(0, 2, 1, 3).reshape(B * 97, T, 64)        # sequences (b, f) over time
lstm = torch.nn.LSTM(64, 64, batch_first=True).double().to(dev)
with torch.no_grad():
    lstm.weight_ih_l0.copy_(sd[pre + "inter_rnn.weight_ih_l0"]); lstm.weight_hh_l0.copy_(sd[pre + "inter_rnn.weight_hh_l0"])
    lstm.bias_ih_l0.copy_(sd[pre + "inter_rnn.bias_ih_l0"]); lstm.bias_hh_l0.copy_(sd[pre + "inter_rnn.bias_hh_l0"])
    hseq, _ = lstm(seq, (h0.double()[None], c0.double()[None]))
    proj = hseq @ sd[pre + "inter_linear.weight"].t() + sd[pre + "inter_linear.bias"]
    ref = xd64 + proj.reshape(B, 97, T, 64).permute(0, 2, 1, 3)
for i, name in enumerate(sys.argv[1:3]):
    for k in ("inter_matvec", "inter_block"):
        print(f"{os.path.basename(name):28s} {k:14s} max |out - fp64 reference| = {float((res[i][k].double() - ref).abs().max()):.3e}")
