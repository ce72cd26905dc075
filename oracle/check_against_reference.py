"""TEST INFRASTRUCTURE. Pins `oracle/tfgridnet_oracle.py` against the *unmodified* reference model,
imported from /root/reference under `oracle/ref_stubs.py` (only possible in the build container).

    python -m oracle.check_against_reference

Checks (fp64 for tight agreement, fp32 for the working precision):
  * offline forward, zero state, several lengths (mod-pad path included)
  * forward from NON-ZERO random state (all rings / LSTM state / conv tails populated) + returned state
  * streaming (T=1 chunks, the `_causal_unfold_chunk` early-return branch) == offline
  * stft filterbank restatement == stub filterbank buffer; param manifest == reference state_dict keys/shapes
"""
import sys
import torch

from oracle import ref_stubs
from oracle import tfgridnet_oracle as O


def rand_state(cfg, B, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    st = O.init_state(cfg, B, dtype)

    def fill(t):
        t.copy_(torch.randn(t.shape, generator=g, dtype=torch.float64).to(dtype) * 0.5)
    fill(st["conv_buf"]); fill(st["deconv_buf"]); fill(st["istft_buf"])
    for b in st["gridnet_bufs"].values():
        for t in b.values():
            fill(t)
    return st


def clone_state(st):
    return {k: (clone_state(v) if isinstance(v, dict) else v.clone()) for k, v in st.items()}


def flat_state(st, pre=""):
    out = {}
    for k, v in st.items():
        if isinstance(v, dict):
            out.update(flat_state(v, pre + k + "."))
        else:
            out[pre + k] = v
    return out


def main():
    torch.set_num_threads(8)
    Net = ref_stubs.reference_net_class()
    cfg = O.Cfg(**O.TSH_PARAMS)
    ref = Net(**O.TSH_PARAMS).eval()
    rsd = ref.state_dict()
    man = O.param_manifest(cfg)
    assert set(man) == set(rsd), (set(man) ^ set(rsd))
    for k, shp in man.items():
        assert tuple(rsd[k].shape) == tuple(shp), (k, rsd[k].shape, shp)
    print("manifest: %d tensors, %d params OK" % (len(man), sum(p.numel() for p in ref.parameters())))
    fb = rsd["tfgridnet.enc.filterbank._filters"][:, 0]
    d = (fb - O.stft_filters(cfg.nfft, cfg.hop)).abs().max().item()
    print("filterbank restatement vs stub: %.3e" % d)
    assert d < 1e-7
    sd = O.synthetic_state_dict(cfg, seed=0)
    ref.load_state_dict(sd, strict=True)
    worst = 0.0
    for dtype, tol in ((torch.float64, 1e-10), (torch.float32, 2e-4)):
        refd = ref.double() if dtype == torch.float64 else ref.float()
        sdd = {k: v.to(dtype) for k, v in sd.items()}
        g = torch.Generator().manual_seed(7)
        for (B, N) in ((2, 8000), (1, 8100), (1, 1000)):
            x = (torch.randn(B, 2, N, generator=g, dtype=torch.float64) * 0.05).to(dtype)
            e = torch.randn(B, 1, 256, generator=g, dtype=torch.float64).abs()
            e = (e / e.norm(dim=-1, keepdim=True)).to(dtype)
            with torch.no_grad():
                yr = refd(x, e, input_state=O.init_state(cfg, B, dtype))
                yo = O.forward(cfg, sdd, x, e, dtype=dtype)
                yo2 = O.forward(cfg, sdd, x, e, dtype=dtype, fast_lstm=True)
            err = (yr - yo).abs().max().item(); err2 = (yr - yo2).abs().max().item()
            print(f"{dtype} offline B={B} N={N}: max|ref-oracle|={err:.3e} fast_lstm={err2:.3e} (amp {yr.abs().max():.3f})")
            assert yr.shape == yo.shape and err < tol and err2 < tol
            worst = max(worst, err)
        # non-zero state, state in/out
        B, N = 2, 128 * 12 + 64
        x = (torch.randn(B, 2, N, generator=g, dtype=torch.float64) * 0.05).to(dtype)
        e = torch.randn(B, 256, generator=g, dtype=torch.float64).abs().to(dtype)
        st0 = rand_state(cfg, B, dtype, 3)
        with torch.no_grad():
            yr, sr = refd.predict(x, e, clone_state(st0), pad=False)
            yo, so = O.predict(cfg, sdd, x, e, clone_state(st0), pad=False, dtype=dtype)
        err = (yr - yo).abs().max().item()
        fr, fo = flat_state(sr), flat_state(so)
        serr = max((fr[k] - fo[k]).abs().max().item() for k in fr)
        assert all(fr[k].shape == fo[k].shape for k in fr)
        print(f"{dtype} non-zero state: out {err:.3e} state {serr:.3e}")
        assert err < tol and serr < tol
        # streaming == offline (T=1 chunks)
        nchunk = 60
        xs = (torch.randn(1, 2, 128 * nchunk + 64, generator=g, dtype=torch.float64) * 0.05).to(dtype)
        e1 = e[:1]
        with torch.no_grad():
            y_off, _ = O.predict(cfg, sdd, xs, e1, None, pad=False, dtype=dtype)
            st_r = O.init_state(cfg, 1, dtype)
            st_o = None
            outs_r, outs_o = [], []
            for i in range(nchunk):
                ch = xs[:, :, i * 128: i * 128 + 192]
                yr, st_r = refd.predict(ch, e1, st_r, pad=False)
                yo, st_o = O.predict(cfg, sdd, ch, e1, st_o, pad=False, dtype=dtype)
                outs_r.append(yr); outs_o.append(yo)
            ys_r, ys_o = torch.cat(outs_r, -1), torch.cat(outs_o, -1)
        print(f"{dtype} streaming: ref-stream vs oracle-stream {(ys_r - ys_o).abs().max():.3e}; "
              f"oracle-stream vs oracle-offline {(ys_o - y_off).abs().max():.3e}")
        assert (ys_r - ys_o).abs().max() < tol and (ys_o - y_off).abs().max() < max(tol, 1e-9)
    print("ORACLE PINNED AGAINST REFERENCE: OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
