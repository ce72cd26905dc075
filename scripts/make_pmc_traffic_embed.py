"""`embed` section of profiles/pmc_traffic.json: HBM bytes and busy fractions per C-ABI call of the enrollment embedder
(BASELINE configs[4]) from the rocprofv3 passes of `bench.py --mode embed` (scripts/gpu.sh profile_embed).

    python scripts/make_pmc_traffic_embed.py <kernel_stats.csv> <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv> <batch> <n_forwards> \
        <pmc_traffic.json to update> [<pmc_sq.csv> <pmc_sq2.csv>]

A C-ABI call of the embedder launches several kernels; `hbm_bytes_per_call` = sum over its kernels of (bytes per launch x
launches per call), launches per call = dispatches / (n_forwards x calls per forward).  Units / corrections as in
scripts/make_pmc_traffic.py (KiB; FETCH_SIZE x2 on gfx950).  `bench.py` reads the section for `roofline.traffic` of
`--mode embed` and of `secondary.embed_b64`.
"""
import csv
import json
import os
import sys

from make_pmc_traffic import load, load_counters

# call -> (calls per forward, [kernel-name substrings])   (mangled names: k_emb_recILb0E = <false> = intra axis)
CALLS = {
    "lh_emb_frontend": (1, ["k_emb_std", "k_emb_stft_conv", "k_emb_gn"]),
    "lh_emb_axis.intra": (3, ["k_emb_recILb0E", "k_emb_convt2ILb0E"]),
    "lh_emb_axis.inter": (3, ["k_emb_recILb1E", "k_emb_convt2ILb1E"]),
    "lh_emb_attn_block": (3, ["k_emb_qkv", "k_emb_vt", "k_gemm_nt", "k_emb_softmax", "k_emb_proj"]),
    "lh_emb_head": (1, ["k_emb_head"]),
}
AE = 1251 * 65 * 64 * 4.0
QK = 4 * 1251 * 520 * 4.0
ALG = {"lh_emb_axis.intra": 6 * AE, "lh_emb_axis.inter": 6 * AE, "lh_emb_attn_block": 6 * AE + 4 * QK, "lh_emb_head": AE,
       "lh_emb_frontend": 2 * AE}        # per clip (bench.py EMBED_CALL_BYTES)


def kernel_times(stats_csv):
    """first table of rocpd_summary's kernel stats: kernel -> (calls, avg_us)"""
    out = {}
    for row in csv.reader(open(stats_csv)):
        if len(row) == 5 and row[0] != "kernel" and not row[0].startswith("#"):
            try:
                out[row[0]] = (int(row[1]), float(row[3]))
            except ValueError:
                break
        elif not row:
            break
    return out


def main(stats_csv, fetch_csv, write_csv, batch, n_forwards, out_json, sq_csv=None, sq2_csv=None):
    B, nf = int(batch), int(n_forwards)
    fe, wr, tm = load(fetch_csv), load(write_csv), kernel_times(stats_csv)
    sq = {}
    for path in (sq_csv, sq2_csv):
        if path and os.path.exists(path):
            for k, v in load_counters(path).items():
                sq.setdefault(k[0], {})
                for c, x in v.items():          # several grids of one kernel: keep the larger launch's counters
                    sq[k[0]][c] = max(sq[k[0]].get(c, 0.0), x)
    calls = {}
    for call, (per_fwd, pats) in CALLS.items():
        n_calls = nf * per_fwd
        kerns, tot_bytes, tot_ms = {}, 0.0, 0.0
        for (name, grid), (n, kib) in fe.items():
            if not any(p in name for p in pats):
                continue
            w = wr.get((name, grid))
            if w is None:
                continue
            b_launch = 2 * 1024 * kib + 1024 * w[1]
            per_call = n / n_calls
            e = kerns.setdefault(name, {"launches_per_call": 0.0, "hbm_bytes_per_call": 0.0})
            e["launches_per_call"] += per_call
            e["hbm_bytes_per_call"] += b_launch * per_call
            tot_bytes += b_launch * per_call
        for name, e in kerns.items():
            if name in tm:
                e["avg_us"] = tm[name][1]
                e["ms_per_call"] = tm[name][1] * tm[name][0] / n_calls / 1e3
                tot_ms += e["ms_per_call"]
            c = sq.get(name)
            if c and c.get("GRBM_GUI_ACTIVE"):
                simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
                if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                    e["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
                if "SQ_ACTIVE_INST_VALU" in c:
                    e["valu_busy"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles
                if c.get("SQ_LDS_IDX_ACTIVE"):
                    e["lds_conflict_share"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        if not kerns:
            continue
        dom = max(kerns, key=lambda k: kerns[k].get("ms_per_call", 0.0))
        entry = {"hbm_bytes_per_call": tot_bytes, "algorithmic_bytes_per_call": ALG[call] * B,
                 "ratio_to_algorithmic": tot_bytes / (ALG[call] * B), "profiled_ms_per_call": tot_ms,
                 "dominant_kernel": dom, "dominant_kernel_avg_ms": kerns[dom].get("avg_us", 0.0) / 1e3,
                 "dominant_kernel_hbm_bytes_per_launch": kerns[dom]["hbm_bytes_per_call"] / max(kerns[dom]["launches_per_call"], 1e-9),
                 "kernels": kerns}
        for k in ("mfma_busy", "valu_busy"):
            if k in kerns[dom]:
                entry[k] = kerns[dom][k]
        calls[call] = entry
        print(f"{call:20s} {tot_bytes / 1e9:8.3f} GB measured  {ALG[call] * B / 1e9:8.3f} GB algorithmic  x{entry['ratio_to_algorithmic']:.2f}"
              f"  {tot_ms:7.3f} ms  dominant {dom[:40]} {entry.get('mfma_busy', float('nan')):.2f} mfma busy")
    tj = json.load(open(out_json)) if os.path.exists(out_json) else {}
    tj["embed"] = {"source": f"{os.path.basename(stats_csv)} + {os.path.basename(fetch_csv)} + {os.path.basename(write_csv)}"
                             + (f" + {os.path.basename(sq_csv)} + {os.path.basename(sq2_csv)}" if sq_csv and sq2_csv else "")
                             + " (rocprofv3 passes of `bench.py --mode embed --steps 3 --warmup 1`, scripts/gpu.sh profile_embed)",
                   "batch_per_gpu": B, "forwards_profiled": nf, "commit": os.environ.get("LOOKONCE_COMMIT", "unknown"),
                   "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950), WRITE_SIZE uncorrected", "calls": calls}
    json.dump(tj, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main(*sys.argv[1:9])
