"""profiles/pmc_traffic.json from the two PMC passes (FETCH_SIZE, WRITE_SIZE) summarised by scripts/rocpd_summary.py.

    python scripts/make_pmc_traffic.py <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv> <batch> <out.json> [<pmc_sq.csv> <pmc_sq2.csv>]

With the two SQ passes (scripts/gpu.sh profile) every kernel also gets `mfma_busy` = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 XCCs x 1024 SIMDs), `valu_busy` = 4 x SQ_ACTIVE_INST_VALU / the same (the SQ_ACTIVE_* counters tick in
quad-cycles) and the raw counters; LOOKONCE_COMMIT in the environment stamps the file with the commit it was measured on.

Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950
FETCH_SIZE tallies 128-byte requests as 64 bytes, so it is doubled; WRITE_SIZE is used as is.  Values are averages per
kernel launch (dispatch).  The same kernel function serves several C-ABI calls; launches are told apart by grid size.
"""
import csv
import json
import sys

T, F = 625, 97


def load(path):
    out = {}
    for row in csv.reader(open(path)):
        if len(row) == 5 and row[0] != "kernel" and not row[0].startswith("#"):
            out[(row[0], int(row[1]))] = (int(row[3]), float(row[4]))
    return out


def load_counters(path):
    """kernel, grid, counter, dispatches, avg -> {(kernel, grid): {counter: avg per dispatch}}"""
    out = {}
    for row in csv.reader(open(path)):
        if len(row) == 5 and row[0] != "kernel" and not row[0].startswith("#"):
            out.setdefault((row[0], int(row[1])), {})[row[2]] = float(row[4])
    return out


def kernel_times(stats_csv):
    """first table of rocpd_summary's kernel stats (rocprofv3 --kernel-trace --stats): kernel -> (calls, avg_us)"""
    out = {}
    for row in csv.reader(open(stats_csv)):
        if not row:
            break
        if len(row) == 5 and row[0] != "kernel" and not row[0].startswith("#"):
            try:
                out[row[0]] = (int(row[1]), float(row[3]))
            except ValueError:
                break
    return out


def main(fetch_csv, write_csv, batch, out_json, sq_csv=None, sq2_csv=None, stats_csv=None):
    import os
    B = int(batch)
    tm = kernel_times(stats_csv) if stats_csv and os.path.exists(stats_csv) else {}
    fe, wr = load(fetch_csv), load(write_csv)
    sq = {}
    for path in (sq_csv, sq2_csv):
        if path and os.path.exists(path):
            for k, v in load_counters(path).items():
                sq.setdefault(k, {}).update(v)
    A = 15.52e6 * B
    qk = 5.82e6 * B
    wg = lambda nseq: ((nseq + 15) // 16) * 256          # LSTM launches: 16 sequences per 256-thread workgroup
    # C-ABI call -> (kernel-name substring, grid size or None, algorithmic bytes per kernel launch)
    table = {
        "lh_intra_block": (("k_intra_xp", "k_ln_lstm_lin"), None, 2.5 * A),
        "lh_inter_block": (("k_lstm_lin8", "k_inter_xp"), None, 2.0 * A),
        "lh_qkv_proj_ln": ("k_qkv_proj_ln", None, 2 * A + 2 * qk),
        "lh_local_attn": ("k_local_attn", None, 2 * qk + 2 * A),
        "lh_proj_ln_res": ("k_proj_ln_res", None, 3 * A),
        "lh_stft_conv_in": ("k_stft_conv_in", None, 16.16e6 * B),
        "lh_deconv_istft": ("k_deconv_istft", None, 16.16e6 * B),
    }
    kernels = {}
    for call, (pat, grid, alg) in table.items():
        pats = (pat,) if isinstance(pat, str) else pat
        hit = lambda k: any(q in k[0] for q in pats) and (grid is None or k[1] == grid)
        f = [(k, v) for k, v in fe.items() if hit(k)]
        w = [(k, v) for k, v in wr.items() if hit(k)]
        if not f or not w:
            continue
        fn = sum(v[0] for _, v in f); wn = sum(v[0] for _, v in w)
        fetch_kib = sum(v[0] * v[1] for _, v in f) / fn
        write_kib = sum(v[0] * v[1] for _, v in w) / wn
        kernels[call] = {
            "kernel": f[0][0][0], "grid_size": f[0][0][1], "dispatches": fn,
            "fetch_bytes_per_launch": 2 * 1024 * fetch_kib, "write_bytes_per_launch": 1024 * write_kib,
            "hbm_bytes_per_launch": 2 * 1024 * fetch_kib + 1024 * write_kib,
            "algorithmic_bytes_per_launch": alg,
        }
        kernels[call]["ratio_to_algorithmic"] = kernels[call]["hbm_bytes_per_launch"] / alg
        if call == "lh_intra_block":
            # this kernel's "algorithmic" figure is what ITS two-launch structure must move (forward: read x,
            # write out; reverse: read x, read + rewrite out = 5 A per call); SURVEY.md §8(d) budgets the intra STAGE at 2 A
            # (read A, write A).  Both ratios are stated; 2 A is not reachable on this chip.
            kernels[call]["survey_8d_bytes_per_call"] = 2.0 * A
            kernels[call]["ratio_to_survey_8d_per_call"] = 2 * kernels[call]["hbm_bytes_per_launch"] / (2.0 * A)
        if f[0][0][0] in tm:        # the same launch as rocprofv3's tracer times it (every dispatch separated: longer than back to back)
            kernels[call]["rocprof_avg_launch_us"] = tm[f[0][0][0]][1]
        c = sq.get(f[0][0])
        if c and c.get("GRBM_GUI_ACTIVE"):
            # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCCs (6.4e6 for a 0.41 ms launch at ~1.9 GHz); 1024 SIMDs per chip
            simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                kernels[call]["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
            if "SQ_ACTIVE_INST_VALU" in c:
                kernels[call]["valu_busy"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles
            kernels[call]["sq_counters_per_launch"] = c
    json.dump({
        "source": f"{fetch_csv} + {write_csv} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                  "summarised by scripts/rocpd_summary.py --pmc)",
        "batch_per_gpu": B, "commit": os.environ.get("LOOKONCE_COMMIT", "unknown"),
        "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM "
                       "section); WRITE_SIZE uncorrected",
        "kernels": kernels}, open(out_json, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:18s} {v['hbm_bytes_per_launch'] / 1e9:8.3f} GB measured  {v['algorithmic_bytes_per_launch'] / 1e9:8.3f} GB algorithmic"
              f"  x{v['ratio_to_algorithmic']:.2f}")


if __name__ == "__main__":
    main(*sys.argv[1:8])
