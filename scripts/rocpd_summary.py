"""Dump a rocprofv3 (rocpd sqlite) result database to the small text summaries kept under profiles/.

    python scripts/rocpd_summary.py <results.db> <out.csv>          # kernel stats (top_kernels view)
    python scripts/rocpd_summary.py <results.db> <out.csv> --pmc    # per-kernel average of each collected counter
    python scripts/rocpd_summary.py <results.db> <out.txt> --timeline [marker]   # every dispatch of the LAST step (between the
                                              # last two dispatches whose name contains `marker`, default k_stft_conv_in):
                                              # name, duration, gap to the previous dispatch's end — what a step consists of
"""
import csv
import sqlite3
import sys


def short(name):
    if name.startswith("_ZN2lh12k_ln_lstm_h3ILi"):
        return "lh::k_ln_lstm_h3<%s>" % name[len("_ZN2lh12k_ln_lstm_h3ILi")]
    return name.split("(")[0].replace("void ", "")[:100]


def timeline(db, out, marker):
    c = sqlite3.connect(db)
    objs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    for t in sorted(objs, key=lambda n: (not n.startswith("kernels"), n)):
        try:
            tc = [r[1] for r in c.execute(f"pragma table_info('{t}')")]
        except Exception:
            continue
        ncol = next((x for x in ("name", "kernel_name", "kernel") if x in tc), None)
        if not (ncol and "start" in tc and "end" in tc):
            continue
        rows = c.execute(f"select {ncol}, start, end from '{t}' order by start").fetchall()
        marks = [i for i, r in enumerate(rows) if isinstance(r[0], str) and marker in r[0]]
        if len(marks) < 2:
            continue
        a, b = marks[-2], marks[-1]
        with open(out, "w") as f:
            f.write(f"# dispatches {a}..{b - 1} of {t}: one step, from `{marker}` to the next one (us)\n")
            f.write(f"# {'kernel':<60} {'start_us':>9} {'dur_us':>9} {'gap_us':>8}   (start: from the step's first dispatch; dispatches of "
                    f"concurrent streams overlap: gap < 0)\n")
            tot = gaps = small = 0.0
            for i in range(a, b):
                n, st, en = rows[i]
                gap = (st - rows[i - 1][2]) / 1e3 if i > a else 0.0
                d = (en - st) / 1e3
                tot += d
                gaps += max(gap, 0.0)
                if d < 100.0:
                    small += d
                f.write(f"{short(n)[:60]:<62} {(st - rows[a][1]) / 1e3:9.1f} {d:9.1f} {gap:8.1f}\n")
            f.write(f"# span {(rows[b][1] - rows[a][1]) / 1e3:.1f} us = kernels {tot:.1f} + gaps {gaps:.1f}; "
                    f"dispatches shorter than 100 us: {small:.1f} us\n")
        print("wrote", out, b - a, "dispatches")
        return
    print("no table with a name / start / end and two markers; objects:", " ".join(objs))


def main(db, out, pmc=False):
    c = sqlite3.connect(db)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        if not pmc:
            rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
            w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
            for name, calls, tot, avg, pct in rows:
                w.writerow([short(name), calls, f"{tot / 1e3:.1f}" if tot > 1e6 else f"{tot:.1f}", f"{avg:.1f}", f"{pct:.2f}"])
            print("wrote", out, len(rows), "kernels")
            # the same kernel serves several stages (e.g. k_ln_lstm_lin: intra and inter): split by launch grid.
            # rocpd schemas differ between versions: discover a table/view with a kernel name, a grid size and times.
            w.writerow([])
            done = False
            objs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
            for t in sorted(objs, key=lambda n: (not n.startswith("kernels"), n)):
                try:
                    tc = [r[1] for r in c.execute(f"pragma table_info('{t}')")]
                except Exception:
                    continue
                ncol = next((x for x in ("name", "kernel_name", "kernel") if x in tc), None)
                if "grid_size" in tc:
                    gexpr = "grid_size"
                elif all(x in tc for x in ("grid_size_x", "grid_size_y", "grid_size_z")):
                    gexpr = "grid_size_x * grid_size_y * grid_size_z"
                elif all(x in tc for x in ("grid_x", "grid_y", "grid_z")):
                    gexpr = "grid_x * grid_y * grid_z"
                else:
                    gexpr = None
                if not (ncol and gexpr and "start" in tc and "end" in tc):
                    continue
                try:
                    rows = c.execute(f"select {ncol}, {gexpr}, count(*), avg(end - start), sum(end - start) from '{t}' "
                                     f"group by 1, 2 order by 5 desc").fetchall()
                except Exception:
                    continue
                w.writerow([f"# per (kernel, grid size) from {t}"])
                w.writerow(["kernel", "grid_size", "calls", "avg_us", "total_us"])
                for name, g, n, avg, tot in rows:
                    if isinstance(name, str) and ("lh::" in name or "_ZN2lh" in name):
                        w.writerow([short(name), g, n, f"{avg / 1e3:.1f}", f"{tot / 1e3:.1f}"])
                done = True
                break
            if not done:
                w.writerow(["# per-grid split unavailable; tables/views: " + " ".join(objs)])
            return
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        w.writerow(["# counters_collection columns: " + " ".join(cols)])
        kcol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
        ccol = "counter_name" if "counter_name" in cols else None
        vcol = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
        if not (kcol and ccol and vcol):
            print("unknown schema", cols)
            return
        did = "dispatch_id" if "dispatch_id" in cols else None
        g = "grid_size" if "grid_size" in cols else "0"
        q = (f"select {kcol}, {g}, {ccol}, count(*), sum({vcol}) from counters_collection group by {kcol}, {g}, {ccol}")
        if did:  # a counter may have several rows per dispatch (per XCC/instance): sum within a dispatch first
            q = (f"select k, g, cn, count(*), avg(v) from (select {kcol} as k, {g} as g, {ccol} as cn, {did} as d, "
                 f"sum({vcol}) as v from counters_collection group by {kcol}, {g}, {ccol}, {did}) group by k, g, cn")
        w.writerow(["kernel", "grid_size", "counter", "dispatches", "avg_per_dispatch"])
        n = 0
        for k, gs, cn, cnt, v in c.execute(q):
            if "lh::" in k or "_ZN2lh" in k:
                w.writerow([short(k), gs, cn, cnt, f"{v:.6g}"])
                n += 1
        print("wrote", out, n, "rows")


if __name__ == "__main__":
    if "--timeline" in sys.argv:
        i = sys.argv.index("--timeline")
        timeline(sys.argv[1], sys.argv[2], sys.argv[i + 1] if len(sys.argv) > i + 1 else "k_stft_conv_in")
    else:
        main(sys.argv[1], sys.argv[2], "--pmc" in sys.argv)
