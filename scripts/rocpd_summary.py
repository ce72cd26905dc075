"""Dump a rocprofv3 (rocpd sqlite) result database to the small text summaries kept under profiles/.

    python scripts/rocpd_summary.py gpurun_out/prof_stats/r1_results.db profiles/r01_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for name, calls, tot, avg, pct in rows:
            short = name.split("(")[0].replace("void ", "")
            w.writerow([short[:120], calls, f"{tot:.1f}", f"{avg:.1f}", f"{pct:.2f}"])
    try:
        pm = c.execute("select k.name, p.pmc_name if 0 else '' from kernels k limit 0").fetchall()
    except Exception:
        pass
    print("wrote", out, len(rows), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
