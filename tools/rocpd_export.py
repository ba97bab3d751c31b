"""Export the summaries we commit under profiles/ from rocprofv3's rocpd SQLite output (`*_results.db`).

usage: python tools/rocpd_export.py stats   <results.db>   > kernel_stats.csv     (what `--stats` prints, per symbol)
       python tools/rocpd_export.py trace   <results.db>   > kernel_trace.csv     (input of tools/trace_summary.py)
       python tools/rocpd_export.py counters <results.db>.. > pmc_summary.csv     (mean per counter, symbol, grid size)
"""
import csv
import sqlite3
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    mode, paths = sys.argv[1], sys.argv[2:]
    out = csv.writer(sys.stdout)
    if mode == "stats":
        cur = sqlite3.connect(paths[0]).cursor()
        rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                           "group by name order by sum(duration) desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for n, c, s, a, lo, hi in rows:
            out.writerow([short(n), c, s, round(a, 1), lo, hi, round(100.0 * s / total, 2)])
    elif mode == "trace":
        cur = sqlite3.connect(paths[0]).cursor()
        out.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Stream_Id", "Grid_Size_X"])
        for n, s, e, st, g in cur.execute("select name, start, end, stream_id, grid_x from kernels order by start"):
            out.writerow([n, s, e, st, g])
    elif mode == "counters":
        acc = defaultdict(list)
        for p in paths:
            cur = sqlite3.connect(p).cursor()
            for cn, kn, g, v in cur.execute("select counter_name, kernel_name, grid_size, value from counters_collection"):
                acc[(cn, short(kn), int(g))].append(float(v))
        out.writerow(["counter", "kernel", "grid_threads", "launches", "mean_value", "min_value", "max_value"])
        for (cn, kn, g), vals in sorted(acc.items()):
            out.writerow([cn, kn, g, len(vals), round(sum(vals) / len(vals), 1), round(min(vals), 1), round(max(vals), 1)])
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
