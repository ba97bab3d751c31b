"""Per-call-site kernel durations from a rocprofv3 --kernel-trace CSV.

`--stats` lumps every launch of one kernel symbol together; bench.py launches e.g. the gather kernel six times a step
with different shapes.  Launches on one HIP stream keep program order, so the j-th launch of a symbol within a step is
always the same call site: this groups by (symbol, j mod launches-per-step) and prints avg/min/max per call site.

usage: python tools/trace_summary.py <kernel_trace.csv> <launches_of_one_step> > summary.csv
       where <launches_of_one_step> is e.g. "gather_fwd_lds_kernel=6,ball_query_kernel=3" (others default to 1)
"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    per_step = {}
    if len(sys.argv) > 2:
        for item in sys.argv[2].split(","):
            k, v = item.split("=")
            per_step[k] = int(v)
    seq = defaultdict(int)
    acc = defaultdict(list)
    rows = list(csv.DictReader(open(path, newline="")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        name = r["Kernel_Name"]
        if "v3d::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "")
        n = 1
        for k, v in per_step.items():
            if k in short:
                n = v
        key = (short, r["Stream_Id"])
        slot = seq[key] % n
        seq[key] += 1
        acc[(short, slot, int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "call_site_in_step", "grid_threads", "launches", "avg_ns", "min_ns", "max_ns"])
    for (short, slot, grid), v in sorted(acc.items()):
        out.writerow([short, slot, grid, len(v), round(sum(v) / len(v), 1), min(v), max(v)])


if __name__ == "__main__":
    main()
