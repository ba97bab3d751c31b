"""Per-call-site kernel durations from the kernel trace (tools/rocpd_export.py trace): `--stats` lumps every launch of a
symbol together, while bench.py launches e.g. the group-gather kernel at six shapes per step; the launch geometry
(grid size) tells the call sites apart.

usage: python tools/callsite_stats.py <kernel_trace.csv> > callsite_stats.csv"""
import csv
import sys
from collections import defaultdict


def main():
    acc = defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1], newline="")):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "v3d::" not in name:
            continue
        acc[(name, int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "grid_x_threads", "launches", "avg_ns", "min_ns", "max_ns"])
    for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        out.writerow([name, grid, len(v), round(sum(v) / len(v), 1), min(v), max(v)])


if __name__ == "__main__":
    main()
