"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per (counter, kernel, grid size).

usage: python tools/pmc_summary.py <dir-with-*_counter_collection.csv> [...]   (prints CSV to stdout)
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(list)
    for root in sys.argv[1:]:
        for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"].split("(")[0]
                    acc[(row["Counter_Name"], name, int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
    out = csv.writer(sys.stdout)
    out.writerow(["counter", "kernel", "grid_threads", "launches", "mean_value"])
    for (counter, name, grid), vals in sorted(acc.items()):
        out.writerow([counter, name, grid, len(vals), round(sum(vals) / len(vals), 1)])


if __name__ == "__main__":
    main()
