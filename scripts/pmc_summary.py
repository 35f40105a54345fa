#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per kernel (one row per kernel, one column per counter)."""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")).replace(", ", ".") if m else name[:40]


def main(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(root + "/pmc*/*counter_collection.csv")):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if not k.startswith("k_"):
                    continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    counters = sorted({c for k in acc for c in acc[k]})
    print("kernel," + ",".join(counters))
    for k in sorted(acc):
        vals = []
        for c in counters:
            v = acc[k].get(c)
            vals.append("%.4g" % (sum(v) / len(v)) if v else "")
        print(k + "," + ",".join(vals))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof")
