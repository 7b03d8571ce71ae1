#!/usr/bin/env python
"""Summarise a tools/profile.sh output directory into one small text file for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def main(d, out):
    lines = []
    st = find(os.path.join(d, "trace"), "*kernel_stats.csv")
    if st:
        lines.append("== rocprofv3 --kernel-trace --stats (kernel_stats.csv) ==")
        with open(st) as f:
            rows = list(csv.DictReader(f))
        lines.append("%-70s %8s %14s %14s %8s" % ("Name", "Calls", "TotalNs", "AverageNs", "Pct"))
        for r in rows[:25]:
            lines.append("%-70s %8s %14s %14.1f %8s" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"],
                                                        float(r["AverageNs"]), r["Percentage"]))
    for tag in ["fetch", "write", "sq"]:
        cc = find(os.path.join(d, tag), "*counter_collection.csv")
        if not cc:
            continue
        acc = defaultdict(lambda: defaultdict(list))
        with open(cc) as f:
            for r in csv.DictReader(f):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        lines.append("")
        lines.append("== rocprofv3 --pmc pass '%s' (per-dispatch mean) ==" % tag)
        for k, cs in acc.items():
            for c, v in cs.items():
                lines.append("%-70s %-28s n=%-4d mean=%.6g" % (k[:70], c, len(v), sum(v) / len(v)))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
