#!/usr/bin/env python
"""Summarise a tools/profile.sh output directory into one small text file for profiles/.

    summarize_prof.py <prof dir> <summary.txt> [<workload> <kernel substring> <traffic.json>
                                                [<algorithmic bytes or flops per launch> <peak per second>]]

With the three optional arguments the HBM bytes per launch of the named kernel are written into
traffic.json under the workload's key: FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950
FETCH_SIZE counts a 128-byte read request as 64 bytes for wide coalesced reads (16 bytes per lane,
which is what the LDS-DMA and the stationary-row loads of the score kernels are), so it is doubled;
WRITE_SIZE was calibrated on the score matrix itself (it reports n1 * n2 * 4 bytes exactly).
With the last two arguments the summary also states the roofline fraction of that kernel recomputed
from the profiler's own average duration: algorithmic work per launch / average duration / peak --
the number bench.py's `roofline.frac` (from HIP events in the timed region) has to agree with."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return max(r, key=os.path.getmtime) if r else None  # the latest run, if the directory holds several


def main(d, out, traffic=None, roof=None):
    lines = []
    means = {}
    st = find(os.path.join(d, "trace"), "*kernel_stats.csv")
    if st:
        lines.append("== rocprofv3 --kernel-trace --stats (kernel_stats.csv) ==")
        with open(st) as f:
            rows = list(csv.DictReader(f))
        lines.append("%-70s %8s %14s %14s %8s" % ("Name", "Calls", "TotalNs", "AverageNs", "Pct"))
        for r in rows[:25]:
            lines.append("%-70s %8s %14s %14.1f %8s" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"],
                                                        float(r["AverageNs"]), r["Percentage"]))
    for tag in ["fetch", "write", "sq", "sqg"]:
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
                means[(k, c)] = (sum(v), len(v))
    if traffic:
        workload, sub, path = traffic
        # every kernel whose name holds the substring (template variants of one kernel): mean over all
        # their dispatches
        def mean_of(counter):
            hit = [v for (k, c), v in means.items() if sub in k and c == counter]
            return [sum(t for t, _ in hit) / sum(n for _, n in hit)] if hit else []

        fetch, write = mean_of("FETCH_SIZE"), mean_of("WRITE_SIZE")
        if fetch and write:
            try:
                with open(path) as f:
                    db = json.load(f)
            except (OSError, ValueError):
                db = {}
            db[workload] = {"kernel": sub, "fetch_bytes_corrected": fetch[0] * 1024 * 2,
                            "write_bytes": write[0] * 1024,
                            "hbm_bytes_per_launch": fetch[0] * 1024 * 2 + write[0] * 1024,
                            "source": os.path.basename(out)}
            with open(path, "w") as f:
                json.dump(db, f, indent=1, sort_keys=True)
            lines.append("")
            lines.append("HBM traffic per launch of %s: fetch %.3f GB (x2 corrected), write %.3f GB"
                         % (sub, fetch[0] * 2048 / 1e9, write[0] * 1024 / 1e9))
    if traffic and roof and st:
        work, peak = float(roof[0]), float(roof[1])
        hit = [r for r in rows if traffic[1] in r["Name"]]
        if hit:
            calls = sum(int(r["Calls"]) for r in hit)
            avg_s = sum(float(r["TotalDurationNs"]) for r in hit) / calls * 1e-9
            lines.append("")
            lines.append("roofline of %s (%d kernel variant(s)) from this trace: %.6g algorithmic units per launch / "
                         "%.4f ms average over %d launches = %.6g per s = %.4f of the peak %.6g"
                         % (traffic[1], len(hit), work, avg_s * 1e3, calls, work / avg_s, work / avg_s / peak, peak))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:6] if len(sys.argv) >= 6 else None,
         sys.argv[6:8] if len(sys.argv) >= 8 else None)
