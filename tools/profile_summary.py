#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output written by tools/profile.sh: per kernel the
trace statistics (calls, average duration) and the mean of every PMC counter
per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    out = []
    for path in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)):
        with open(path) as f:
            rows = list(csv.DictReader(f))
        out.append("== kernel stats (%s)" % os.path.relpath(path, root))
        for r in rows[:12]:
            out.append("  %-48s calls=%s avg_ns=%s total_ns=%s pct=%s" % (
                r.get("Name", "?")[:48], r.get("Calls"), r.get("AverageNs"),
                r.get("TotalDurationNs"), r.get("Percentage")))
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        acc = defaultdict(lambda: defaultdict(list))
        for path in files:
            with open(path) as f:
                for r in csv.DictReader(f):
                    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out.append("== counters (%s)" % os.path.basename(d))
        for k, cs in acc.items():
            if not k.startswith("sf_"):
                continue
            out.append("  " + k[:60])
            for c, vals in sorted(cs.items()):
                out.append("    %-28s mean/dispatch=%.6g  n=%d" % (c, sum(vals) / len(vals), len(vals)))
    text = "\n".join(out)
    print(text)
    with open(os.path.join(root, "summary.txt"), "w") as f:
        f.write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1])
