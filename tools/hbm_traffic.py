#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh into the
JSON bench.py reads for `roofline.traffic` (HBM bytes per launch of the dominant
kernel, keyed by the FULL kernel name, which ends in the hash of the generated
source: counters of one code object are never applied to another).  gfx950 correction per
MI355X_MICROARCH.md (HBM / rocprofv3 section): counters are in KiB and
FETCH_SIZE reports half of a wide coalesced read stream -> x2.
usage: hbm_traffic.py <round-tag> <out.json>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def counter_means(root, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("sf_"):
                    acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    tag, out_path = sys.argv[1], sys.argv[2]
    result = {}
    if os.path.exists(out_path):  # keep the entries of workloads not profiled this time
        try:
            with open(out_path) as f:
                result = json.load(f)
        except (OSError, ValueError):
            result = {}
    for wl in ("c3", "c2", "c5", "box", "wide", "generic"):
        root = "gpurun_out/prof_%s_%s" % (tag, wl)
        if not os.path.isdir(root):
            continue
        fetch, write = counter_means(root, "FETCH_SIZE"), counter_means(root, "WRITE_SIZE")
        for kernel in fetch:
            if kernel not in write:
                continue
            family = re.sub(r"_[0-9a-f]{8}$", "", kernel)
            # one record per kernel family: a new code object replaces the old record
            for old in [k for k in result if re.sub(r"_[0-9a-f]{8}$", "", k) == family]:
                del result[old]
            result[kernel] = {
                "kernel": kernel,
                "hbm_bytes_per_launch": fetch[kernel] * 1024.0 * 2.0 + write[kernel] * 1024.0,
                "fetch_size_kb_reported": fetch[kernel],
                "write_size_kb_reported": write[kernel],
                "correction": "gfx950: FETCH_SIZE reports half of a wide coalesced read stream -> x2 "
                              "(MI355X_MICROARCH.md, HBM)",
                "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- "
                           "python3 bench.py <%s workload of tools/profile_round.sh> --steps 1 --warmup 0 "
                           "--no-cpu-baseline" % wl,
                "round": tag,
            }
    with open(out_path, "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
