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


def resident_waves_per_simd(root):
    """kernel -> waves of it that fit one SIMD at a time, from the launch records of the PMC passes (work-group size,
    LDS per block, VGPRs per lane): min over the limits of registers (512 per lane and SIMD), LDS (160 KiB per unit),
    wave slots (8 per SIMD)."""
    out = {}
    for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"]
                if not k.startswith("sf_") or k in out:
                    continue
                wpb = max(1, int(r["Workgroup_Size"]) // 64)
                vgpr = max(8, (int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count") or 0) + 7) // 8 * 8)
                lds = int(r["LDS_Block_Size"])
                blocks = min(4 * (512 // vgpr) // wpb, (160 * 1024 // lds) if lds > 0 else 64, 32 // wpb)
                out[k] = max(1, blocks) * wpb / 4.0
    return out


def counter_sums(root, counter):
    """kernel -> (sum over dispatches, number of dispatches) of one PMC pass."""
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("sf_"):
                    acc[r["Kernel_Name"]][0] += float(r["Counter_Value"])
                    acc[r["Kernel_Name"]][1] += 1
    return acc


def slab_records(tag, result):
    """The slab kernels of bench.py --gpus N (tools/slab_traffic.py under the same PMC passes): their launches
    cover plane ranges of many lengths, so the record is the SUM of the counters over one process divided by the
    full-slab launches it made (planes launched / planes of the slab)."""
    for root in sorted(glob.glob("gpurun_out/prof_%s_slab*" % tag)):
        try:
            with open(os.path.join(root, "planes.json")) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        kernel = rec["kernel"]
        fetch, write = counter_sums(root, "FETCH_SIZE"), counter_sums(root, "WRITE_SIZE")
        if kernel not in fetch or kernel not in write:
            continue
        expected = rec["executions"] * rec["launches_per_execution"]
        if fetch[kernel][1] != expected or write[kernel][1] != expected:
            print("slab%d: %d / %d dispatches counted, %d expected -- record skipped" %
                  (rec["world"], fetch[kernel][1], write[kernel][1], expected), file=sys.stderr)
            continue
        full = rec["executions"] * rec["full_slab_launches_per_execution"]
        family = "slab%d" % rec["world"]
        for old in [k for k, v in result.items() if isinstance(v, dict) and v.get("family") == family]:
            del result[old]
        result[kernel] = {
            "kernel": kernel, "family": family,
            "hbm_bytes_per_launch": (fetch[kernel][0] * 1024.0 * 2.0 + write[kernel][0] * 1024.0) / full,
            "fetch_size_kb_summed": fetch[kernel][0], "write_size_kb_summed": write[kernel][0],
            "dispatches": expected, "full_slab_launches": full, "n_local": rec["n_local"], "halo": rec["halo"],
            "correction": "gfx950: FETCH_SIZE reports half of a wide coalesced read stream -> x2 "
                          "(MI355X_MICROARCH.md, HBM)",
            "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- "
                       "python3 tools/slab_traffic.py --world %d (rank 0's launches of bench.py --gpus %d, no "
                       "neighbour; per launch = per 512 planes written)" % (rec["world"], rec["world"]),
            "round": tag,
        }


def main():
    tag, out_path = sys.argv[1], sys.argv[2]
    result = {}
    if os.path.exists(out_path):  # keep the entries of workloads not profiled this time
        try:
            with open(out_path) as f:
                result = json.load(f)
        except (OSError, ValueError):
            result = {}
    for wl in ("c3", "c2", "c5", "box", "wide", "cross3", "dense", "fork", "generic"):
        root = "gpurun_out/prof_%s_%s" % (tag, wl)
        if not os.path.isdir(root):
            continue
        fetch, write = counter_means(root, "FETCH_SIZE"), counter_means(root, "WRITE_SIZE")
        for kernel in fetch:
            if kernel not in write:
                continue
            family = re.sub(r"_[0-9a-f]{8}$", "", kernel)
            # one record per kernel family AND workload: a new code object replaces the old record of the same
            # workload (the fork program runs star kernels of the family C3's kernel belongs to)
            for old in [k for k, v in result.items()
                        if isinstance(v, dict) and v.get("family", re.sub(r"_[0-9a-f]{8}$", "", k)) == family and
                        v.get("workload", wl) == wl and k != kernel]:
                del result[old]
            # vector-issue occupancy from the third PMC pass: every wave executes a vector instruction for
            # SQ_ACTIVE_INST_VALU of its SQ_WAVE_CYCLES (both in quad-cycles; one instruction = one quad-cycle on
            # gfx950) = `valu_share_per_wave`; times the waves that share a SIMD = the average number of waves per
            # SIMD inside a vector instruction.  Not a fraction of a roofline by itself: f32 instructions of two waves
            # overlap (the dense kernel reaches 1.34), f64 ones hold the double-precision pipe (C2 sits at 1.02).
            active, cycles, waves = (counter_means(root, "SQ_ACTIVE_INST_VALU"), counter_means(root, "SQ_WAVE_CYCLES"),
                                     counter_means(root, "SQ_WAVES"))
            busy = None
            if kernel in active and kernel in cycles and kernel in waves and cycles[kernel] > 0:
                # (a launch with more waves than fit at once keeps `resident` of them on a SIMD)
                resident = resident_waves_per_simd(root).get(kernel, waves[kernel] / 1024.0)
                busy = active[kernel] / cycles[kernel] * min(waves[kernel] / 1024.0, resident)
            result[kernel] = {
                "kernel": kernel, "family": family, "workload": wl,
                "valu_busy": busy,
                "valu_share_per_wave": (active[kernel] / cycles[kernel]) if (kernel in active and kernel in cycles and cycles[kernel] > 0) else None,
                "valu_instructions_per_launch": active.get(kernel),
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
    slab_records(tag, result)
    with open(out_path, "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
