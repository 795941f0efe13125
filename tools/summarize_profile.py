#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by tools/profile_r1.sh / tools/profile.sh) into
profiles/<name>/: the rocprofv3 kernel_stats.csv files plus summary.json with the
per-launch means of every PMC counter for the carve kernels."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_key(name):
    for k in ("coarse_fill", "coarse", "classify", "exact", "fused", "fill"):
        if "carve_" + k in name:
            return k
    # arvx_set_views_device (round 1 names, then csrc/views_kernels.h)
    for k in ("mask_to_bits", "sat_rows", "sat_cols", "views_strip", "views_bits", "views_tile_sums",
              "views_table", "views_rows", "views_cols"):
        if k in name:
            return k
    return None


def main():
    name = sys.argv[1]
    tags = sys.argv[2:]
    out_dir = os.path.join(ROOT, "profiles", name)
    os.makedirs(out_dir, exist_ok=True)
    summary = {}
    for tag in tags:
        d = {}
        base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
        for sub in ("fetch", "write", "sq", "l2"):
            for f in glob.glob(f"{base}/{sub}/runc/*_counter_collection.csv"):
                agg = collections.defaultdict(lambda: collections.defaultdict(list))
                for r in csv.DictReader(open(f)):
                    k = kernel_key(r["Kernel_Name"])
                    if k:
                        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                for k, cs in agg.items():
                    for c, v in cs.items():
                        d.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v)}
        for f in glob.glob(f"{base}/stats/runc/*_kernel_stats.csv"):
            os.makedirs(os.path.join(out_dir, tag), exist_ok=True)
            shutil.copy(f, os.path.join(out_dir, tag, "kernel_stats.csv"))
            for r in csv.DictReader(open(f)):
                k = kernel_key(r["Name"])
                if k:
                    d.setdefault(k, {})["kernel_avg_ns"] = float(r["AverageNs"])
                    d[k]["kernel_calls"] = int(r["Calls"])
        summary[tag] = d
    json.dump(summary, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
    for tag, d in summary.items():
        for k, cs in d.items():
            print(tag, k, {c: (round(v["mean"]) if isinstance(v, dict) else v) for c, v in cs.items()})


if __name__ == "__main__":
    main()
