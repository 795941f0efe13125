#!/usr/bin/env python3
"""profiles/<name>/summary.json (tools/summarize_profile.py) -> profiles/traffic.json: the PMC
byte counts bench.py prints next to its timings.

FETCH_SIZE / WRITE_SIZE are rocprofv3's derived counters in KiB, collected in separate passes
(tools/profile.sh).  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced reads, WRITE_SIZE is exact for wide stores, other
widths are uncalibrated -- `bytes_per_launch` therefore uses 2 x FETCH + WRITE (an upper bound
for these kernels, whose reads are mostly 2..16-byte table and record accesses) and the raw
counters are kept beside it.

    python tools/make_traffic.py r2_kernel_v14 r2_512:512x512x512x36 r2_1024:1024x1024x1024x36
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ar_voxel_project_amd import build  # noqa: E402  (source_stamp: ties the pass to the kernel sources)

CARVE = ("coarse_fill", "coarse", "fill", "classify", "exact", "fused")
VIEWS = ("views_strip", "views_bits", "views_tile_sums", "views_table", "views_rows", "views_cols",
         "mask_to_bits", "sat_rows", "sat_cols")


def main():
    name = sys.argv[1]
    summ = json.load(open(os.path.join(ROOT, "profiles", name, "summary.json")))
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    out = json.load(open(tpath)) if os.path.exists(tpath) else {}  # entries of other runs stay
    for arg in sys.argv[2:]:
        tag, key = arg.split(":")
        ks = summ[tag]

        def total(group, counter):
            return sum(ks[k].get(counter, {}).get("mean", 0.0) for k in group if k in ks)

        def ns(group):
            return sum(ks[k].get("kernel_avg_ns", 0.0) for k in group if k in ks)

        f_c, w_c = total(CARVE, "FETCH_SIZE") * 1024, total(CARVE, "WRITE_SIZE") * 1024
        f_v, w_v = total(VIEWS, "FETCH_SIZE") * 1024, total(VIEWS, "WRITE_SIZE") * 1024
        out[key] = {
            "source_stamp": build.source_stamp(),  # of the sources in THIS tree: run on the tree that was profiled
            "bytes_per_launch": 2 * f_c + w_c,
            "bytes_per_step": 2 * (f_c + f_v) + w_c + w_v,
            "raw": {"carve_FETCH_SIZE_bytes": f_c, "carve_WRITE_SIZE_bytes": w_c,
                    "views_FETCH_SIZE_bytes": f_v, "views_WRITE_SIZE_bytes": w_v},
            "kernel_avg_ns_rocprofv3": {k: ks[k]["kernel_avg_ns"] for k in CARVE + VIEWS
                                        if k in ks and "kernel_avg_ns" in ks[k]},
            "carve_kernels_ns": ns(CARVE), "views_kernels_ns": ns(VIEWS),
            "valu_wave_instructions": {k: ks[k]["SQ_INSTS_VALU"]["mean"] for k in CARVE
                                       if k in ks and "SQ_INSTS_VALU" in ks[k]},
            "valu_wave_instructions_views": {k: ks[k]["SQ_INSTS_VALU"]["mean"] for k in VIEWS
                                             if k in ks and "SQ_INSTS_VALU" in ks[k]},
            "source": f"profiles/{name}/summary.json tag {tag} (rocprofv3 --pmc FETCH_SIZE / "
                      "WRITE_SIZE / SQ_* in separate passes over `bench.py --steps 10 --warmup 2 "
                      "--no-cpu --no-ablation --no-workloads --jobs 1 --jobs-in-flight 0 --no-e2e --rounds 1 --extra-grid 0`, tools/profile.sh; bytes = 2 x FETCH "
                      "+ WRITE per the guide's gfx950 correction, raw counters beside it)"}
    json.dump(out, open(tpath, "w"), indent=1)
    for k, v in out.items():
        print(k, {a: (round(b) if isinstance(b, float) else b) for a, b in v.items()
                  if a in ("bytes_per_launch", "bytes_per_step", "carve_kernels_ns", "views_kernels_ns")})


if __name__ == "__main__":
    main()
