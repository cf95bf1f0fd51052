#!/usr/bin/env python3
"""profiles/<round>/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_gpu.sh (separate rocprofv3 --pmc
runs of the default bench command): median over the fine-pass launches of the headline kernel, with the hash of the kernel
sources the passes were measured on (bench.py reports `roofline.traffic` only while that hash still holds).
usage: make_traffic_json.py gpurun_out/prof_<tag> profiles/r02/traffic.json <summary file the numbers are quoted from>"""
import csv
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

prof, out, quoted = sys.argv[1], sys.argv[2], sys.argv[3]


def fine_pass_values(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "mlp_fused_kernel" in r["Kernel_Name"]]
    grids = sorted({int(r["Grid_Size"]) for r in rows})
    # the coarse (262144 points) and the fine (786432 points) pass run the same persistent grid: tell them apart by the counter
    vals = sorted(float(r["Counter_Value"]) for r in rows)
    half = len(vals) // 2
    return vals[half:], grids          # the larger half = the fine passes (3x the points: more bytes)


fetch, _ = fine_pass_values(os.path.join(prof, "pmc_fetch", "pmc_counter_collection.csv"), "FETCH_SIZE")
write, _ = fine_pass_values(os.path.join(prof, "pmc_write", "pmc_counter_collection.csv"), "WRITE_SIZE")
f_kb, w_kb = statistics.median(fetch), statistics.median(write)
n_pts = 4096 * 192
doc = {
    "source": f"{quoted} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, of `python bench.py --steps 10 --warmup 2 "
              f"--no-cpu-baseline`; medians over the {len(fetch)} fine-pass launches)",
    "kernel": f"mlp_fused_kernel<0,true,0> fine pass ({n_pts} points)",
    "kernel_source_sha16": bench.kernel_source_hash(),
    "fetch_size_kb": round(f_kb, 1),
    "write_size_kb": round(w_kb, 1),
    "correction": "FETCH_SIZE x2 for wide (16 B/lane) coalesced reads on gfx950 (MI355X_MICROARCH.md HBM section); WRITE_SIZE "
                  f"calibrates exactly: {n_pts * 16 // 1024} KB = {n_pts} points x 16 B of raw output",
    "hbm_bytes_per_launch": int(round((2 * f_kb + w_kb) * 1024)),
    "algorithmic_bytes_per_launch": 18628608,
    "note": "measured on the build whose kernel sources (mlp_fused.hip + mlp_common.h) hash to kernel_source_sha16; bench.py reports "
            "it only while that still holds. reads = one fill of the 2.7 MiB packed weight stream per XCD L2 (8x) + z_vals (3.1 MB) "
            "+ per-ray data; ~6 GB/s -- the kernel is MFMA-bound, the 2x over the algorithmic bytes is the per-XCD weight fill",
}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc, indent=1))
