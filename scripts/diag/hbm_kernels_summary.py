#!/usr/bin/env python3
"""Per-launch counter table of scripts/profile_hbm_kernels.sh's passes: for ray_setup_kernel, composite_importance_kernel (eval /
train launches told apart by dispatch order) and composite_kernel<3> at the C5 chunk -- instructions by class, wave cycles, busy
cycles, LDS waits, HBM bytes (2 x FETCH_SIZE + WRITE_SIZE: MI355X_MICROARCH.md's gfx950 correction) next to the algorithmic bytes and
the duration of the kernel-trace pass.  usage: hbm_kernels_summary.py gpurun_out/hbm_<tag>"""
import csv
import glob
import os
import statistics
import sys
from collections import OrderedDict, defaultdict

prof = sys.argv[1]
REPS = 6
R = int(os.environ.get("NSOS_HBM_RAYS", "65536"))
NC, NF, NI = 64, 192, 128
CASES = OrderedDict([
    ("ray_setup_kernel", (lambda n: "ray_setup_kernel" in n, 0, R * (12 + 8 + 4 * NC + 12))),
    ("composite_importance_kernel eval", (lambda n: "composite_importance_kernel" in n, 0, R * (4 * NC * 6 + 4 * NC + 12 + 4 * NC + 32 + 4 * NF + 4 * NI + 4))),
    ("composite_importance_kernel train", (lambda n: "composite_importance_kernel" in n, 1, R * (4 * NC * 6 + 4 * NC + 12 + 4 * NC + 32 + 4 * NF + 4 * NI + 4 + 4 * NI + 4 * NC))),
    ("composite_kernel<3>", (lambda n: "composite_kernel" in n and "importance" not in n, 0, R * (4 * NF * 6 + 4 * NF + 12 + 4 * NF + 32))),
])
vals = defaultdict(dict)          # case -> counter -> median per launch
for path in sorted(glob.glob(os.path.join(prof, "pmc_*", "pmc_counter_collection.csv"))):
    rows = list(csv.DictReader(open(path)))
    for counter in sorted({r["Counter_Name"] for r in rows}):
        mine = sorted((r for r in rows if r["Counter_Name"] == counter), key=lambda r: int(r["Dispatch_Id"]))
        for case, (match, idx, _) in CASES.items():
            v = [float(r["Counter_Value"]) for r in mine if match(r["Kernel_Name"])]
            # the first composite_importance launch of the driver is the set-up call (z1): skip it
            skip = 1 if "composite_importance" in case else 0
            # ray_setup also has one set-up launch
            skip = 1 if case == "ray_setup_kernel" else skip
            v = v[skip + idx * REPS: skip + (idx + 1) * REPS]
            if v:
                vals[case][counter] = statistics.median(v)
dur = {}
tpath = os.path.join(prof, "trace", "trace_kernel_trace.csv")
if os.path.exists(tpath):
    rows = sorted(csv.DictReader(open(tpath)), key=lambda r: int(r["Start_Timestamp"]))
    for case, (match, idx, _) in CASES.items():
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if match(r["Kernel_Name"])]
        skip = 1 if ("composite_importance" in case or case == "ray_setup_kernel") else 0
        d = d[skip + idx * REPS: skip + (idx + 1) * REPS]
        if d:
            dur[case] = statistics.median(d)
print(f"HBM / latency-bound kernels at {R} rays, 6 channels: rocprofv3 --pmc passes (one counter group per pass), medians of {REPS} launches")
for case, (_, _, algo) in CASES.items():
    c = vals.get(case, {})
    print(f"\n== {case}")
    us = dur.get(case)
    hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024 if "FETCH_SIZE" in c else None
    print(f"   duration {us:.1f} us" if us else "   duration n/a", f"| algorithmic {algo / 1e6:.1f} MB" + (f" = {algo / us / 1e6:.2f} TB/s" if us else ""),
          (f"| HBM counters {hbm / 1e6:.1f} MB (x{hbm / algo:.2f} algorithmic) = {hbm / us / 1e6:.2f} TB/s" if hbm and us else ""))
    waves = c.get("SQ_WAVES")
    for k in sorted(c):
        per_wave = f"  ({c[k] / waves:10.1f} per wave)" if waves and k.startswith("SQ_") and k != "SQ_WAVES" else ""
        print(f"   {k:28s} {c[k]:16.0f}{per_wave}")
    if "SQ_INSTS_VALU" in c and "SQ_WAVE_CYCLES" in c and waves:
        # one VALU instruction of a 64-lane wave occupies its SIMD for 4 cycles (16 lanes per cycle)
        print(f"   -> VALU issue cycles per wave ~ 4 x {c['SQ_INSTS_VALU'] / waves:.0f} = {4 * c['SQ_INSTS_VALU'] / waves:.0f}; wave lifetime {c['SQ_WAVE_CYCLES'] / waves:.0f} cycles "
              f"(x 4: the counter ticks every 4 cycles on some parts -- compare with duration)")
