#!/usr/bin/env python3
"""Condense rocprofv3 csv outputs (kernel stats + per-dispatch PMC) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
DRIVER = "--driver" in sys.argv or os.environ.get("PROFILE_CMD", "").endswith("traffic_driver.py")   # the profiled command was the kernel driver


def find(pat):
    return sorted(glob.glob(os.path.join(out, pat), recursive=True))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:70]


for f in find("trace/**/*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(f"  {short(r['Name']):70s} calls {r['Calls']:>5s} total_ns {r['TotalDurationNs']:>12s} "
              f"avg_ns {float(r['AverageNs']):>12.0f} pct {r['Percentage']:>6s}")
for f in find("trace/**/*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    agg = defaultdict(list)
    for r in rows:
        agg[(short(r["Kernel_Name"]), r.get("Grid_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""),
             r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("== kernel trace (per kernel x grid):")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
        v2 = sorted(v)
        print(f"  {k[0]:70s} grid {k[1]:>8s} vgpr {k[2]:>4s} agpr {k[3]:>4s} lds {k[4]:>6s} scr {k[5]:>3s} n {len(v):4d} "
              f"avg_us {sum(v) / len(v) / 1e3:10.1f} med_us {v2[len(v2) // 2] / 1e3:10.1f} min_us {v2[0] / 1e3:10.1f}")
for d in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write", "pmc_tcc"):
    for f in find(f"{d}/**/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = defaultdict(lambda: defaultdict(list))
        # the coarse (262144-point) and fine (786432-point) MLP launches share kernel name and grid: they
        # alternate, so split them by dispatch parity within the kernel (warm-up pack/first launches are even)
        seen = defaultdict(dict)
        for r in sorted(rows, key=lambda r: int(r.get("Dispatch_Id", 0))):
            name = short(r["Kernel_Name"])
            if DRIVER and ("mlp_lp" in name or "mlp_x3" in name or "mlp_fused" in name):
                # scripts/diag/traffic_driver.py: REPS launches per kernel and shape, in a fixed order
                d = seen[name]
                if r["Dispatch_Id"] not in d:
                    d[r["Dispatch_Id"]] = len(d)
                name += f" [launch set {d[r['Dispatch_Id']] // 6}]"
            elif "mlp_fused" in name or "mlp_lp_kernel" in name:
                d = seen[name]
                if r["Dispatch_Id"] not in d:
                    d[r["Dispatch_Id"]] = len(d)
                name += " [fine]" if d[r["Dispatch_Id"]] % 2 else " [coarse]"
            agg[(name, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"== pmc pass: mean counter value per dispatch")
        for k, cs in sorted(agg.items(), key=lambda kv: -len(kv[1]))[:40]:
            if not any(t in k[0] for t in ("mlp_", "composite", "importance", "wgrad", "pair_")):
                continue
            print(f"  {k[0]:60s} grid {k[1]:>8s} " + "  ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(cs.items())))
