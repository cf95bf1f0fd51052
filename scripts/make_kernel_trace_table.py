#!/usr/bin/env python3
"""profiles/<round>/a_kernel_trace_table.txt from the --kernel-trace pass of scripts/profile_traffic.sh: per kernel and grid size
the registers, scratch bytes, LDS bytes, number of launches and median duration.
usage: make_kernel_trace_table.py gpurun_out/traffic_<tag> <out.txt>"""
import csv
import os
import statistics
import sys
from collections import OrderedDict

prof, out = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(os.path.join(prof, "trace", "trace_kernel_trace.csv"))))
groups = OrderedDict()
for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("nsos::lp::", "").split("(")[0]
    if "pack" in name or "at::native" in name or "rocclr" in name:
        continue
    grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))
    groups.setdefault((name, grid), []).append(r)
lines = ["# rocprofv3 --kernel-trace of scripts/diag/traffic_driver.py (the round's final kernels): per kernel and launch shape;",
         "# launches alternate shapes in the driver's order (see scripts/diag/traffic_driver.py), median over each kernel's launches",
         f"{'kernel':50s} {'grid':>8s} {'vgpr':>5s} {'agpr':>5s} {'scratch_B':>9s} {'lds_B':>7s} {'n':>3s} {'median_us':>10s}"]
for (name, grid), rs in groups.items():
    d = statistics.median(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
    r0 = rs[0]
    lines.append(f"{name[:50]:50s} {grid:8d} {r0.get('VGPR_Count', '?'):>5s} {r0.get('Accum_VGPR_Count', '?'):>5s} {r0.get('Scratch_Size', '?'):>9s} "
                 f"{r0.get('LDS_Block_Size', '?'):>7s} {len(rs):3d} {d:10.1f}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
