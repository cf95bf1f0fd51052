#!/usr/bin/env python3
"""profiles/<round>/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_traffic.sh (separate rocprofv3 --pmc
runs of scripts/diag/traffic_driver.py, which launches the fine-pass MLP kernel of every precision path a fixed number of times):
per kernel x launch shape the median counter values, the HBM bytes per launch after the guide's gfx950 correction, the
algorithmic bytes next to them, and the hash of the kernel sources the passes were measured on (bench.py reports
`roofline.traffic` for a variant only while that hash still holds).

usage: make_traffic_json.py gpurun_out/traffic_<tag> profiles/r06/traffic.json [<summary text file to write>]
"""
import csv
import json
import os
import statistics
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

prof, out = sys.argv[1], sys.argv[2]
summary_path = sys.argv[3] if len(sys.argv) > 3 else None
REPS = 6          # scripts/diag/traffic_driver.py
S = 192

# (key used by bench.py, kernel-name test, launch index range inside that kernel's dispatch order, rays, C, SAVE bytes/pt, sem)
LAUNCHES = [
    ("c2_fp32", lambda n: "mlp_fused_kernel<0, true, 0>" in n, 0, 4096, 4, 0, 0),
    ("c2_fp16x3", lambda n: "mlp_x316_kernel<0, false>" in n, 0, 4096, 4, 0, 0),
    ("c5_fp16", lambda n: "mlp_lp16_kernel" in n and "F16, 2, false" in n and "BF16" not in n, 0, 65536, 6, 0, 2),
    ("c3_bf16", lambda n: "mlp_lp16_kernel" in n and "BF16, 2, true" in n, 0, 4096, 6, 896, 2),
    ("c4_bf16", lambda n: "mlp_lp16_kernel" in n and "BF16, 2, true" in n, 1, 8192, 6, 896, 2),
    ("bf16_inference_4096", lambda n: "mlp_lp16_kernel" in n and "BF16, 2, false" in n, 0, 4096, 6, 0, 2),
    ("bf16_inference_8192", lambda n: "mlp_lp16_kernel" in n and "BF16, 2, false" in n, 1, 8192, 6, 0, 2),
]
PACKED_BYTES = {("fp32", 0): 4096 + 73 * 36864, ("x3", 0): None, ("lp", 2): None}


def rows_of(kind, counter):
    path = os.path.join(prof, f"pmc_{kind}", "pmc_counter_collection.csv")
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "pack" not in r["Kernel_Name"]]
    return sorted(rows, key=lambda r: int(r["Dispatch_Id"]))


fetch_rows, write_rows = rows_of("fetch", "FETCH_SIZE"), rows_of("write", "WRITE_SIZE")
trace = {}
tpath = os.path.join(prof, "trace", "trace_kernel_trace.csv")
if os.path.exists(tpath):
    for r in csv.DictReader(open(tpath)):
        trace.setdefault(r["Kernel_Name"], []).append(r)

doc = OrderedDict()
doc["source"] = (f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (no trace domain combined with --pmc), of "
                 f"scripts/diag/traffic_driver.py: {REPS} launches per kernel and shape, medians; durations and register / scratch "
                 f"columns from a --kernel-trace pass of the same driver ({os.path.basename(prof)})")
doc["correction"] = ("MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports half the bytes of wide (16 B / lane) coalesced reads "
                     "-> x2; WRITE_SIZE calibrates exactly on these kernels (c2: 12288 KB = 786432 points x 16 B; c5: 294912 KB = "
                     "12582912 points x 24 B).  hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE")
doc["isolated_launch_median_us"] = ("duration of the driver's back-to-back launches of ONE kernel under the profiler -- NOT bench.py's kernel_ms, which is the "
                                    "HIP-event time of the same kernel inside a whole step (cold L2 weights, a different clock state): the short 16-bit launches "
                                    "differ by 5-10 % between the two; the counters are per launch and do not depend on it")
doc["kernels"] = OrderedDict()
lines = []
for key, match, idx, rays, C, save, sem in LAUNCHES:
    f = [float(r["Counter_Value"]) for r in fetch_rows if match(r["Kernel_Name"])][idx * REPS:(idx + 1) * REPS]
    w = [float(r["Counter_Value"]) for r in write_rows if match(r["Kernel_Name"])][idx * REPS:(idx + 1) * REPS]
    if len(f) != REPS or len(w) != REPS:
        raise SystemExit(f"{key}: expected {REPS} launches in each pass, found {len(f)} / {len(w)}")
    name = next(r["Kernel_Name"] for r in fetch_rows if match(r["Kernel_Name"]))
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("nsos::lp::", "").split("(")[0]
    pts = rays * S
    f_kb, w_kb = statistics.median(f), statistics.median(w)
    # algorithmic: z in (4 B / point), raw out (4 C), the SAVE operands, per-ray o / d / viewdir (36 B), one pass over the packed weights
    packed = {"c2_fp32": 4096 + 73 * 36864}.get(key)
    algo = pts * (4 + 4 * C + save) + rays * 36
    tr = [r for r in trace.get(name, [])][idx * REPS:(idx + 1) * REPS]
    dur = statistics.median(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr) / 1e3 if tr else None
    e = OrderedDict(kernel=short, rays=rays, points=pts, fetch_size_kb=round(f_kb, 1), write_size_kb=round(w_kb, 1),
                    hbm_bytes_per_launch=int(round((2 * f_kb + w_kb) * 1024)), algorithmic_bytes_per_launch_without_weights=int(algo),
                    ratio=round((2 * f_kb + w_kb) * 1024 / algo, 3),
                    kernel_source_sha16=bench.kernel_source_hash(key), isolated_launch_median_us=dur,
                    vgpr=tr[0].get("VGPR_Count") if tr else None, agpr=tr[0].get("Accum_VGPR_Count") if tr else None,
                    scratch_bytes=tr[0].get("Scratch_Size") if tr else None, lds_bytes=tr[0].get("LDS_Block_Size") if tr else None)
    doc["kernels"][key] = e
    lines.append(f"{key:22s} {short:44s} rays {rays:6d}  FETCH_SIZE {f_kb:10.1f} KB  WRITE_SIZE {w_kb:10.1f} KB  HBM {e['hbm_bytes_per_launch'] / 1e6:9.2f} MB  "
                 f"algorithmic {algo / 1e6:9.2f} MB (+ weights)  x{e['ratio']:.2f}  {dur if dur is None else round(dur, 1)} us  vgpr {e['vgpr']} agpr {e['agpr']} "
                 f"scratch {e['scratch_bytes']} B  lds {e['lds_bytes']}")
doc["note"] = ("reads = one fill of the packed weight stream per XCD L2 (8 x 2.7-2.9 MiB) + z_vals + per-ray data; the kernels are MFMA-bound "
               "(a few GB/s to ~100 GB/s), except the training variant whose saved operands are real HBM traffic (896 B per point)")
json.dump(doc, open(out, "w"), indent=1)
text = "\n".join(lines)
print(text)
if summary_path:
    open(summary_path, "w").write("# " + doc["source"] + "\n# " + doc["correction"] + "\n" + text + "\n")
