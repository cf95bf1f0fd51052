"""Where a training step's wall time goes, from a rocprofv3 --kernel-trace CSV of scripts/diag/graph_step_time.py <B> <capture>
<overlap>: the last `steps` steps (a step = from one render_draws_kernel to the next), kernel time per step, idle time between
consecutive kernels (queue gaps), and the kernels in front of the largest gaps.
usage: step_timeline.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
marks = [i for i, e in enumerate(ev) if "render_draws_kernel" in e[2]]
marks = marks[-(steps + 1):]
sel = ev[marks[0]:marks[-1]]
wall = (ev[marks[-1]][0] - ev[marks[0]][0]) / steps / 1e3
busy_end, busy, gaps, gap_after = sel[0][0], 0, 0, defaultdict(lambda: [0, 0])
prev = None
for s, e, n in sel:
    if s > busy_end:
        gaps += s - busy_end
        if prev:
            gap_after[prev][0] += s - busy_end
            gap_after[prev][1] += 1
    busy += max(0, e - max(s, busy_end))
    if e > busy_end:
        busy_end, prev = e, n
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")[:70]
print(f"{wall:8.1f} us wall per step; GPU busy {busy / steps / 1e3:8.1f} us, idle between kernels {gaps / steps / 1e3:8.1f} us, {len(sel) / steps:.1f} launches per step")
print("idle time in front of the next kernel, by the kernel that ended before it (us per step, gaps per step, mean gap us):")
for n, (t, c) in sorted(gap_after.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {t / steps / 1e3:7.1f} {c / steps:5.1f} {t / c / 1e3:6.2f}  {short(n)}")
dur = defaultdict(lambda: [0, 0])
for s, e, n in sel:
    dur[n][0] += e - s
    dur[n][1] += 1
print("kernel time per step (us, launches per step):")
for n, (t, c) in sorted(dur.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"  {t / steps / 1e3:7.1f} {c / steps:5.1f}  {short(n)}")
