#!/usr/bin/env python3
"""cProfile of the host side of a bench configuration's TIMED loop only (where do the ~2 ms of Python per C3 step go?).
usage: host_profile.py [c3|c4] [n_lines]"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_lines = int(sys.argv[2]) if len(sys.argv) > 2 else 60
import bench

pr = cProfile.Profile()
orig_timed = bench.timed


def timed(ctx, step, warmup, steps):
    def prof_step(i):
        if i >= warmup:
            pr.enable()
        step(i)
        pr.disable()
    return orig_timed(ctx, prof_step, warmup, steps)


bench.timed = timed
sys.argv = ["bench.py", "--config", cfg, "--steps", "100", "--warmup", "5", "--no-cpu-baseline", "--no-variants"]
try:
    bench.main()
except SystemExit:
    pass
out = io.StringIO()
st = pstats.Stats(pr, stream=out)
st.sort_stats("cumulative").print_stats(n_lines)
print(out.getvalue())
