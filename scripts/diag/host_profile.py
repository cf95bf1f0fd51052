#!/usr/bin/env python3
"""cProfile of the host side of a bench configuration (where do the ~2 ms of Python per C3 step go?).
usage: host_profile.py [c3|c4]"""
import cProfile
import io
import os
import pstats
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
sys.argv = ["bench.py", "--config", cfg, "--steps", "100", "--warmup", "5", "--no-cpu-baseline", "--no-variants"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(45)
print(out.getvalue())
