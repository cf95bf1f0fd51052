#!/usr/bin/env python3
"""One-off stress run of tests/test_generic_arch.py's seeded fuzz beyond the 14 committed seeds: random constructor arguments (depth 1-12,
width 8-320, every head shape, with / without view directions and embedding), forward + every parameter's gradient (+ ray gradients on odd
seeds) against the CPU port.  Usage: fuzz_generic.py [first_seed last_seed]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import pytest
import test_generic_arch as t
a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (14, 140)
bad = skipped = 0
for seed in range(a, b):
    try:
        t.test_generic_fuzz_forward_and_gradients_vs_port(seed)
    except pytest.skip.Exception:
        skipped += 1
    except Exception as e:
        bad += 1
        print("seed", seed, "FAILED:", repr(e)[:400], flush=True)
print(f"done: seeds {a}..{b - 1}, failures {bad}, skipped {skipped}")
