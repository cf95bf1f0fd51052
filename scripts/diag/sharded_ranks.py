#!/usr/bin/env python3
"""tests/test_gpu_sharded.py's worker for an arbitrary number of ranks sharing cuda:0 over gloo, with every rank's traceback
printed.   usage: sharded_ranks.py [world]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.multiprocessing as mp


def run(rank, world, port):
    import test_gpu_sharded as t

    class Q:
        def put(self, x):
            print("RESULT", x, flush=True)
    try:
        t._worker(rank, world, port, "gloo", Q())
    except BaseException:
        print(f"RANK {rank} FAILED:\n" + traceback.format_exc(), flush=True)
        raise


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    mp.spawn(run, args=(world, 29877), nprocs=world)
