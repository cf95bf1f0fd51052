import sys, os, runpy, warnings
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.set_sync_debug_mode("warn")
warnings.simplefilter("always")
sys.argv = ["bench.py", "--config", os.environ.get("CFG", "c3"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-variants"]
runpy.run_path("bench.py", run_name="__main__")
