"""Scratch (spill) instructions and register counts of every MLP kernel in the built library, with the code around each
scratch access of one chosen kernel.  usage: scratch_report.py [substring of the kernel to list]"""
import importlib.util, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("chk", os.path.join(ROOT, "scripts", "check_lds_ring.py"))
chk = importlib.util.module_from_spec(spec); spec.loader.exec_module(chk)
lib = os.path.join(ROOT, "nerf-sos_amd", "libnerf_sos_hip.so")
ks = {k: v for k, v in chk.disassemble(lib).items() if "mlp_" in k and "pack" not in k}
for name, ins in sorted(ks.items()):
    n = [i for i in ins if i.startswith("scratch_")]
    if "lp8" in name or n:
        print(f"{len(n):4d} scratch ops, {len(ins):6d} instructions  {name}")
pick = sys.argv[1] if len(sys.argv) > 1 else None
if pick:
    for name, ins in ks.items():
        if pick in name:
            idx = [i for i, t in enumerate(ins) if t.startswith("scratch_")]
            print("==", name)
            for i in idx:
                print(f"  [{i}] " + " | ".join(ins[max(0, i - 2): i + 3]))
