"""Dump the gfx950 ISA of the kernels whose (mangled) name contains argv[1] into /tmp/isa_<n>.s and print register /
scratch statistics (diagnostic helper for hand-scheduled kernels)."""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
pat = sys.argv[1]
tmp = tempfile.mkdtemp()
shutil.copy(os.path.join(ROOT, "nerf-sos_amd", "libnerf_sos_hip.so"), tmp + "/lib.so")
subprocess.run([OBJDUMP, "--offloading", tmp + "/lib.so"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
k = 0
for n in sorted(os.listdir(tmp)):
    if "amdgcn" not in n:
        continue
    text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", tmp + "/" + n], capture_output=True, text=True).stdout
    cur, name = None, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if cur is not None:
                open(f"/tmp/isa_{k}.s", "w").write("\n".join(cur)); print(f"/tmp/isa_{k}.s", name, len(cur), "lines,", sum("scratch_" in l for l in cur), "scratch ops"); k += 1
            name = m.group(1)
            cur = [] if pat in name else None
            continue
        if cur is not None:
            cur.append(line)
    if cur is not None:
        open(f"/tmp/isa_{k}.s", "w").write("\n".join(cur)); print(f"/tmp/isa_{k}.s", name, len(cur), "lines,", sum("scratch_" in l for l in cur), "scratch ops"); k += 1
