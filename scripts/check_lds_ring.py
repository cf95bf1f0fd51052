#!/usr/bin/env python3
"""Static guard for the hand-counted LDS read rings in the MLP kernels.

The kernels issue their A-operand `ds_read_b128`s from inline asm and wait for them with hand-counted
`s_waitcnt lgkmcnt(n)` (nerf-sos_amd/csrc/mlp_common.h: a_pipeline).  hipcc knows nothing about that protocol: it
treats an asm output as defined the moment the asm statement ends, so under register pressure it may SPILL (or
copy) a destination register before the data has landed -- a silent race that showed up as run-to-run noise in the
reduced-precision kernel.  This script disassembles the gfx950 code objects inside the built library and replays the
protocol: LDS reads return in order, `s_waitcnt lgkmcnt(n)` retires all but the youngest n, and any instruction
that touches a register whose read is still outstanding is an error.  It also reports scratch (spill) traffic.

Usage: python scripts/check_lds_ring.py [path/to/libnerf_sos_hip.so]     (exit status 1 on a violation)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

_REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(text):
    """Set of ('v'|'a', index) named in an operand string."""
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def disassemble(lib_path):
    """{kernel name: [instruction text, ...]} for every gfx950 code object bundled in lib_path."""
    kernels = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" not in name:
                continue
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, name)], check=True,
                                  capture_output=True, text=True).stdout
            cur = None
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = kernels.setdefault(m.group(1), [])
                    continue
                if cur is None or not line.startswith(("\t", " ")):
                    continue
                ins = line.split("//")[0].strip()
                if ins:
                    cur.append(ins)
    return kernels


def check_kernel(ins_list):
    """Replay the in-order LDS queue.  Returns (violations, n_lds_reads, n_scratch_ops)."""
    pending = []  # [(dest regs, instruction index)] oldest first
    bad = []
    n_reads = n_scratch = 0
    for idx, ins in enumerate(ins_list):
        op, _, rest = ins.partition(" ")
        if op.startswith("scratch_"):
            n_scratch += 1
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                keep = int(m.group(1))
                pending = pending[len(pending) - keep:] if keep else []
            continue
        if op.startswith("s_"):
            # divergent control flow: hipcc's own LDS reads under an exec mask are its business (it waits at the join
            # point, which a linear replay cannot see).  The hand-written rings never span a branch or an exec change.
            if op.startswith(("s_cbranch", "s_branch")) or re.search(r"\bexec\b", rest.split(",")[0]) or "saveexec" in op:
                pending = []
            continue
        is_lds_read = op.startswith("ds_read") or op.startswith("ds_load")
        # a later LDS read may overwrite a pending destination (returns are in order: plain WAW on a dead value,
        # which is what hipcc makes of the never-consumed reads of a padded chunk); its address operand may not
        touched = regs_of(rest.partition(",")[2] if is_lds_read else rest)
        for dest, at in pending:
            if dest & touched:
                bad.append((idx, ins, ins_list[at]))
                break
        if is_lds_read:
            n_reads += 1
            pending.append((regs_of(rest.split(",")[0]), idx))
        elif op.startswith("ds_"):
            # every LDS operation occupies a slot of the in-order queue that lgkmcnt counts: writes and ds_swizzle / ds_permute /
            # ds_bpermute (their results carry a destination) as well as reads.  A wait of lgkmcnt(n) that a compiler-issued
            # ds_bpermute pair makes look loose is not: the two youngest slots are theirs, an older ds_read has landed.
            has_dest = op.startswith(("ds_bpermute", "ds_permute", "ds_swizzle")) or "_rtn" in op
            pending.append((regs_of(rest.split(",")[0]) if has_dest else set(), idx))
    return bad, n_reads, n_scratch


def main(argv):
    lib = argv[1] if len(argv) > 1 else os.path.join(ROOT, "nerf-sos_amd", "libnerf_sos_hip.so")
    kernels = disassemble(lib)
    status = 0
    for name, ins in sorted(kernels.items()):
        if "mlp_" not in name or "pack" in name:
            continue
        bad, n_reads, n_scratch = check_kernel(ins)
        print(f"{name[:90]:90s} ds_reads={n_reads:5d} scratch_ops={n_scratch:4d} violations={len(bad)}")
        for idx, what, read in bad[:5]:
            print(f"    #{idx}: `{what}` touches the destination of a still-pending `{read}`")
        if bad:
            status = 1
    return status


if __name__ == "__main__":
    sys.exit(main(sys.argv))
