#!/usr/bin/env python3
"""Static instruction mix of the trunk kernels: hipcc -S on conv_trunk.hip, then count instruction classes per kernel and
per basic block (blocks with MFMAs are the pipelined steps; blocks without are prologue / epilogue / address code).
usage: tools/isa_mix.py [kernel-substring]   (reads /tmp/ct.s if it exists, else compiles it)"""
import os, re, subprocess, sys
from collections import Counter
S = "/tmp/ct.s"
if not os.path.exists(S) or "--rebuild" in sys.argv:
    src = os.path.join(os.path.dirname(__file__), "..", "sentinel2-super-resolution-poc_amd", "csrc", "conv_trunk.hip")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S",
                    "--cuda-device-only", "-w", src, "-o", S], check=True)
want = [a for a in sys.argv[1:] if not a.startswith("--")]
lines = open(S).read().split("\n")
def cls(i):
    if i.startswith("v_mfma"): return "mfma"
    if i.startswith("v_accvgpr"): return "acc_mov"
    if i.startswith("v_"): return "valu"
    if i.startswith("s_waitcnt") or i.startswith("s_barrier") or i.startswith("s_nop"): return "wait"
    if i.startswith("s_"): return "salu"
    if i.startswith("ds_"): return "lds"
    if i.startswith(("global_", "buffer_", "scratch_")): return "vmem"
    return "other"
name = None
for ln in lines:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name = m.group(1); blocks = []; cur = Counter(); label = "entry"
        continue
    if name is None: continue
    if re.match(r"^\.LBB\d+_\d+:", ln):
        blocks.append((label, cur)); cur = Counter(); label = ln.split(":")[0]; continue
    t = ln.strip()
    if ln.startswith("\t") and t and not t.startswith((".", ";")):
        op = t.split()[0]; cur[cls(op)] += 1
        if op == "s_endpgm":
            blocks.append((label, cur))
            short = re.sub(r"_ZN4s2sr12_GLOBAL__N_1\d+", "", name).replace("EvNS_10ConvParamsE", "")
            if not want or any(w in short for w in want):
                tot = Counter()
                for _, c in blocks: tot.update(c)
                print(f"{short}: {sum(tot.values())} instr (~{sum(tot.values()) * 6 // 1024} KiB) {dict(tot)}")
                for lb, c in blocks:
                    if sum(c.values()) >= 40: print(f"    {lb:12s} {sum(c.values()):5d} {dict(c)}")
            name = None
