#!/usr/bin/env python3
"""Static check of the conv kernels' hidden (inline-asm) global loads.

The epilogue's residual loads are inline asm so that hipcc's waitcnt pass does not drain the LDS-DMA
ring (conv3x3.hip, asm_load16).  The price: the compiler does not know their destination registers are
in flight.  This script reads the device assembly and reports every instruction that touches a
destination register of such a load between the load and the next `s_waitcnt vmcnt(0)` -- the pattern
behind r01's fault in the 4-wave variant (a copy of a not-yet-arrived register, then reuse of the
register as an address).

    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only conv3x3.hip -o conv.s
    tools/check_asm_loads.py conv.s
"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"([av])\[(\d+):(\d+)\]", tok)
    if m:
        return {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([av])(\d+)", tok)
    return {f"{m.group(1)}{m.group(2)}"} if m else set()


def check(path, branches=False):
    """A hidden load is complete once an `s_waitcnt vmcnt(N)` has executed with at most N vector-memory operations
    issued after it (vmcnt completes in issue order); until then nothing may touch its destination registers."""
    bad = 0
    kernel = None
    mfma_at = {}          # AGPR -> wait-state position of the inline-asm MFMA that wrote it last (hipcc does not know these
                          # asm statements are MFMAs and inserts no hazard nops: a VALU / memory read of the result needs 18
                          # wait states behind a 16-pass MFMA; dependent MFMAs on the same accumulator are interlocked)
    pipe_free = 0
    ws = 0                # wait states so far: 1 per instruction, N + 1 per s_nop N (straight-line count)
    pending = {}          # reg -> (line of the load, index of the load in the vector-memory stream)
    nvm = 0               # vector-memory operations issued so far in this kernel (straight-line count; loops only
                          # make the real distance larger or equal within one trip)
    in_asm = False
    at_branch = {}        # label -> [(pending at a forward branch to it, nvm there)]: a load in flight at a branch is in flight at
                          # its target too, wherever the block sits in the file (r04: a loader-wave branch taken with bias loads
                          # pending reused their destination registers; the straight-line scan had passed a wait by then)
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith("_Z") and ":" in t and not t.startswith("_ZZ"):
            kernel, pending, nvm, mfma_at, ws, pipe_free, at_branch = t.split(":")[0], {}, 0, {}, 0, 0, {}
        lab = re.fullmatch(r"(\.LBB\d+_\d+):", t)
        if lab:
            for pend, n_at in at_branch.pop(lab.group(1), []):
                for r, (l0, idx) in pend.items():
                    pending.setdefault(r, (l0, nvm - (n_at - idx)))      # as many operations behind the load as on that path
            continue
        br = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
        if br and pending and branches:
            at_branch.setdefault(br.group(1), []).append((dict(pending), nvm))
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        toks = re.findall(r"(?<![0-9a-zA-Z_])(?:[av]\[\d+:\d+\]|[av]\d+)(?![0-9a-zA-Z_])", t)
        mn = re.match(r"s_nop\s+(\d+)", t)
        if t.startswith("v_mfma"):
            # the matrix pipe takes one MFMA per `passes` wait states: an MFMA behind another issues when the pipe is free
            passes = 16 if "32x32x64" in t else 8
            ws = max(ws + 1, pipe_free)
            pipe_free = ws + passes
            if in_asm and toks:
                for r in regs(toks[0]):
                    if r.startswith("a"):
                        mfma_at[r] = (ws, passes + 3)      # LLVM GFX940_XDL_N_PassWriteVgprVALUMemExpReadWaitStates
        else:
            ws += int(mn.group(1)) + 1 if mn else 1
            if mfma_at and not t.startswith("s_"):
                used = set()
                for tok in toks:
                    used |= regs(tok)
                for r in sorted(used & set(mfma_at)):
                    at, need = mfma_at.pop(r)
                    if ws - at < need:
                        bad += 1
                        print(f"{path}:{ln}: {kernel}: `{t}` reads {r} {ws - at} wait state(s) behind the inline MFMA that wrote it (need {need})")
        is_vm = bool(re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", t))
        if in_asm and t.startswith("global_load_dword") and "lds" not in t:
            nvm += 1
            for r in (regs(toks[0]) if toks else set()):
                pending[r] = (ln, nvm)
            continue
        m = re.search(r"vmcnt\((\d+)\)", t) if "s_waitcnt" in t else None
        if m:
            n = int(m.group(1))
            pending = {r: v for r, v in pending.items() if nvm - v[1] < n}      # still among the n youngest
            continue
        if pending:
            used = set()
            for tok in toks:
                used |= regs(tok)
            hit = used & set(pending)
            if hit:
                bad += 1
                print(f"{path}:{ln}: {kernel}: `{t}` touches {sorted(hit)} loaded at line {min(pending[r][0] for r in hit)} before its vmcnt")
        if is_vm:
            nvm += 1
    return bad


if __name__ == "__main__":
    # --branches: also carry the loads in flight at a forward branch to its target block (finds a hidden load whose destination
    # is dead on a side path, e.g. a role branch -- but every statically possible path counts, feasible or not: the counted waits
    # of the ring are only right on the feasible ones, so this mode is for reading its report, not for the test suite)
    br = "--branches" in sys.argv
    n = sum(check(p, br) for p in sys.argv[1:] if p != "--branches")
    print(f"{n} hazard(s)")
    sys.exit(1 if n else 0)
