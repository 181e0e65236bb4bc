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


def _tokens(t):
    return re.findall(r"(?<![0-9a-zA-Z_])(?:[av]\[\d+:\d+\]|[av]\d+)(?![0-9a-zA-Z_])", t)


def check_cfg(path):
    """The rule that can live in the test suite (r04's fault, profiles/r04_latency_anatomy.txt section 3): from every hidden
    (inline-asm, register-destination) global load follow EVERY path of the kernel's control-flow graph -- fall-through, both sides
    of each s_cbranch, the target of each s_branch, wherever the blocks sit in the file -- up to the first `s_waitcnt` on that
    path that names vmcnt at all; no instruction on the way may touch the load's destination registers.  Nothing is counted, so
    an infeasible path cannot make a counted wait look too weak (the false positives `--branches` can give): what is reported is
    a register handed to something else while a load into it is certainly still allowed to be in flight.  r04: the bias requests sat
    in front of the loader wave's role branch; on the loader's side their destinations were dead, the compiler reused them for
    DMA offsets and the landing loads overwrote those."""
    kernels, cur = [], None
    in_asm = False
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith("_Z") and ":" in t and not t.startswith("_ZZ"):
            cur = {"name": t.split(":")[0], "ins": [], "labels": {}}
            kernels.append(cur)
            in_asm = False
            continue
        if cur is None:
            continue
        lab = re.fullmatch(r"(\.LBB\d+_\d+):.*", t)
        if lab:
            cur["labels"][lab.group(1)] = len(cur["ins"])
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur["ins"].append((ln, t.split(";")[0].strip(), in_asm))
    bad = 0
    for k in kernels:
        ins, labels = k["ins"], k["labels"]

        def hidden(i):
            _, t, a = ins[i]
            return a and t.startswith("global_load_dword") and "lds" not in t

        def succ(i):
            t = ins[i][1]
            m = re.match(r"s_branch\s+(\.LBB\d+_\d+)", t)
            if m:
                return [labels[m.group(1)]] if m.group(1) in labels else []
            if t.startswith("s_endpgm") or t.startswith("s_setpc") or t.startswith("s_swappc"):
                return []
            out = [i + 1] if i + 1 < len(ins) else []
            m = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", t)
            if m and m.group(1) in labels:
                out.append(labels[m.group(1)])
            return out

        seen_reports = set()
        for i0 in range(len(ins)):
            if not hidden(i0):
                continue
            toks = _tokens(ins[i0][1])
            dest = regs(toks[0]) if toks else set()
            if not dest:
                continue
            stack, seen = list(succ(i0)), set()
            while stack:
                i = stack.pop()
                if i in seen or i >= len(ins):
                    continue
                seen.add(i)
                ln, t, _ = ins[i]
                if "s_waitcnt" in t and "vmcnt(" in t:
                    continue                         # the first vmcnt wait of this path: the counted rule (check) takes over from here
                if not hidden(i):
                    used = set()
                    for tok in _tokens(t):
                        used |= regs(tok)
                    hit = used & dest
                    if hit and (ln, ins[i0][0]) not in seen_reports:
                        seen_reports.add((ln, ins[i0][0]))
                        bad += 1
                        print(f"{path}:{ln}: {k['name']}: `{t}` touches {sorted(hit)} on a path from the hidden load at line {ins[i0][0]} "
                              f"(`{ins[i0][1]}`) with no vmcnt wait in between")
                        continue
                stack.extend(succ(i))
    return bad


if __name__ == "__main__":
    # --branches: also carry the loads in flight at a forward branch to its target block (finds a hidden load whose destination
    # is dead on a side path, e.g. a role branch -- but every statically possible path counts, feasible or not: the counted waits
    # of the ring are only right on the feasible ones, so this mode is for reading its report, not for the test suite)
    # --cfg: additionally the control-flow rule (check_cfg): every path from a hidden load to its first vmcnt wait; no counting, so
    # it is the one the test suite runs (tests/test_abi_cpu.py)
    br = "--branches" in sys.argv
    cfg = "--cfg" in sys.argv
    files = [p for p in sys.argv[1:] if not p.startswith("--")]
    n = sum(check(p, br) for p in files) + (sum(check_cfg(p) for p in files) if cfg else 0)
    print(f"{n} hazard(s)")
    sys.exit(1 if n else 0)
