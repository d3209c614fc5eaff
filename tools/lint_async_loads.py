"""Build-time check of the inline-asm prefetches (rk_classify.hip: load_bases / bucket_load_async).

Those loads are issued through `asm volatile` so that hipcc does not count them (it would otherwise drain vmcnt inside
the hashing loop).  The price: hipcc does not know the destination VGPRs are in flight, so nothing stops its register
allocator from placing a copy (v_mov of a loop-carried value) or any other use of such a register between the load and
the `s_waitcnt vmcnt(0)` that retires it -- the copy would then read the register's OLD contents, depending on timing.

This script walks the control-flow graph of every kernel in a `hipcc -save-temps` .s file and reports each instruction
that reads or writes a VGPR while an asm-issued load into it may still be outstanding (forward may-analysis; only
`s_waitcnt vmcnt(0)` retires loads).  Exit status 1 when anything is found.

    python tools/lint_async_loads.py build/rk_classify-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import sys

REG1 = re.compile(r"\bv(\d+)\b")
REGN = re.compile(r"\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REGN.finditer(text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in REG1.finditer(text):
        out.add(int(m.group(1)))
    return out


def kernels(lines):
    """(symbol, body lines) of every function.  A body runs to the function's .Lfunc_end label, NOT to its first s_endpgm:
    hipcc may place basic blocks after the exit block."""
    cur, name = None, None
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if re.match(r"^\.Lfunc_end\d+:", l):
                if any("s_endpgm" in x for x in cur):
                    yield name, cur
                cur = None
                continue
            cur.append(l)
    if cur is not None and any("s_endpgm" in x for x in cur):   # text without the end label (the unit tests)
        yield name, cur


def analyse(name, body):
    # instruction list with flags
    ins = []      # (text, in_asm)
    lab_of = []
    labels = {}
    cur_label = "entry"
    in_asm = False
    for l in body:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            cur_label = m.group(1)
            continue
        c = t.split(";")[0].strip()
        if not c or c.startswith("."):
            continue
        ins.append((c, in_asm))
        lab_of.append(cur_label)
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (c, _) in enumerate(ins):
        op = c.split()[0]
        if op == "s_endpgm":
            continue
        if op == "s_branch":
            succ[i].append(labels[c.split()[1]])
            continue
        if op.startswith("s_cbranch"):
            succ[i].append(labels[c.split()[1]])
        if i + 1 < n:
            succ[i].append(i + 1)
    state_in = [None] * n
    state_in[0] = frozenset()
    work = [0]
    found = {}
    while work:
        i = work.pop()
        st = set(state_in[i])
        c, asm = ins[i]
        op = c.split()[0]
        used = regs_of(c)
        if asm and op.startswith("global_load"):
            dest = regs_of(c.split()[1].rstrip(","))
            src = used - dest
            if st & src:
                found[i] = (c, sorted(st & src))
            st |= dest
        elif op == "s_waitcnt" and "vmcnt(0)" in c:
            st.clear()
        else:
            if st & used and not dont_care_high_half(ins, succ, i, st):
                found[i] = (c, sorted(st & used))
        fs = frozenset(st)
        for j in succ[i]:
            if state_in[j] is None or not fs <= state_in[j]:
                state_in[j] = fs if state_in[j] is None else (state_in[j] | fs)
                work.append(j)
    res = []
    for i, (c, r) in sorted(found.items()):
        res.append((i, c, r, witness(ins, succ, lab_of, i, r)))
    return res


MAD = re.compile(r"v_mad_u64_u32 v\[(\d+):(\d+)\], (.*), v\[(\d+):(\d+)\]$")
MOV64 = re.compile(r"v_mov_b64_e32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]$")


def high_half_passthrough(c):
    """(dest high register, source high register, other registers) of an instruction that only forwards the high half of a
    64-bit source into the high half of its result: v_mad_u64_u32's addend, v_mov_b64's source."""
    m = MAD.match(c)
    if m:
        return int(m.group(2)), int(m.group(5)), regs_of(m.group(3)) | {int(m.group(1)), int(m.group(2)), int(m.group(4))}
    m = MOV64.match(c)
    if m:
        return int(m.group(2)), int(m.group(4)), {int(m.group(1)), int(m.group(2)), int(m.group(3))}
    return None


def value_is_dead(ins, succ, i, reg, depth=0):
    """True when the value instruction i leaves in VGPR `reg` is never used: on every path it is overwritten before it
    is read -- where a read as the (undefined) high half of another v_mad_u64_u32 addend only passes the question on to
    that instruction's own high result."""
    if depth > 6:
        return False
    seen, stack = set(), list(succ[i])
    while stack:
        j = stack.pop()
        if j in seen:
            continue
        seen.add(j)
        cj = ins[j][0]
        op = cj.split()[0]
        parts = cj.split(None, 1)
        if len(parts) == 2 and not op.startswith("s_"):
            ops = parts[1].split(", ")
            writes = op.startswith("v_") or op.startswith("ds_read") or op.startswith("global_load") or op.startswith("ds_bpermute")
            dest = regs_of(ops[0]) if writes else set()
            srcs = regs_of(", ".join(ops[1:])) if writes else regs_of(parts[1])
            if op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
                dest, srcs = set(), regs_of(parts[1])
            if reg in srcs:
                hp = high_half_passthrough(cj)
                if hp and hp[1] == reg and reg not in (hp[2] - {hp[0]}):
                    if not value_is_dead(ins, succ, j, hp[0], depth + 1):
                        return False
                    if reg in dest:
                        continue
                else:
                    return False
            if reg in dest:
                continue
        stack.extend(succ[j])
    return True


def dont_care_high_half(ins, succ, i, inflight):
    """hipcc computes a 32-bit multiply-add with `v_mad_u64_u32 v[d:d+1], a, b, v[x:x+1]` and leaves the addend's high half
    x+1 undefined -- whichever register follows x, possibly one in flight.  Harmless when the result's high half d+1 is
    dead (never read before it is overwritten, on any path)."""
    hp = high_half_passthrough(ins[i][0])
    if not hp:
        return False
    d_hi, x_hi, others = hp
    if (set(inflight) & regs_of(ins[i][0])) != {x_hi} or x_hi in others:
        return False
    return value_is_dead(ins, succ, i, d_hi)


def witness(ins, succ, lab_of, hit, regs):
    """Shortest wait-free path from an asm load of one of `regs` to instruction `hit`, as a list of block labels."""
    from collections import deque
    for ld, (c, asm) in enumerate(ins):
        if not (asm and c.split()[0].startswith("global_load")):
            continue
        if not (regs_of(c.split()[1].rstrip(",")) & set(regs)):
            continue
        prev = {ld: None}
        dq = deque([ld])
        while dq:
            i = dq.popleft()
            if i == hit and prev[i] is not None:
                break
            for j in succ[i]:
                cj = ins[j][0]
                if j in prev and j != hit:
                    continue
                if cj.startswith("s_waitcnt") and "vmcnt(0)" in cj:
                    continue
                if j not in prev or (j == hit and j == ld):
                    prev[j] = i
                    dq.append(j)
        if hit in prev and prev[hit] is not None:
            path, x, seen = [], hit, set()
            while x is not None and x not in seen:
                seen.add(x)
                path.append(x)
                x = prev[x]
            blocks = []
            for x in reversed(path):
                if not blocks or blocks[-1] != lab_of[x]:
                    blocks.append(lab_of[x])
            return "load #%d -> %s" % (ld, " ".join(blocks))
    return "?"


def asm_loads(body):
    """Number of global loads issued from inline asm in one kernel body (the loads hipcc does not track)."""
    n, in_asm = 0, False
    for l in body:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and t.startswith("global_load"):
            n += 1
    return n


def sgpr_hazards(body):
    """Asm-issued saddr-form loads (`global_load_* vD, vOff, s[a:b]`) whose base SGPRs were written by a VALU instruction
    (v_readlane / v_readfirstlane / v_cmp ... writing SGPRs) fewer than 5 wait states earlier.  gfx9: "VALU writes SGPR ->
    VMEM reads that SGPR" needs 5 wait states; hipcc's hazard recogniser does not look inside inline asm.  s_nop N counts N+1
    wait states, any other instruction 1.  Straight-line scan per basic block (a label resets the window: conservative enough,
    a branch target is never reached within 5 states of a v_readlane in this code, and a miss here only loses a diagnostic)."""
    res = []
    recent = []  # (sgpr numbers written by VALU, wait states since)
    in_asm = False
    for l in body:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if re.match(r"^\.LBB\d+_\d+:", t):
            recent = []
            continue
        c = t.split(";")[0].strip()
        if not c or c.startswith("."):
            continue
        op = c.split()[0]
        if in_asm and op.startswith("global_load") and re.search(r",\s*s\[(\d+):(\d+)\]", c):
            m = re.search(r",\s*s\[(\d+):(\d+)\]", c)
            base = set(range(int(m.group(1)), int(m.group(2)) + 1))
            for regs, age in recent:
                if regs & base and age < 5:
                    res.append((c, sorted(regs & base), age))
        states = 1
        m = re.match(r"^s_nop\s+(\d+)", c)
        if m:
            states = int(m.group(1)) + 1
        recent = [(r, a + states) for r, a in recent if a + states < 8]
        if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
            d = c.split()[1].rstrip(",")
            m = re.match(r"^s(\d+)$", d)
            if m:
                recent.append(({int(m.group(1))}, 0))
    return res


def check_file(lines, min_tile_kernels=0, out=print):
    """Returns the exit status: 0 clean, 1 hazards, 2 the input does not look like what this gate protects (fails CLOSED:
    no kernels, fewer k_classify_tile instantiations than expected, or a k_classify_tile without any asm-issued load --
    i.e. the symbol regex, the -save-temps layout or the ASMSTART markers stopped matching)."""
    bad = nk = ntile = nloads = 0
    blind = []
    for name, body in kernels(lines):
        nk += 1
        la = asm_loads(body)
        nloads += la
        if "k_classify_tile" in name:
            ntile += 1
            if la == 0:
                blind.append(name)
        sh = sgpr_hazards(body)
        if sh:
            bad += 1
            out("%s: %d asm load(s) reading an SGPR base too soon after a VALU wrote it" % (name[:70], len(sh)))
            for c, regs, age in sh[:8]:
                out("    %-50s s%s written %d wait state(s) earlier (needs 5)" % (c, ",s".join(str(r) for r in regs), age))
        res = analyse(name, body)
        if res:
            bad += 1
            out("%s: %d hazard(s)" % (name[:70], len(res)))
            for i, c, r, w in res[:8]:
                out("    #%d  %-50s in flight: %s   [%s]" % (i, c, ",".join("v%d" % x for x in r), w))
    out("%d kernels checked (%d k_classify_tile, %d asm-issued loads), %d finding(s): uses of an in-flight register / SGPR bases read too soon" % (nk, ntile, nloads, bad))
    if nk == 0:
        out("lint_async_loads: NO kernel found in the input -- refusing to pass (symbol regex or file layout changed?)")
        return 2
    if ntile < min_tile_kernels:
        out("lint_async_loads: %d k_classify_tile instantiations, expected at least %d -- refusing to pass" % (ntile, min_tile_kernels))
        return 2
    if blind:
        out("lint_async_loads: %d k_classify_tile kernel(s) without any asm-issued global_load (e.g. %s) -- the ASMSTART markers "
            "were not recognised or the prefetch is gone; refusing to pass" % (len(blind), blind[0][:70]))
        return 2
    return 1 if bad else 0


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("isa")
    ap.add_argument("--min-tile-kernels", type=int, default=0,
                    help="fail unless at least this many k_classify_tile instantiations were analysed")
    a = ap.parse_args()
    return check_file(open(a.isa).read().split("\n"), a.min_tile_kernels)


if __name__ == "__main__":
    sys.exit(main())
