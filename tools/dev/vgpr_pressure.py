#!/usr/bin/env python3
"""VGPR pressure along a kernel's assembly (development aid): backward liveness over the basic blocks of one function of a
hipcc -S listing, printed as the number of live vector registers at every N-th instruction together with the nearest landmark
(loads, stores, DPP shifts, branches).  Approximate: writes under a partial EXEC mask are taken as full definitions.

    python tools/dev/vgpr_pressure.py <file.s> <function-name-prefix> [step]
"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    if m:
        return [int(m.group(1))]
    return []


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    step = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    lines = open(path).read().split("\n")
    i = [k for k, l in enumerate(lines) if l.startswith(prefix) and ":" in l][0]
    j = [k for k, l in enumerate(lines) if l.startswith(".Lfunc_end") and k > i][0]
    ins = []  # (label or None, op, defs, uses, target)
    labels = {}
    for l in lines[i + 1:j]:
        s = l.split(";")[0].strip()
        if not s:
            continue
        if s.endswith(":"):
            labels[s[:-1]] = len(ins)
            continue
        if s.startswith("."):
            continue
        parts = s.replace(",", " ").split()
        op, toks = parts[0], parts[1:]
        rs = [regs(t.strip("|-").replace("|", "")) for t in toks]
        nodef = op.startswith(("global_store", "scratch_store", "ds_write", "v_cmp", "s_", "buffer_store", "v_readlane", "v_readfirstlane", "ds_bpermute_dummy")) or (op.startswith("global_atomic") and not (toks and regs(toks[0]) and len([r for r in rs if r]) >= 3 and "glc" in s or "sc0" in s))
        if op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
            defs, uses = [], [x for r in rs for x in r]
        elif nodef:
            defs, uses = [], [x for r in rs for x in r]
        else:
            defs = rs[0] if rs else []
            uses = [x for r in rs[1:] for x in r]
            if op.startswith("v_writelane") or "dpp" in op and "bound_ctrl" not in s or op.startswith(("v_mac", "v_fmac", "v_accvgpr")):
                uses = uses + defs
        tgt = None
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = toks[-1]
        ins.append((op, defs, uses, tgt, s))
    n = len(ins)
    succ = [[] for _ in range(n)]
    for k, (op, d, u, tgt, s) in enumerate(ins):
        if op == "s_endpgm":
            continue
        if op == "s_branch":
            if tgt in labels:
                succ[k].append(labels[tgt])
            continue
        if k + 1 < n:
            succ[k].append(k + 1)
        if tgt and tgt in labels:
            succ[k].append(labels[tgt])
    live_in = [frozenset()] * n
    changed = True
    it = 0
    while changed and it < 60:
        changed = False
        it += 1
        for k in range(n - 1, -1, -1):
            out = set()
            for t in succ[k]:
                if t < n:
                    out |= live_in[t]
            op, d, u, tgt, s = ins[k]
            new = frozenset((out - set(d)) | set(u))
            if new != live_in[k]:
                live_in[k] = new
                changed = True
    if len(sys.argv) > 4:  # dump: the live registers at instruction <at> with the instruction that last defined each (scanning backwards)
        at = int(sys.argv[4])
        for r in sorted(live_in[at]):
            q = at - 1
            while q >= 0 and r not in ins[q][1]:
                q -= 1
            print("v%-4d defined at %6d  %s" % (r, q, ins[q][4][:90] if q >= 0 else "?"))
        return
    peak = max(len(x) for x in live_in)
    print("instructions %d, peak live VGPRs %d (iterations %d)" % (n, peak, it))
    for k in range(0, n, step):
        blk = live_in[k:k + step]
        m = max(range(len(blk)), key=lambda q: len(blk[q]))
        marks = "".join(sorted(set(
            ("L" if ins[q][0].startswith("global_load") else "") + ("S" if ins[q][0].startswith("global_store") else "") +
            ("x" if "row_sh" in ins[q][4] else "") + ("A" if ins[q][0].startswith("global_atomic") else "") + ("d" if ins[q][0].startswith("ds_") else "") +
            ("p" if "scratch_" in ins[q][0] else "")
            for q in range(k, min(n, k + step)))))
        print("%6d  max %3d at %6d  %s" % (k, len(blk[m]), k + m, marks))


if __name__ == "__main__":
    main()
