"""Instruction-class histogram of one kernel (or of its hottest loop) in hipcc -S output.

usage: isa_count.py file.s mangled_substring [--loops]
Counts VALU / MFMA / VMEM / LDS / SALU per basic block; with --loops prints every block that is a backward-branch target
(label .LBBx_y that some later s_cbranch jumps back to) — the loop bodies.
"""
import re, sys, collections

def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"): return "mfma"
    if op.startswith("v_cvt_pk_bf16") or op.startswith("v_cvt"): return "valu_cvt"
    if op.startswith("v_exp") or op.startswith("v_rcp") or op.startswith("v_log") or op.startswith("v_rsq") or op.startswith("v_sqrt"): return "valu_trans"
    if op.startswith("v_accvgpr"): return "valu_acc"
    if op.startswith("v_"): return "valu"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and key in l and l.rstrip().endswith(":") is False and ":" in l:
            start = i; break
        if l.startswith("_Z") and key in l.split(":")[0]:
            start = i; break
    if start is None: sys.exit("kernel not found")
    end = start
    while end < len(lines) and not lines[end].startswith(".Lfunc_end"): end += 1
    body = lines[start:end]
    blocks = []; cur = ("entry", [])
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur); cur = (m.group(1), [])
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."): continue
        cur[1].append(t.split()[0])
        if t.startswith("s_cbranch") or t.startswith("s_branch"):
            cur[1][-1] = t.split()[0] + " " + t.split()[1]
    blocks.append(cur)
    order = {b[0]: i for i, b in enumerate(blocks)}
    total = collections.Counter()
    for name, ops in blocks:
        c = collections.Counter(classify(o.split()[0]) for o in ops)
        total.update(c)
        back = [o.split()[1] for o in ops if o.startswith("s_cbranch") and len(o.split()) > 1 and order.get(o.split()[1], 1 << 30) <= order[name]]
        if "--loops" in sys.argv and not back: continue
        if len(ops) < 20 and "--all" not in sys.argv: continue
        print(f"{name:12s} n={len(ops):5d} back->{back} ", dict(c))
        if "--ops" in sys.argv:
            oc = collections.Counter(o.split()[0] for o in ops)
            print("   ", oc.most_common(40))
    print("TOTAL", dict(total))

main()
