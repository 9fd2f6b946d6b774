"""Instruction histogram of the blocks hipcc marks as belonging to a loop ('in Loop: Header=...' / 'Loop Header') of one kernel.
usage: loop_count.py file.s mangled_substring"""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
i0 = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
inloop = False
ops = collections.Counter()
for l in lines[i0:i1]:
    if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l):
        inloop = ("in Loop" in l) or ("Loop Header" in l)
        continue
    t = l.strip()
    if not inloop or not t or t.startswith(";") or t.startswith("."):
        continue
    ops[t.split()[0]] += 1
def cls(o):
    if o.startswith("v_mfma"): return "mfma"
    if o.startswith("v_"): return "valu"
    if o.startswith(("buffer_", "global_", "scratch_", "flat_")): return "vmem"
    if o.startswith("ds_"): return "lds"
    if o.startswith("s_nop"): return "nop"
    if o.startswith("s_waitcnt"): return "wait"
    return "salu"
c = collections.Counter()
for o, n in ops.items(): c[cls(o)] += n
print("total", sum(ops.values()), dict(c))
print(ops.most_common(50))
