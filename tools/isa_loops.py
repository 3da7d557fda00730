#!/usr/bin/env python
"""instruction census of the loops of one kernel in a hipcc -S listing: for every backward branch, the instruction mix of the range it
closes (MFMA / VALU / SALU / LDS / VMEM / waits).  usage: tools/isa_loops.py file.s <kernel substring> [min mfma]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
start = next(i for i, l in enumerate(s) if re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
body = s[start:end]
labels = {}
ins = []
for l in body:
    t = l.strip()
    m = re.match(r"^(\.LBB\S+):", t)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    ins.append(t.split(";")[0].strip())


def cls(i):
    op = i.split()[0]
    if "mfma" in op:
        return "mfma"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


print(f"kernel {key}: {len(ins)} instructions")
for idx, i in enumerate(ins):
    m = re.match(r"s_cbranch\S*\s+(\.LBB\S+)|s_branch\s+(\.LBB\S+)", i)
    if not m:
        continue
    tgt = labels.get(m.group(1) or m.group(2))
    if tgt is None or tgt > idx:
        continue
    c = Counter(cls(x) for x in ins[tgt:idx + 1])
    if c["mfma"] < min_mfma:
        continue
    print(f"loop [{tgt}, {idx}] {idx - tgt + 1} instr: " + "  ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
    if "-v" in sys.argv:
        ops = Counter(x.split()[0] for x in ins[tgt:idx + 1] if cls(x) in ("valu", "salu"))
        print("   ", ", ".join(f"{k} {v}" for k, v in ops.most_common(25)))
