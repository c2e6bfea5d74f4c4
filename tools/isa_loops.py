#!/usr/bin/env python3
"""tools/isa_loops.py <file.s> <kernel-name-substring> -- development aid: the loops of one kernel of a gfx950 assembly listing
(backward branches), with the instruction mix of each loop body (VALU / VOP3P / LDS / VMEM / SALU / s_nop / s_waitcnt)."""
import re
import sys
from collections import Counter

src, want = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and want in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i


def cls(op):
    if op.startswith("v_pk_"): return "vop3p"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op == "s_nop": return "s_nop"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_"): return "salu"
    return "other"


for i, l in enumerate(body):
    m = re.match(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        a = labels[m.group(2)]
        c = Counter()
        nops = 0
        for x in body[a:i + 1]:
            t = x.strip().split()
            if not t or t[0].endswith(":") or t[0].startswith((";", ".")): continue
            c[cls(t[0])] += 1
            if t[0] == "s_nop": nops += int(t[1]) + 1
        tot = sum(c.values())
        print(f"loop {m.group(2)} lines {start + a + 1}-{start + i + 1}: {tot} instr  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + f"  nop_cycles={nops}")
