#!/usr/bin/env python
"""Instruction mix of the loops of one kernel in a hipcc -save-temps .s file.
Usage: tools/isa_loops.py file.s <mangled-name-prefix> [max_loops]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
names = [l.split(":")[0] for l in s.split("\n") if l.startswith(sys.argv[2]) and ":" in l]
name = names[0]
i = s.index(name + ":")
body = s[i:s.index(".Lfunc_end", i)]
lines = body.split("\n")
labels = {}
for k, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l.strip())
    if m:
        labels[m.group(1)] = k
loops = []
for k, l in enumerate(lines):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < k:
        loops.append((labels[m.group(1)], k))
print(name, "loops:", loops)
for a, b in loops[: int(sys.argv[3]) if len(sys.argv) > 3 else 3]:
    ins = [l.split()[0] for l in lines[a:b + 1] if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    cnt = collections.Counter(ins)
    cat = collections.Counter()
    for k, v in cnt.items():
        c = "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else "vmem" if k.startswith(("global", "buffer", "flat", "scratch")) else "other"
        cat[c] += v
    print("loop", a, b, "instrs", len(ins), dict(cat))
    print("  ", cnt.most_common(26))
