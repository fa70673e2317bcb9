"""Instruction mix of the largest basic block (= the unrolled steady-state loop) of every kernel in a hipcc .s file.
usage: python profiles/tools/loop_mix.py file.s [name-filter]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):\s*; @\w+\n(.*?)^\.Lfunc_end", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    blocks = re.split(r"^\.LBB\d+_\d+:.*$", body, flags=re.M)
    big = max(blocks, key=lambda b: (b.count("v_mfma"), len(b)))
    ins = [l.split()[0] for l in big.splitlines() if l.strip() and not l.strip().startswith((";", "."))]
    c = collections.Counter(ins)
    print(name, "loop instructions", len(ins))
    print("   " + ", ".join(f"{k}:{v}" for k, v in c.most_common(28)))
