#!/usr/bin/env python3
"""Per-basic-block instruction counts of one kernel in build/rt_kernels.s (development aid).
usage: python tools/isa_blocks.py <mangled-name-substring> [file.s]"""
import re
import sys

name = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else "raytracing_amd/csrc/build/rt_kernels.s"
s = open(path).read()
m = re.search(r"^(\S*" + re.escape(name) + r"\S*):", s, re.M)
i = m.start()
j = s.index(".Lfunc_end", i)
blocks = []
cur = ["entry", 0, 0, 0, 0, 0]
for ln in s[i:j].split("\n"):
    t = ln.strip()
    mm = re.match(r"^(\.LBB\d+_\d+):", t)
    if mm:
        blocks.append(cur)
        cur = [mm.group(1), 0, 0, 0, 0, 0]
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        cur[1] += 1
        if "pk_fma" in op:
            cur[5] += 1
    elif op.startswith("s_"):
        cur[2] += 1
    elif op.startswith("ds_"):
        cur[3] += 1
    else:
        cur[4] += 1
blocks.append(cur)
tot = [sum(b[k] for b in blocks) for k in range(1, 6)]
for b in blocks:
    if b[1] + b[2] >= 8:
        print(f"{b[0]:12s} valu {b[1]:4d} (pk_fma {b[5]:3d}) salu {b[2]:4d} lds {b[3]:3d} other {b[4]:3d}")
print("total valu", tot[0], "salu", tot[1], "lds", tot[2], "other", tot[3])
