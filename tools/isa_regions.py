#!/usr/bin/env python3
"""Static instruction counts between the '; MARK name' comments (TREW_MARK, kernels/decide_group.inc) of one kernel of a
--save-temps .s file:  tools/isa_regions.py file.s mangled-kernel-prefix"""
import collections
import re
import sys

path, prefix = sys.argv[1], sys.argv[2]
cur, inside = None, False
order, cnt = [], collections.defaultdict(collections.Counter)
for line in open(path, errors="replace"):
    if line.startswith(prefix):
        inside, cur = True, "head"
        order.append(cur)
        continue
    if not inside:
        continue
    if "s_endpgm" in line:
        break
    m = re.search(r"; MARK (\w+)", line)
    if m:
        cur = m.group(1)
        if cur not in order:
            order.append(cur)
        continue
    t = line.strip().split()
    if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
        continue
    op = t[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "mem"
    cnt[cur][kind] += 1
for r in order:
    c = cnt[r]
    print("%-14s valu %5d  salu %5d  lds %4d  mem %4d" % (r, c["valu"], c["salu"], c["lds"], c["mem"]))
