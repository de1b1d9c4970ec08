#!/usr/bin/env python3
"""Static instruction mix per kernel of a gfx950 assembly file (hipcc -S --cuda-device-only): python tools/isa_mix.py file.s [name-filter]"""
import collections
import re
import sys

s = open(sys.argv[1]).read().splitlines()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur, ops = None, None
for line in s:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur, ops = m.group(1), collections.Counter()
        continue
    if cur is None:
        continue
    t = line.strip()
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if op.startswith("global_load_lds"): ops["dma"] += 1
    elif op.startswith(("global_load", "flat_load", "buffer_load")): ops["vmem_ld"] += 1
    elif op.startswith("scratch_"): ops["scratch"] += 1
    elif op.startswith(("global_store", "flat_store", "global_atomic")): ops["vmem_st"] += 1
    elif op.startswith("ds_"): ops["lds"] += 1
    elif op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): ops["lane"] += 1
    elif op.startswith("v_"): ops["valu"] += 1
    elif op.startswith("s_waitcnt"): ops["wait"] += 1
    elif op.startswith("s_nop"): ops["nop"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): ops["branch"] += 1
    elif op.startswith("s_"): ops["salu"] += 1
    if op == "s_endpgm":
        if flt in cur:
            print(cur, sum(ops.values()), dict(ops))
        cur = None
