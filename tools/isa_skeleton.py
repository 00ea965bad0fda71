#!/usr/bin/env python3
"""Prints one kernel's vector-memory operations, waits, barriers and branches (MFMAs as counts) from a hipcc -S file.
usage: isa_skeleton.py file.s <mangled-name-prefix>"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ":" in l][0]
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
mf = va = ds = 0
for i in range(start, end):
    l = lines[i]
    if "v_mfma" in l:
        mf += 1
        continue
    if re.match(r"\s+ds_", l):
        ds += 1
        continue
    if re.search(r"global_|buffer_|scratch_|vmcnt|s_barrier|s_cbranch|^\.LBB|s_endpgm", l):
        if mf or va or ds:
            print(f"        [{mf} mfma, {ds} ds, {va} other]")
            mf = va = ds = 0
        print(i + 1, l.strip()[:110])
    elif re.match(r"\s+[vs]_", l):
        va += 1
