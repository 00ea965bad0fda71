#!/usr/bin/env python3
"""Instruction-class timeline of a kernel's ISA (from tools/kernel_isa.sh): M mfma, v vector ALU, t transcendental, L LDS read,
l LDS write/atomic, G global/buffer load, g store, D LDS-DMA, w s_waitcnt, B barrier, . scalar / other.
usage: isa_timeline.py kernel.s [first_line last_line]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(lines)
out = []
for l in lines[lo - 1:hi]:
    l = l.split(";")[0].strip()
    if not l or l.endswith(":") or l.startswith("."):
        if l.endswith(":"):
            out.append("|")
        continue
    op = l.split()[0]
    if "mfma" in op: c = "M"
    elif op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): c = "t"
    elif op.startswith("v_"): c = "v"
    elif op.startswith(("ds_read", "ds_load", "ds_bpermute")): c = "L"
    elif op.startswith("ds_"): c = "l"
    elif " lds" in l and op.startswith(("buffer_load", "global_load")): c = "D"
    elif op.startswith(("global_load", "buffer_load", "scratch_load")): c = "G"
    elif op.startswith(("global_store", "buffer_store", "scratch_store", "global_atomic")): c = "g"
    elif op.startswith("s_waitcnt"): c = "w"
    elif op.startswith("s_barrier"): c = "B"
    else: c = "."
    out.append(c)
s = "".join(out)
for i in range(0, len(s), 120):
    print(s[i:i + 120])
