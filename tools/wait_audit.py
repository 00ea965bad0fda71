#!/usr/bin/env python3
"""Lists, per kernel of a `hipcc -S` device file, the full drains (`s_waitcnt vmcnt(0)`) that sit inside loops.

DESIGN.md section 5a: hipcc's vmcnt(N) is exact only while every vector-memory operation between a load and its use is
issued unconditionally; behind a load or store under a branch the next use of ANY loaded register waits vmcnt(0), i.e.
for everything the wave has in flight, prefetches and its own stores included.  This prints where that happened.

usage: wait_audit.py file.gfx950.s [kernel-name-substring]
"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), text=True,
                             capture_output=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    kernels, cur, depth = [], None, 0
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z[\w$.]+):", line)
        if m and "@function" not in line:
            cur = {"name": m.group(1), "loop0": [], "all0": 0, "vmem_loop": 0, "lines": 0}
            kernels.append(cur)
            depth = 0
            continue
        if cur is None:
            continue
        if "s_endpgm" in line:
            cur = None
            continue
        b = re.match(r"^(\.LBB\d+_\d+|; %bb\.\d+):", line)
        if b:
            d = re.search(r"Depth=(\d+)", line)
            depth = int(d.group(1)) if d else 0
            continue
        if re.search(r"\b(global|buffer|flat|scratch)_(load|store|atomic)", line) and depth:
            cur["vmem_loop"] += 1
        if re.search(r"s_waitcnt\b.*vmcnt\(0\)", line):
            cur["all0"] += 1
            if depth:
                cur["loop0"].append((ln, depth))
    names = demangle([k["name"] for k in kernels])
    for k in kernels:
        nm = names[k["name"]]
        if pat and pat not in nm:
            continue
        where = " ".join(f"{ln}(d{d})" for ln, d in k["loop0"][:12])
        print(f"{len(k['loop0']):4d} in loops / {k['all0']:4d} total, {k['vmem_loop']:4d} vmem ops in loops  "
              f"{nm[:70]}\n       {where}")


if __name__ == "__main__":
    main()
