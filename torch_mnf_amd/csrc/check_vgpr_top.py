"""Build gate for mnf_nsf_bwd_tile.hip (two-waves-per-SIMD shapes): its kernels keep their weight-gradient sums in hand-assigned
VECTOR registers at the top of the file, v[BASE] .. v255, which `amdgpu_num_vgpr(BASE / 2)` reserves (the compiler allocates below BASE only).

This script reads the device assembly and fails when, in a kernel named on the command line, an instruction OUTSIDE an
inline-asm block names a vector register at or above that kernel's BASE, when the kernel's register count is not 256
(the clobber that makes the hardware allocate the hand-assigned registers went missing), or when the kernel has more
than `max_scratch` scratch instructions (a few spills outside the tile loop are tolerated; a spilling loop shows as
dozens).

usage: python check_vgpr_top.py file.s max_scratch name_substring=BASE [name_substring=BASE ...]"""
import re
import sys


def main(path, max_scratch, limits):
    inside, kernel, base = False, None, None
    bad, scratch, counts, seen = [], {}, {}, set()
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        m = re.match(r"(_Z\w+):", t)
        if t.startswith(";;#ASMSTART"):
            inside = True
        elif t.startswith(";;#ASMEND"):
            inside = False
        elif m:
            kernel = m.group(1)
            base = next((b for k, b in limits.items() if k in kernel), None)
            if base is not None:
                seen.add(next(k for k in limits if k in kernel))
        elif t.startswith(".amdhsa_next_free_vgpr") and kernel and base is not None:
            counts[kernel] = int(t.split()[1])
        elif base is not None and not inside and t and t[0] not in ";.":
            code = t.split(";")[0]
            if code.startswith("scratch_"):
                scratch[kernel] = scratch.get(kernel, 0) + 1
            regs = [int(g) for r in re.finditer(r"\bv\[?(\d+)(?::(\d+))?\]?", code) for g in r.groups() if g is not None]
            if regs and max(regs) >= base:
                bad.append((n, kernel, code.strip()))
    for n, kernel, code in bad[:20]:
        print(f"{path}:{n}: {kernel}: compiler-generated `{code}` in the hand-assigned registers", file=sys.stderr)
    rc = 1 if bad else 0
    for k, c in scratch.items():
        if c > max_scratch:
            print(f"{path}: {k}: {c} scratch instructions (limit {max_scratch})", file=sys.stderr)
            rc = 1
    for k, c in counts.items():
        if c != 256:
            print(f"{path}: {k}: next_free_vgpr {c}, expected 256", file=sys.stderr)
            rc = 1
    for k in limits:
        if k not in seen:
            print(f"{path}: no kernel named *{k}*", file=sys.stderr)
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], int(sys.argv[2]), {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[3:]}))
