"""Build gate for the kernels that keep rows in hand-assigned accumulator registers (mnf_agpr.h).

A clobber list marks a0..aN as part of the kernel's register budget, but it does not stop the register allocator from
parking its own values there between two asm statements, which would overwrite rows that are still in use.  This
script reads the device assembly of such a file and fails when any instruction OUTSIDE an inline-asm block names an
accumulator register, or when the kernel touches scratch memory (a scratch access would join the vector-memory
queue the kernel's hand-counted s_waitcnt vmcnt(N) values describe).

usage: python check_agpr.py file.s"""
import re
import sys


def main(path):
    inside, bad, kernel = False, [], None
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            inside = True
        elif t.startswith(";;#ASMEND"):
            inside = False
        elif t.endswith(":") and t.startswith("_Z"):
            kernel = t[:-1]
        elif not inside and t and t[0] not in ";.":
            code = t.split(";")[0]
            if re.search(r"\ba\[?\d", code) or code.startswith("scratch_"):
                bad.append((n, kernel, code.strip()))
    for n, kernel, code in bad[:20]:
        print(f"{path}:{n}: {kernel}: compiler-generated `{code}`", file=sys.stderr)
    if bad:
        print(f"{path}: {len(bad)} compiler-generated accumulator-register / scratch accesses: the hand-managed "
              "registers are not safe", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
