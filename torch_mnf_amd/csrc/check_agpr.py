"""Build gate for the kernels that keep rows in hand-assigned accumulator registers (mnf_agpr.h).

A clobber list marks a0..aN as part of the kernel's register budget, but it does not stop the register allocator from
parking its own values there between two asm statements, which would overwrite rows that are still in use.  This
script reads the device assembly of such a file and fails when any instruction OUTSIDE an inline-asm block names an
accumulator register, or when the kernel touches scratch memory (a scratch access would join the vector-memory
queue the kernel's hand-counted s_waitcnt vmcnt(N) values describe).

With a second argument N the compiler may use a0 .. a(N-1) itself (a kernel whose own values overflow the 256 vector
registers: it allocates accumulator registers from a0 upwards, the hand-managed ones then sit at the top of the file).

usage: python check_agpr.py file.s [N]"""
import re
import sys


def main(path, allowed=0):
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
            regs = [int(g) for m in re.finditer(r"\ba\[?(\d+)(?::(\d+))?", code) for g in m.groups() if g is not None]
            if (regs and max(regs) >= allowed) or code.startswith("scratch_"):
                bad.append((n, kernel, code.strip()))
    for n, kernel, code in bad[:20]:
        print(f"{path}:{n}: {kernel}: compiler-generated `{code}`", file=sys.stderr)
    if bad:
        print(f"{path}: {len(bad)} compiler-generated accumulator-register / scratch accesses: the hand-managed "
              "registers are not safe", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0))
