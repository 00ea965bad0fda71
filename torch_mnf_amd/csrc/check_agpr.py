"""Build gate for the kernels that keep rows in hand-assigned accumulator registers (mnf_agpr.h).

A clobber list marks a0..aN as part of the kernel's register budget, but it does not stop the register allocator from
parking its own values there between two asm statements, which would overwrite rows that are still in use.  This
script reads the device assembly of such a file and fails when any instruction OUTSIDE an inline-asm block names an
accumulator register, or when the kernel touches scratch memory (a scratch access would join the vector-memory
queue the kernel's hand-counted s_waitcnt vmcnt(N) values describe).

With a second argument N the compiler may use a0 .. a(N-1) itself (a kernel whose own values overflow the 256 vector
registers: it allocates accumulator registers from a0 upwards, the hand-managed ones then sit at the top of the file).

Arguments of the form name=N check only the kernels whose (mangled) names contain `name`, each with its own N (a file
whose kernels hand-assign different numbers of registers, or that also holds kernels with another register scheme).

usage: python check_agpr.py file.s [N | name=N ...]"""
import re
import sys


def main(path, allowed=0, limits=None):
    inside, bad, kernel, seen = False, [], None, set()
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            inside = True
        elif t.startswith(";;#ASMEND"):
            inside = False
        elif re.match(r"_Z\w+:", t):
            kernel = t.split(":")[0]
            if limits is not None:
                key = next((k for k in limits if k in kernel), None)
                allowed = limits[key] if key is not None else None
                if key is not None:
                    seen.add(key)
        elif limits is not None and (kernel is None or allowed is None):
            continue
        elif not inside and t and t[0] not in ";.":
            code = t.split(";")[0]
            regs = [int(g) for m in re.finditer(r"\ba\[?(\d+)(?::(\d+))?", code) for g in m.groups() if g is not None]
            if (regs and max(regs) >= allowed) or code.startswith("scratch_"):
                bad.append((n, kernel, code.strip()))
    for n, kernel, code in bad[:20]:
        print(f"{path}:{n}: {kernel}: compiler-generated `{code}`", file=sys.stderr)
    for k in (limits or {}):
        if k not in seen:
            print(f"{path}: no kernel named *{k}*", file=sys.stderr)
            return 1
    if bad:
        print(f"{path}: {len(bad)} compiler-generated accumulator-register / scratch accesses: the hand-managed "
              "registers are not safe", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    args = sys.argv[2:]
    if args and "=" in args[0]:
        sys.exit(main(sys.argv[1], 0, {a.split("=")[0]: int(a.split("=")[1]) for a in args}))
    sys.exit(main(sys.argv[1], int(args[0]) if args else 0))
