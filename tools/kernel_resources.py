#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch table from a `hipcc --cuda-device-only -S` file (the amdhsa metadata at its end).

usage: kernel_resources.py file.s [name-filter]
"""
import re
import subprocess
import sys


def kernels(path):
    text = open(path).read()
    meta = text[text.rfind("amdhsa.kernels:"):]
    out = []
    for block in re.split(r"\n  - (?=\.agpr_count)", "\n" + meta.split("amdhsa.kernels:", 1)[1])[1:]:
        f = dict(re.findall(r"\n    \.(\w+):\s+(\S+)", "\n    " + block))
        out.append(f)
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def main():
    ks = kernels(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    names = demangle([k["name"] for k in ks])
    print(f"{'vgpr':>5} {'agpr':>5} {'spill':>5} {'sgpr':>5} {'scratch':>7} {'lds':>7}  kernel")
    for k, n in zip(ks, names):
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        if flt and flt not in n:
            continue
        print(f"{k.get('vgpr_count', '?'):>5} {k.get('agpr_count', '0'):>5} {k.get('vgpr_spill_count', '?'):>5} "
              f"{k.get('sgpr_count', '?'):>5} {k.get('private_segment_fixed_size', '?'):>7} "
              f"{k.get('group_segment_fixed_size', '?'):>7}  {n}")


if __name__ == "__main__":
    main()
