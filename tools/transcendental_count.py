#!/usr/bin/env python3
"""Transcendental instructions per spline element of an NSF_CL kernel, read off its device assembly: the loops of the
kernel whose bodies hold v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 are its slot loops (one trip = one
element per lane), so the instructions of those loop bodies are the per-element count.  With the guide's issue cost of 8
cycles per transcendental wave-instruction (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost') this gives the floor no
instruction diet of the plain arithmetic can go below.

usage: transcendental_count.py file.s mangled-kernel-name-prefix   -> prints JSON {"per_element": N, "valu_per_element": M, ...}"""
import json
import re
import sys

TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32", "v_exp_legacy", "v_log_legacy")


def loops(lines):
    """(first, last) line ranges of the innermost backward-branch loops: label .LBBx_y ... branch to it"""
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l)}
    out = []
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return out


def main(path, prefix):
    text = open(path).read().split("\n")
    start = next(i for i, l in enumerate(text) if l.startswith(prefix) and ":" in l)
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    body = [l.strip() for l in text[start:end]]
    found = []
    for a, b in loops(body):
        seg = body[a:b + 1]
        n_t = sum(1 for l in seg if l.startswith(TRANS))
        if n_t == 0:
            continue
        # innermost only: no other transcendental loop nested inside
        n_v = sum(1 for l in seg if l.startswith("v_") and not l.startswith("v_mfma"))
        found.append({"lines": [a, b], "transcendental": n_t, "valu": n_v, "mfma": sum(1 for l in seg if l.startswith("v_mfma"))})
    inner = [f for f in found if not any(g is not f and g["lines"][0] >= f["lines"][0] and g["lines"][1] <= f["lines"][1] for g in found)]
    per_el = max((f["transcendental"] for f in inner), default=0)
    best = next((f for f in inner if f["transcendental"] == per_el), None)
    print(json.dumps({"kernel": prefix, "slot_loops": inner, "per_element": per_el, "valu_per_element": best["valu"] if best else 0,
                      "cycles_per_transcendental": 8}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
