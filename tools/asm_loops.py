#!/usr/bin/env python3
"""Static instruction counts of the innermost loops of a kernel in a `hipcc -S` listing (the RK4 step of the line
search and the RK4-with-sensitivities step of the linearisation are the two big ones of fused_sqp_kernel).
usage: asm_loops.py dev.s <mangled-kernel-substring> [min_size]"""
import collections
import re
import sys


def body(lines, sub):
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l and sub in l.split(":")[0])
    out = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        out.append(l)
    return lines[start].split(":")[0], out


def main():
    lines = open(sys.argv[1]).read().split("\n")
    name, b = body(lines, sys.argv[2])
    min_size = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    lab = {m.group(1): i for i, l in enumerate(b) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(b):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < i:
            loops.append((lab[m.group(1)], i))
    loops = sorted(set(loops))
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] and (o[1] - o[0]) >= min_size for o in loops)]
    print(name)
    for lo, hi in inner:
        if hi - lo < min_size:
            continue
        ops = collections.Counter()
        for l in b[lo:hi + 1]:
            m = re.match(r"\s+([a-z_0-9]+)\s", l)
            if m:
                ops[m.group(1)] += 1
        valu = sum(v for k, v in ops.items() if k.startswith("v_"))
        trans = sum(v for k, v in ops.items() if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", k))
        print("  loop at +%d..+%d: %d instructions, %d VALU (%d transcendental), %d LDS, %d scalar"
              % (lo, hi, sum(ops.values()), valu, trans, sum(v for k, v in ops.items() if k.startswith("ds_")),
                 sum(v for k, v in ops.items() if k.startswith("s_"))))


if __name__ == "__main__":
    main()
