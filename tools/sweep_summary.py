#!/usr/bin/env python3
"""Summarise tools/verify_sweep.sh output (pairs of lines, i8x3 then f16x2): largest relative difference of the variance
components (relative to the largest component) and of their jackknife SEs per shape."""
import sys

lines = [l.strip() for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sweep.log") if "/" in l and "[" in l]


def parse(l):
    name, rest = l.split(" [", 1)
    a, b = rest.rsplit("] [", 1)
    return name, [float(x) for x in a.split(",")], [float(x) for x in b.rstrip("]").split(",")]


for a, b in zip(lines[0::2], lines[1::2]):
    na, sa, ea = parse(a)
    _, sb, eb = parse(b)
    ds = max(abs(x - y) for x, y in zip(sa, sb)) / max(abs(x) for x in sb)
    de = max(abs(x - y) / abs(y) for x, y in zip(ea, eb))
    print(f"{na.rsplit('/', 1)[0]:34s} sigma^2 {ds:.1e}   SE {de:.1e}")
print(len(lines) // 2, "shapes")
