#!/usr/bin/env python3
"""Registers / spills of the kernels in the built library: tools/vgprs.py [substring of the demangled name]."""
import re, subprocess, sys, tempfile, os
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pyrhe_amd", "csrc", "libpyrhe_sketch.so")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
    d = open(fat, "rb").read()
    open(co, "wb").write(d[d.find(b"\x7fELF"):])
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
rows = []
for b in out.split("- .agpr_count")[1:]:
    nm = re.search(r"\.name:\s+(\S+)", b).group(1)
    dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
    if pat not in dem:
        continue
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", b).group(1)
    rows.append((dem.split("(")[0], g("vgpr_count"), b.split("\n")[0].strip(": "), g("vgpr_spill_count"), g("private_segment_fixed_size")))
for r in sorted(rows):
    print(f"{r[0]:90s} vgpr {r[1]:>4s} agpr {r[2]:>4s} spill {r[3]} scratch {r[4]}")
