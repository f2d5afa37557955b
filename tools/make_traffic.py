#!/usr/bin/env python3
"""profiles/r01_final_pmc_summary.txt -> profiles/r01_traffic.json: HBM bytes per algorithmic byte of the four GEMM
kernels (2 * FETCH_SIZE + WRITE_SIZE in units of 1024 B, summed over the launches of the PMC run, divided by the packed
genotype bytes those launches covered: bench.py --n-snp 100000 --jn 10 --steps 1 --warmup 1, i.e. 2 passes per mode)."""
import json, re, sys
txt = open("profiles/r01_final_pmc_summary.txt").read()
N, SNPS, PASSES = 500_000, 100_000, 2
alg = SNPS * ((N + 3) // 4) * PASSES


def grab(section, kern, counter):
    sec = txt.split("== " + section)[1].split("== ")[0]
    blk = sec.split(kern)[1]
    m = re.search(counter + r"\s+n=\s*(\d+)\s+mean=([0-9.e+]+)", blk)
    return int(m.group(1)), float(m.group(2))


names = {"gemm1_i8_kernel": "void gemm1_i8_kernel<3, false, 0, true>", "gemm2_i8p_kernel": "void gemm2_i8p_kernel<1, false, true>",
         "gemm2_kernel_f16x2": "_Z12gemm2_kernelILi1ELi2ELi1ELb1ELb0ELi0",
         "gemm1_kernel_f16": "_Z12gemm1_kernelILi1ELi1ELb0ELb0ELi0ELb1", "gemm2_kernel_f16": "_Z12gemm2_kernelILi1ELi1ELi2ELb0ELb0ELi0"}
out = {"_comment": __doc__.strip(), "n_indv": N, "block_snps": 10000, "kernels": {}}
try:
    prev = json.load(open("profiles/r01_traffic.json"))["kernels"]
except Exception:
    prev = {}
for k, n in names.items():
    try:
        nf, f = grab("pmc_fetch", n, "FETCH_SIZE")
        nw, w = grab("pmc_write", n, "WRITE_SIZE")
    except Exception:
        # not launched by the default bench command any more (the fp16 hi+lo GEMM 2 is the fallback for > 10 sketch
        # columns): keep the figure of the earlier pass of this round
        if k in prev:
            out["kernels"][k] = dict(prev[k], carried_over="earlier PMC pass of this round")
        continue
    total = 2 * f * 1024 * nf + w * 1024 * nw
    out["kernels"][k] = {"launches": nf, "fetch_size_mean": f, "write_size_mean": w, "bytes_per_algorithmic_byte": round(total / alg, 4)}
json.dump(out, open("profiles/r01_traffic.json", "w"), indent=2)
print(json.dumps(out["kernels"], indent=1))
