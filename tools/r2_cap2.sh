#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), round(d["ms_per_step"],3), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "frac", r["frac"])'
for cap in 100000 200000 262144 400000 1000000; do
PYRHE_CAPACITY=$cap timeout -k 10 200 python bench.py --steps 3 --no-cpu-baseline --no-alt --no-cross-check --sub-snps $cap 2>/dev/null | python -c "$fmt" "C3 cap=$cap" || exit 1
done
for cap in 50000 65536 80000; do
PYRHE_CAPACITY=$cap timeout -k 10 200 python bench.py --steps 3 --n-bin 8 --n-cov 10 --no-cpu-baseline --no-alt --no-cross-check --sub-snps $cap 2>/dev/null | python -c "$fmt" "C4 cap=$cap" || exit 1
done
for cap in 65536 131072 262144; do
PYRHE_CAPACITY=$cap timeout -k 10 200 python bench.py --steps 3 --genie G+GxE --no-cpu-baseline --no-alt --no-cross-check --sub-snps $cap 2>/dev/null | python -c "$fmt" "C5 cap=$cap" || exit 1
done
