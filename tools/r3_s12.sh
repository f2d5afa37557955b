#!/bin/bash
# config 2 (N = 50k): entries per engine call (PYRHE_CAPACITY: 26 / 52 / 104 blocks per call)
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,1), "M SNPs/s", round(d["ms_per_step"],3), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"])'
for c in 131072 262144 524288; do
PYRHE_CAPACITY=$c python bench.py --n-indv 50000 --n-snp 500000 --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | tail -1 | python -c "$fmt" "C2 capacity $c"
done
for c in 131072 262144; do
PYRHE_CAPACITY=$c python bench.py --n-indv 100000 --n-snp 500000 --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | tail -1 | python -c "$fmt" "N=100k capacity $c"
done
