#!/bin/bash
# N = 50k: fewer sample splits of GEMM 1 (one round of workgroups, no atomics)
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), "M SNPs/s", round(d["ms_per_step"],3), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"])'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-cross-check --n-indv 50000 --n-snp 500000"
for o in "" "--opt g1_workgroups=500" "--opt g1_workgroups=1000" "--opt g1_workgroups=500 --opt g1_fill=0" "--opt g1_workgroups=1000 --opt g1_fill=0" ""; do
python bench.py $B $o 2>/dev/null | tail -1 | python -c "$fmt" "C2 [$o]"
done
