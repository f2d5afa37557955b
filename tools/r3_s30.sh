#!/bin/bash
# cohort-size sweep: pass_frac and kernels at sizes between the BASELINE configurations (no cliff in the heuristics?)
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), "M SNPs/s", round(d["ms_per_step"],3), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "pass_frac", r["pass_frac"], r["kernels"]["gemm1"], r["kernels"]["gemm2"])'
B="--steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check"
for nm in "10000 1000000" "20000 1000000" "30000 800000" "65000 500000" "70000 500000" "100000 400000" "150000 400000" "190000 300000" "200000 300000" "300000 400000" "400000 500000" "700000 500000" "1000000 400000"; do
set -- $nm
python bench.py $B --n-indv $1 --n-snp $2 2>/dev/null | tail -1 | python -c "$fmt" "N=$1 M=$2"
done
