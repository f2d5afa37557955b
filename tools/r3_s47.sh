#!/bin/bash
# calls of several segment tables through the runner, again (the first measurement had a CPU-bound runner)
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1].ljust(58), round(d["value"]/1e6,2), "M |", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], r["kernels"]["gemm1"])'
B="--steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check"
for cfg in "--n-indv 50000 --n-snp 500000 --n-bin 8 --n-cov 10" "--n-indv 50000 --n-snp 500000 --n-bin 4" "--n-indv 100000 --n-snp 400000 --n-bin 8 --n-cov 10" "--n-bin 8 --n-cov 10" "--n-bin 32 --n-cov 5" "--jn 1000"; do
python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[$cfg]"
PYRHE_LIFT_LISTS=1 python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "   several tables per call:"
done
