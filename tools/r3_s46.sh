#!/bin/bash
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1].ljust(52), round(d["value"]/1e6,2), "M |", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"])'
B="--steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check"
for cfg in "--n-indv 50000 --n-snp 500000 --n-bin 8 --n-cov 10" "--n-indv 50000 --n-snp 500000 --n-bin 8 --n-cov 10" "--n-indv 50000 --n-snp 500000" "" "--n-bin 8 --n-cov 10" "--genie G+GxE" "--jn 30"; do
python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[$cfg]"
done
