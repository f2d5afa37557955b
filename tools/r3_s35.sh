#!/bin/bash
# jackknife block counts beside jn = 100: long blocks (row pieces) and many short ones
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1].ljust(40), round(d["value"]/1e6,2), "M |", round(d["ms_per_step"],1), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "| pass2", c["end_to_end"]["pass2_s"], "|", r["kernels"]["gemm1"], r["kernels"]["gemm2"])'
B="--steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check"
for cfg in "--jn 10" "--jn 20" "--jn 400" "--jn 1000" "--jn 1000 --n-bin 8 --n-cov 10" "--jn 10 --n-bin 8 --n-cov 10" "--n-indv 50000 --n-snp 500000 --jn 1000"; do
python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[$cfg]"
done
