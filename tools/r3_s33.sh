#!/bin/bash
# shapes beside the BASELINE configurations: which kernels, pass 1 / pass 2 times (anything falling back to a slow path?)
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1].ljust(46), round(d["value"]/1e6,2), "M |", round(d["ms_per_step"],1), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "| pass2", c["end_to_end"]["pass2_s"], "|", r["kernels"]["gemm1"], r["kernels"]["gemm2"])'
B="--steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check"
for cfg in "--k 20" "--k 5" "--n-bin 16" "--n-bin 32 --n-cov 5" "--n-bin 4 --n-cov 25" "--n-bin 2 --n-cov 40" "--missing-rate 0.01" "--n-bin 8 --n-cov 10 --missing-rate 0.01" "--genie G+GxE+NxE --n-cov 5" "--genie G+GxE --n-bin 4" "--dom --n-bin 4" "--k 30" "--n-bin 8 --n-cov 10 --compat reference"; do
python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[$cfg]"
done
