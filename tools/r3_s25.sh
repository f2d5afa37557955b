#!/bin/bash
# whole GPU suite at HEAD, then the N = 50k / 8 bins + covariates pass twice (a 24 ms step was seen once)
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), "M SNPs/s", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], r.get("kernels"))'
B="--steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check"
for i in 1 2 3; do
python bench.py $B --n-bin 8 --n-cov 10 --n-indv 50000 --n-snp 500000 2>/dev/null | tail -1 | python -c "$fmt" "C2-8bin run $i"
done
