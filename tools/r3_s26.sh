#!/bin/bash
# three ring stages for the unmerged four-tile GEMM 2 (G + GxE)
set -o pipefail
mkdir -p gpurun_out/r3
python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "multibin or merged or random_shapes" 2>&1 | tail -3 || exit 1
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), "M SNPs/s", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], r.get("kernels"))'
B="--steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check"
for i in 1 2; do
python bench.py $B --genie G+GxE 2>/dev/null | tail -1 | python -c "$fmt" "C5"
done
python bench.py $B --genie G+GxE --n-cov 10 2>/dev/null | tail -1 | python -c "$fmt" "C5cov"
python bench.py $B --genie G+GxE --n-indv 50000 --n-snp 500000 2>/dev/null | tail -1 | python -c "$fmt" "C5 N=50k"
