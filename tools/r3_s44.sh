#!/bin/bash
set -o pipefail
python -m pytest tests/test_headline_gpu.py -x -q -m gpu -k "beyond_4gb or c4_shape or straddle" 2>&1 | tail -3 || exit 1
python -m pytest tests/test_distributed_gpu.py tests/test_midsize_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1].ljust(46), round(d["value"]/1e6,2), "M |", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "|", r["kernels"]["gemm1"], c["sigma_total"][:2])'
B="--steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --verify"
for cfg in "--n-bin 8 --n-cov 10" "--n-bin 32 --n-cov 5" "--n-indv 50000 --n-snp 500000 --n-bin 8 --n-cov 10" "--jn 1000" "--dom --n-bin 8 --n-cov 10" "--n-indv 50000 --n-snp 500000 --n-bin 8 --n-cov 10"; do
python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[$cfg]"
done
