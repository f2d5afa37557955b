#!/bin/bash
set -o pipefail
python -m pytest tests/test_models_gpu.py tests/test_families_gpu.py tests/test_midsize_gpu.py tests/test_distributed_gpu.py tests/test_example_config.py -x -q -m gpu 2>&1 | tail -3 || exit 1
python -m pytest tests/test_headline_gpu.py -x -q -m gpu -k "n50k or c4_shape or one_wide" 2>&1 | tail -3 || exit 1
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1].ljust(58), round(d["value"]/1e6,2), "M |", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], r["kernels"]["gemm1"], c["sigma_total"][:2])'
B="--steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check --verify"
for cfg in "--n-indv 50000 --n-snp 500000 --n-bin 8 --n-cov 10" "--n-indv 50000 --n-snp 500000 --n-bin 4" "--n-indv 100000 --n-snp 400000 --n-bin 8 --n-cov 10" "--n-indv 50000 --n-snp 500000 --genie G+GxE --n-bin 4" "--n-indv 50000 --n-snp 500000"; do
python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[$cfg]"
PYRHE_ONE_TABLE=1 python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "   one table per call:"
done
