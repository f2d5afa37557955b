#!/bin/bash
set -o pipefail
python -m pytest tests/test_headline_gpu.py -x -q -m gpu -k "long_blocks or split_into_row" 2>&1 | tail -3 || exit 1
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1].ljust(36), round(d["value"]/1e6,2), "M |", round(d["ms_per_step"],1), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "| e2e", round(c["end_to_end"]["snps_per_s"]/1e6,2), c["sigma_total"][:2], c["sigma_se"][:2])'
B="--steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --verify --dom"
for cfg in "--jn 40" "--jn 100"; do
python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[dom $cfg]"
PYRHE_BATCH_PIECES=0 python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "   engine chunking"
done
