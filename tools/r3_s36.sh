#!/bin/bash
set -o pipefail
python -m pytest tests/test_headline_gpu.py -x -q -m gpu -k "long_blocks or split_into_row" 2>&1 | tail -5 || exit 1
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1].ljust(40), round(d["value"]/1e6,2), "M |", round(d["ms_per_step"],1), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "| pass2", c["end_to_end"]["pass2_s"])'
B="--steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check"
for cfg in "--jn 10" "--jn 20" "--jn 10 --n-bin 8 --n-cov 10" "--jn 25 --genie G+GxE"; do
python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[$cfg]"
done
