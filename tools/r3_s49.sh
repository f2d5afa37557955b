#!/bin/bash
# N = 500k with bins: whole segment tables per call with a larger call capacity (16 / 24 blocks per call at 8 bins)
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1].ljust(64), round(d["value"]/1e6,2), "M |", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "| rows/launch", r["bytes_per_launch"]//125184)'
B="--steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check"
for cfg in "--n-bin 8 --n-cov 10" "--n-bin 32 --n-cov 5" "--n-bin 4 --n-cov 25"; do
python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[$cfg]"
PYRHE_WHOLE_TABLES=1 PYRHE_CAPACITY=196608 python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "   whole tables, capacity 196608:"
PYRHE_WHOLE_TABLES=1 PYRHE_CAPACITY=262144 python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "   whole tables, capacity 262144:"
done
