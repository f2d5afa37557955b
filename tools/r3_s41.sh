#!/bin/bash
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1].ljust(46), round(d["ms_per_step"],1), "ms | g1", r["gemm1_ms_per_step"], "| rows per launch", r["bytes_per_launch"]//125184, c["sigma_total"][:2], c["sigma_se"][:2])'
B="--steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --verify"
for cfg in "--jn 28 --missing-rate 0.01" "--jn 100 --missing-rate 0.01" "--jn 28 --missing-rate 0.01 --drop-indv 50 --n-bin 2"; do
python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[$cfg]"
PYRHE_BATCH_PIECES=0 python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "$fmt" "   engine chunking:"
done
