#!/bin/bash
# GEMM 1 per pass against the rows one call carries (block size x blocks per call), N = 500k, M = 1M
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1].ljust(16), round(d["value"]/1e6,2), "M |", round(d["ms_per_step"],1), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "| launches", r["avg_launch_ms"], r["bytes_per_launch"]//125184)'
B="--steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check"
for jn in 30 40 50 60 70 80 90 100 125 150 200; do
python bench.py $B --jn $jn 2>/dev/null | tail -1 | python -c "$fmt" "[jn $jn]"
done
