#!/bin/bash
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1].ljust(50), round(d["value"]/1e6,2), "M |", round(d["ms_per_step"],1), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"])'
B="--steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --jn 10"
for o in "" "--opt g1_fill=0" "--opt g1_workgroups=1024" "--opt g1_workgroups=2048" "--opt g1_workgroups=1300 --opt g1_fill=0" "--opt g1_workgroups=1800 --opt g1_fill=0" "--opt g1_workgroups=2500 --opt g1_fill=0" "--opt g1_workgroups=800 --opt g1_fill=0"; do
python bench.py $B $o 2>/dev/null | tail -1 | python -c "$fmt" "[jn 10 $o]"
done
