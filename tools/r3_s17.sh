#!/bin/bash
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), "M SNPs/s", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"])'
for o in "" "--opt g1_workgroups=1024" "--opt g1_workgroups=1280" "--opt g1_workgroups=2048" "--opt g1_fill=0"; do
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check $o 2>/dev/null | tail -1 | python -c "$fmt" "C3 [$o]"
done
for c in 65536 262144; do
PYRHE_CAPACITY=$c python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | tail -1 | python -c "$fmt" "C3 capacity $c"
done
