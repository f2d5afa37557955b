#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "frac", r["frac"])'
for w in 1024 1536 2048 2560 3072 4096; do
timeout -k 10 200 python bench.py --steps 3 --no-cpu-baseline --no-alt --no-cross-check --opt g1_workgroups=$w 2>/dev/null | python -c "$fmt" "wgs=$w" || exit 1
done
