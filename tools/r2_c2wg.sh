#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), round(d["ms_per_step"],3), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"])'
for w in 400 512 768 1024 1536 2048 3072; do
timeout -k 10 200 python bench.py --n-indv 50000 --n-snp 500000 --steps 5 --no-cpu-baseline --no-alt --no-cross-check --opt g1_workgroups=$w 2>/dev/null | python -c "$fmt" "C2 wgs=$w" || exit 1
done
