#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), round(d["ms_per_step"],3), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"])'
for rep in 1 2; do for p in "" "--pipeline"; do
timeout -k 10 200 python bench.py --steps 3 --no-cpu-baseline --no-alt --no-cross-check $p 2>/dev/null | python -c "$fmt" "C3 $p" || exit 1
done; done
for p in "" "--pipeline"; do
timeout -k 10 200 python bench.py --n-indv 50000 --n-snp 500000 --steps 5 --no-cpu-baseline --no-alt --no-cross-check $p 2>/dev/null | python -c "$fmt" "C2 $p" || exit 1
timeout -k 10 200 python bench.py --steps 3 --n-bin 8 --n-cov 10 --no-cpu-baseline --no-alt --no-cross-check $p 2>/dev/null | python -c "$fmt" "C4 $p" || exit 1
done
