#!/bin/bash
# A/B of one engine option on one box: tools/r2_opt_ab.sh NAME V1 V2 -> BASELINE configurations C2..C5 with --opt NAME=V
cd "${GRAFT_REPO_ROOT:-/root/repo}"
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"])'
name=$1; shift
for rep in 1 2; do for v in "$@"; do
  echo "rep $rep $name=$v"
  timeout -k 10 200 python bench.py --n-indv 50000 --n-snp 500000 --steps 3 --no-cpu-baseline --no-alt --no-cross-check --opt $name=$v 2>/dev/null | python -c "$fmt" C2 || exit 1
  timeout -k 10 200 python bench.py --steps 3 --n-bin 8 --n-cov 10 --no-cpu-baseline --no-alt --no-cross-check --opt $name=$v 2>/dev/null | python -c "$fmt" C4 || exit 1
  timeout -k 10 200 python bench.py --steps 3 --genie G+GxE --no-cpu-baseline --no-alt --no-cross-check --opt $name=$v 2>/dev/null | python -c "$fmt" C5 || exit 1
done; done
