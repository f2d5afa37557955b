#!/bin/bash
# A/B on one box: libraries given as arguments (PYRHE_LIB; "-" = the product library), BASELINE configurations C2..C5.
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"])'
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "-" ]; then export PYRHE_LIB=; else export PYRHE_LIB=$lib; fi; echo "rep $rep lib=$lib opts=$OPTS"
  timeout -k 10 200 python bench.py --n-indv 50000 --n-snp 500000 --steps 3 --no-cpu-baseline --no-alt --no-cross-check $OPTS 2>/dev/null | python -c "$fmt" C2 || exit 1
  timeout -k 10 200 python bench.py --steps 3 --no-cpu-baseline --no-alt --no-cross-check $OPTS 2>/dev/null | python -c "$fmt" C3 || exit 1
  timeout -k 10 200 python bench.py --steps 3 --n-bin 8 --n-cov 10 --no-cpu-baseline --no-alt --no-cross-check $OPTS 2>/dev/null | python -c "$fmt" C4 || exit 1
  timeout -k 10 200 python bench.py --steps 3 --genie G+GxE --no-cpu-baseline --no-alt --no-cross-check $OPTS 2>/dev/null | python -c "$fmt" C5 || exit 1
done; done
