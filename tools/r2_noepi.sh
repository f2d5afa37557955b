#!/bin/bash
# GEMM 2 with and without the digit-combining epilogue matrix instructions (diagnostic library, wrong results).
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"])'
for lib in "" pyrhe_amd/csrc/libpyrhe_sketch_noepi.so; do
  export PYRHE_LIB=$lib; echo "lib=$lib"
  timeout -k 10 200 python bench.py --n-indv 50000 --n-snp 500000 --steps 3 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | python -c "$fmt" C2 || exit 1
  timeout -k 10 200 python bench.py --steps 3 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | python -c "$fmt" C3 || exit 1
  timeout -k 10 200 python bench.py --steps 3 --n-bin 8 --n-cov 10 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | python -c "$fmt" C4 || exit 1
  timeout -k 10 200 python bench.py --steps 3 --genie G+GxE --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | python -c "$fmt" C5 || exit 1
done
