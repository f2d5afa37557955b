#!/bin/bash
# headline GEMM 1: the last tile's two live rows from a compact copy (g1_compact)
set -o pipefail
mkdir -p gpurun_out/r3
python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "launch_shape" 2>&1 | tail -3 || exit 1
python -m pytest tests/test_headline_gpu.py -x -q -m gpu -k "headline or n500k" 2>&1 | tail -3 || exit 1
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), "M SNPs/s", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "frac", r["frac"], r.get("kernels"))'
B="--steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check"
for o in "" "--opt g1_compact=0" "" "--opt g1_compact=0"; do
python bench.py $B $o 2>/dev/null | tail -1 | python -c "$fmt" "C3 [$o]"
done
for o in "" "--opt g1_compact=0"; do
python bench.py $B --n-indv 50000 --n-snp 500000 $o 2>/dev/null | tail -1 | python -c "$fmt" "C2 [$o]"
done
