#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "full_line" 2>&1 | tail -3 || exit 1
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), "M SNPs/s", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], r.get("kernels"))'
B="--steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check"
for o in "" "--opt g1_line=0" "--opt g1_workgroups=1024" "--opt g1_workgroups=2304" "--opt g1_split_major=2"; do
python bench.py $B --genie G+GxE $o 2>/dev/null | tail -1 | python -c "$fmt" "C5 [$o]"
done
for o in "" "--opt g1_line=0"; do
python bench.py $B --genie G+GxE --n-cov 10 $o 2>/dev/null | tail -1 | python -c "$fmt" "C5cov [$o]"
python bench.py $B --n-bin 8 --n-cov 10 --materialize-xxuz $o 2>/dev/null | tail -1 | python -c "$fmt" "C4-UZ [$o]"
done
