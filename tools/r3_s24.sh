#!/bin/bash
# merged GEMM 2: rotated segment order per workgroup (g2_rotate)
set -o pipefail
mkdir -p gpurun_out/r3
python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "merged or multibin" 2>&1 | tail -3 || exit 1
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), "M SNPs/s", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], r.get("kernels"))'
B="--steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check"
for o in "" "--opt g2_rotate=0" "" "--opt g2_rotate=0"; do
python bench.py $B --n-bin 8 --n-cov 10 $o 2>/dev/null | tail -1 | python -c "$fmt" "C4 [$o]"
done
for o in "" "--opt g2_rotate=0"; do
python bench.py $B --n-bin 8 --n-cov 10 --materialize-xxuz $o 2>/dev/null | tail -1 | python -c "$fmt" "C4-UZ [$o]"
python bench.py $B --n-bin 8 --n-cov 10 --n-indv 50000 --n-snp 500000 $o 2>/dev/null | tail -1 | python -c "$fmt" "C2-8bin [$o]"
done
