#!/bin/bash
# RHE_DOM: one fused second contraction (dom_fuse) against three launches
set -o pipefail
mkdir -p gpurun_out/r3
python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "large_cohort or dominance or random_shapes" 2>&1 | tail -5 || exit 1
python -m pytest tests/test_headline_gpu.py -x -q -m gpu -k "dom" 2>&1 | tail -5 || exit 1
python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "dom" 2>&1 | tail -5 || exit 1
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), "M SNPs/s", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], r.get("kernels"), "| e2e", d["config"].get("end_to_end"))'
B="--steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check"
for o in "" "--opt dom_fuse=0"; do
python bench.py $B --dom $o 2>/dev/null | tail -1 | python -c "$fmt" "DOM [$o]"
python bench.py $B --dom --n-bin 8 --n-cov 10 $o 2>/dev/null | tail -1 | python -c "$fmt" "DOM-C4 [$o]"
done
