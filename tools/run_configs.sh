#!/bin/bash
# The BASELINE.json configurations on one GPU (sketch pass + end to end); one JSON line each under gpurun_out/configs/.
mkdir -p gpurun_out/configs
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), "M SNPs/s", round(d["ms_per_step"],1), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "| e2e", d["config"]["end_to_end"])'
python bench.py --n-indv 50000 --n-snp 500000 --steps 3 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | tee gpurun_out/configs/c2.json | python -c "$fmt" C2
python bench.py --steps 5 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | tee gpurun_out/configs/c3.json | python -c "$fmt" C3
python bench.py --steps 3 --n-bin 8 --n-cov 10 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | tee gpurun_out/configs/c4.json | python -c "$fmt" C4
python bench.py --steps 3 --genie G+GxE --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | tee gpurun_out/configs/c5.json | python -c "$fmt" C5
python bench.py --steps 3 --precision f16x2 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | tee gpurun_out/configs/c3_f16x2.json | python -c "$fmt" C3-f16x2
python bench.py --steps 3 --precision f16 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | tee gpurun_out/configs/c3_f16.json | python -c "$fmt" C3-f16
