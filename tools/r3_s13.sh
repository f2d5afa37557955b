#!/bin/bash
mkdir -p gpurun_out/r3
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), "M SNPs/s", round(d["ms_per_step"],3), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "| e2e", round(d["config"]["end_to_end"]["snps_per_s"]/1e6,2))'
python bench.py --n-indv 50000 --n-snp 500000 --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | tee gpurun_out/r3/c2_final.json | tail -1 | python -c "$fmt" "C2 default"
for o in "" "--opt g2_merge=0" "--opt g2_merge=768" "--opt g2_merge=3072" "--opt g2_merge=6144"; do
python bench.py --n-bin 8 --n-cov 10 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check $o 2>/dev/null | tail -1 | python -c "$fmt" "C4 [$o]"
done
python -m pytest tests/test_headline_gpu.py -x -q -k "n50k" 2>&1 | tail -2
