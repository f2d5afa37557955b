#!/bin/bash
# config 2 (N = 50k): GEMM 1's sample-split target against the per-workgroup cost of its integer atomics (short loops at small N)
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,1), "M SNPs/s", round(d["ms_per_step"],3), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], r["kernels"]["gemm1"])'
for o in "" "--opt g1_workgroups=512" "--opt g1_workgroups=768" "--opt g1_workgroups=1024" "--opt g1_workgroups=2304" "--opt g1_wide=0" "--opt g1_wide=0 --opt g1_workgroups=768"; do
python bench.py --n-indv 50000 --n-snp 500000 --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-cross-check $o 2>/dev/null | tail -1 | python -c "$fmt" "C2 [$o]"
done
