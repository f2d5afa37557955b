#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_midsize_gpu.py -x -q -k "pass2_device_assembly" > gpurun_out/r3/t10.log 2>&1; echo "pass2 tests rc=$?"; tail -3 gpurun_out/r3/t10.log
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), "M SNPs/s", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], r["kernels"]["gemm1"])'
for o in "" "--opt g1_workgroups=768" "--opt g1_workgroups=3072" "--opt g1_workgroups=6144" "--opt g1_split_major=1" "--opt g1_zero_rows=0"; do
python bench.py --genie G+GxE --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check $o 2>/dev/null | tail -1 | python -c "$fmt" "C5 [$o]"
done
