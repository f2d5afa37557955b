#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_engine_gpu.py tests/test_midsize_gpu.py -x -q -k "dots or pass2 or segments_and_dots" > gpurun_out/r3/t7.log 2>&1; echo "dots tests rc=$?"; tail -4 gpurun_out/r3/t7.log
bash tools/r3_dots_f32.sh > gpurun_out/r3/dots_f32_xcd.txt 2>&1; cut -c1-120 gpurun_out/r3/dots_f32_xcd.txt
python bench.py --n-indv 50000 --n-snp 500000 --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-cross-check > gpurun_out/r3/c2_rule.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3/c2_rule.json')); r=d['roofline']; print('C2 with the 32-step rule', round(d['value']/1e6,1), 'M SNPs/s', round(d['ms_per_step'],3), 'ms g1', r['gemm1_ms_per_step'], 'g2', r['gemm2_ms_per_step'])"
