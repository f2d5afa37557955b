#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q > gpurun_out/r3/full3.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/r3/full3.log
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), "M SNPs/s", round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], r["kernels"], "| e2e", d["config"]["end_to_end"], d["config"].get("sigma_total"), d["config"].get("sigma_se"))'
python bench.py --genie G+GxE --n-cov 10 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --verify 2>/dev/null | tail -1 | python -c "$fmt" "G+GxE + 10 cov, one wide pass"
python bench.py --genie G+GxE --n-cov 10 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --verify --materialize-xxuz 2>/dev/null | tail -1 | python -c "$fmt" "G+GxE + 10 cov, U Z columns  "
python bench.py --genie G+GxE --n-cov 5 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | tail -1 | python -c "$fmt" "G+GxE + 5 cov"
python bench.py --n-cov 25 --n-bin 4 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --verify 2>/dev/null | tail -1 | python -c "$fmt" "4 bins + 25 cov (36 columns)"
python bench.py --n-cov 25 --n-bin 4 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --verify --materialize-xxuz 2>/dev/null | tail -1 | python -c "$fmt" "4 bins + 25 cov, U Z columns "
