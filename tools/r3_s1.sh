#!/bin/bash
# round 3, session 1: new engine tests + the rest of the suite from where the first run stopped, the Infinity-Cache gate,
# C4 with covariates as cross columns (1 digit) against the materialised U Z columns, the two-tile GEMM 1 probe (k = 9)
set -o pipefail
mkdir -p gpurun_out/r3
python -m pytest tests/test_engine_gpu.py -x -q -k "cross" > gpurun_out/r3/t_cross.log 2>&1; echo "cross tests rc=$?"; tail -3 gpurun_out/r3/t_cross.log
python -m pytest tests/test_midsize_gpu.py tests/test_models_gpu.py tests/test_families_gpu.py -x -q > gpurun_out/r3/t_rest.log 2>&1; echo "rest rc=$?"; tail -3 gpurun_out/r3/t_rest.log
timeout -k 10 300 tools/bin/ic_pipeline > gpurun_out/r3/ic_pipeline.txt 2>&1; echo "ic rc=$?"
python bench.py --n-bin 8 --n-cov 10 --no-cpu-baseline --no-alt --steps 3 --warmup 2 > gpurun_out/r3/c4_cross.json 2> gpurun_out/r3/c4_cross.err; echo "c4 cross rc=$?"
python bench.py --n-bin 8 --n-cov 10 --no-cpu-baseline --no-alt --steps 3 --warmup 2 --materialize-xxuz > gpurun_out/r3/c4_mat.json 2> gpurun_out/r3/c4_mat.err; echo "c4 mat rc=$?"
python bench.py --k 9 --no-cpu-baseline --no-alt --no-cross-check --steps 3 --warmup 2 > gpurun_out/r3/c3_k9.json 2> gpurun_out/r3/c3_k9.err; echo "k9 rc=$?"
python bench.py --no-cpu-baseline --no-alt --no-cross-check --steps 3 --warmup 2 > gpurun_out/r3/c3.json 2> gpurun_out/r3/c3.err; echo "c3 rc=$?"
