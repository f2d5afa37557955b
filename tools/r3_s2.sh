#!/bin/bash
# round 3, session 2: the Infinity-Cache gate (flag version), the tests changed since session 1, RCCL on one rank
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 400 tools/bin/ic_pipeline > gpurun_out/r3/ic_pipeline.txt 2>&1; echo "ic rc=$?"; tail -40 gpurun_out/r3/ic_pipeline.txt
python -m pytest tests/test_families_gpu.py tests/test_distributed_gpu.py tests/test_headline_gpu.py tests/test_models_gpu.py -x -q -k "not beyond_4gb" > gpurun_out/r3/t2.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3/t2.log
python -m pytest tests/test_engine_gpu.py -x -q -k "cross or upload" > gpurun_out/r3/t2b.log 2>&1; echo "engine tests rc=$?"; tail -3 gpurun_out/r3/t2b.log
python tools/upload_rate.py > gpurun_out/r3/upload_rate.txt 2>&1; cat gpurun_out/r3/upload_rate.txt
