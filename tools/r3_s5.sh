#!/bin/bash
mkdir -p gpurun_out/r3
bash tools/r3_rank_proxy.sh > gpurun_out/r3/rank_proxy.txt 2>&1; cat gpurun_out/r3/rank_proxy.txt
python -m pytest tests/test_headline_gpu.py -x -q --durations=30 > gpurun_out/r3/t5.log 2>&1; echo "headline rc=$?"; tail -45 gpurun_out/r3/t5.log
