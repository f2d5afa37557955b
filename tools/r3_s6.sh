#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_headline_gpu.py -x -q --durations=30 > gpurun_out/r3/t6.log 2>&1; echo "headline rc=$?"; tail -40 gpurun_out/r3/t6.log
bash tools/r3_c2_knobs.sh > gpurun_out/r3/c2_knobs.txt 2>&1; cat gpurun_out/r3/c2_knobs.txt
