#!/bin/bash
# round 3, session 3: the Infinity-Cache gate (leader-progress pacing), RHE_DOM through the segmented flow, tests touched since
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 tools/bin/ic_pipeline > gpurun_out/r3/ic_pipeline.txt 2>&1; echo "ic rc=$?"; tail -34 gpurun_out/r3/ic_pipeline.txt
python -m pytest tests/test_headline_gpu.py tests/test_families_gpu.py tests/test_midsize_gpu.py tests/test_models_gpu.py -x -q -k "dom or DOM" > gpurun_out/r3/t3.log 2>&1; echo "dom tests rc=$?"; tail -5 gpurun_out/r3/t3.log
python -m pytest tests/test_distributed_gpu.py -x -q > gpurun_out/r3/t3b.log 2>&1; echo "dist tests rc=$?"; tail -3 gpurun_out/r3/t3b.log
python bench.py --dom --no-cpu-baseline --no-alt --no-cross-check --steps 2 --warmup 1 > gpurun_out/r3/dom.json 2> gpurun_out/r3/dom.err; echo "dom bench rc=$?"; tail -c 1500 gpurun_out/r3/dom.json
python bench.py --dom --no-cpu-baseline --no-alt --no-cross-check --steps 2 --warmup 1 --n-bin 8 --n-cov 10 > gpurun_out/r3/dom_c4.json 2> gpurun_out/r3/dom_c4.err; echo "dom c4 bench rc=$?"; tail -c 1500 gpurun_out/r3/dom_c4.json
