#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_headline_gpu.py tests/test_models_gpu.py -m gpu -q -s -k "headline or c4_shape or genie" > $O/tests3.log 2>&1
echo "pytest rc $?"; grep -E "^\[N=|passed|failed|^FAILED|^E  " $O/tests3.log | cut -c1-400 | tail -n 30
timeout -k 10 200 tools/bin/pattern2 > $O/pattern2.txt 2>&1; echo "pattern2 rc $?"; cat $O/pattern2.txt
timeout -k 10 900 bash tools/pmc_config.sh c5 --genie G+GxE > $O/pmc_c5.txt 2>&1; echo "pmc rc $?"; tail -n 90 $O/pmc_c5.txt
