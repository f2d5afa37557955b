#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_headline_gpu.py -m gpu -q -x -k "n50k" > $O/t4.log 2>&1
rc=$?; echo "pytest rc $rc"; grep -E "passed|failed|^FAILED|^E  " $O/t4.log | cut -c1-300 | tail -n 12
