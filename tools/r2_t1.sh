#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=25 > $O/t1.log 2>&1
rc=$?; echo "pytest rc $rc"; grep -E "passed|failed|^FAILED|^E  " $O/t1.log | cut -c1-300 | tail -n 30
