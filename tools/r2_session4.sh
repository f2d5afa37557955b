#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=25 > $O/tests4.log 2>&1
rc=$?; echo "pytest rc $rc"; grep -E "passed|failed|^FAILED|^E  " $O/tests4.log | cut -c1-300 | tail -n 30
if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/run_configs.sh 2>&1 | tail -n 12
