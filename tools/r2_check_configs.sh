#!/bin/bash
# The whole GPU suite, then the BASELINE configurations (tools/run_configs.sh) at this tree.
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=25 > $O/tests_check.log 2>&1
rc=$?; echo "pytest rc $rc"; grep -E "passed|failed|^FAILED|^E  " $O/tests_check.log | cut -c1-300 | tail -n 30
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 bash tools/run_configs.sh
