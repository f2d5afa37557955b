#!/bin/bash
# Final state check on the GPU box: build entry, smoke, the whole GPU suite, the default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3
timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=25 > $O/tests_final.log 2>&1
rc=$?; echo "pytest rc $rc"; grep -E "passed|failed|^FAILED|^E  " $O/tests_final.log | cut -c1-300 | tail -n 30
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc $?"; tail -c 2600 $O/bench_final.json
