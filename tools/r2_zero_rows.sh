#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_headline_gpu.py tests/test_engine_gpu.py -m gpu -q -x > $O/t2.log 2>&1
rc=$?; echo "pytest rc $rc"; grep -E "passed|failed|^FAILED|^E  " $O/t2.log | cut -c1-300 | tail -n 10
if [ $rc -ne 0 ]; then exit $rc; fi
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "frac", r["frac"])'
for rep in 1 2 3; do for z in 1 0; do
  timeout -k 10 200 python bench.py --steps 3 --no-cpu-baseline --no-alt --no-cross-check --opt g1_zero_rows=$z 2>/dev/null | python -c "$fmt" "C3 zero_rows=$z" || exit 1
done; done
