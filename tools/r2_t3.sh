#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_headline_gpu.py tests/test_engine_gpu.py tests/test_midsize_gpu.py -m gpu -q -x > $O/t3.log 2>&1
rc=$?; echo "pytest rc $rc"; grep -E "passed|failed|^FAILED|^E  " $O/t3.log | cut -c1-300 | tail -n 12
if [ $rc -ne 0 ]; then exit $rc; fi
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"], "frac", r["frac"], "launch ms", r["avg_launch_ms"], "bytes", r["bytes_per_launch"])'
for rep in 1 2; do
timeout -k 10 200 python bench.py --steps 3 --no-cpu-baseline --no-alt 2>$O/b.err | python -c "$fmt" "C3" || { tail -5 $O/b.err; exit 1; }
done
timeout -k 10 200 python bench.py --steps 3 --no-cpu-baseline --no-alt --no-cross-check --sub-snps 131072 2>$O/b.err | python -c "$fmt" "C3 cap131072" || { tail -5 $O/b.err; exit 1; }
