#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]/1e6,2), round(d["ms_per_step"],3), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], r["kernels"])'
timeout -k 10 200 python bench.py --steps 2 --no-cpu-baseline --no-alt --no-cross-check --opt g2_packed=0 2>$O/b5.err | python -c "$fmt" "C3 g2_packed=0" || { tail -5 $O/b5.err; exit 1; }
timeout -k 10 600 python -m pytest tests/test_headline_gpu.py tests/test_midsize_gpu.py -m gpu -q -x 2>&1 | tail -2
