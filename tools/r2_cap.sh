#!/bin/bash
# Blocks per engine call (PYRHE_CAPACITY) at the headline shape: 1, 2, 3 blocks per launch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["ms_per_step"],2), "ms | g1", r["gemm1_ms_per_step"], "g2", r["gemm2_ms_per_step"], "mid", r["mid_ms_per_step"])'
for rep in 1 2; do for cap in 10000 20000 65536; do
  PYRHE_CAPACITY=$cap PYRHE_SUB_SNPS=$cap timeout -k 10 200 python bench.py --steps 3 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | python -c "$fmt" "cap=$cap" || exit 1
done; done
