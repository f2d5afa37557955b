#!/bin/bash
# pass 2 with float32 copies of the fixed rows: tests, then pass-2 time per configuration and tile count
set -o pipefail
mkdir -p gpurun_out/r3
python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "loo_dots" 2>&1 | tail -3 || exit 1
python -m pytest tests/test_midsize_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for cfg in "--n-bin 8 --n-cov 10" "--dom --n-bin 8 --n-cov 10" "--genie G+GxE" "--n-bin 8 --n-cov 10 --compat reference"; do for o in "dots_tiles=2" "dots_tiles=1" "dots_tiles=4"; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --verify --h2 0.3 --opt $o $cfg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('[$cfg] $o pass2_s', c['end_to_end']['pass2_s'], 'e2e', round(c['end_to_end']['snps_per_s']/1e6,2), repr(c['sigma_total'][:3]), repr(c['sigma_se'][:3]))"
done; done
