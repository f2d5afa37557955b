#!/bin/bash
# pass 2 with 16 estimates: <XXz, U Z> through the projections / a second pass for fixed rows beyond 16
set -o pipefail
python -m pytest tests/test_midsize_gpu.py -x -q -m gpu -k "pass2_device" 2>&1 | tail -3 || exit 1
python tools/dbg_multi.py 2>&1 | grep -E "multi_ok" | cut -c1-90 | sort | uniq -c | head -5
for cfg in "--dom --n-bin 8 --n-cov 10" "--genie G+GxE --n-bin 8 --n-cov 10" "--n-bin 8 --n-cov 10"; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --verify --h2 0.3 $cfg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('[$cfg] pass1', round(d['ms_per_step'],1), 'pass2_s', c['end_to_end']['pass2_s'], 'e2e', round(c['end_to_end']['snps_per_s']/1e6,2), repr(c['sigma_total'][:3]), repr(c['sigma_se'][:3]))"
done
