#!/bin/bash
set -o pipefail
python -m pytest tests/test_midsize_gpu.py -x -q -m gpu -k "pass2_device" 2>&1 | tail -5 || exit 1
for cfg in "--n-bin 32 --n-cov 5" "--n-bin 24" "--dom --n-bin 12 --n-cov 5"; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --verify --h2 0.3 $cfg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('[$cfg] pass1', round(d['ms_per_step'],1), 'pass2_s', c['end_to_end']['pass2_s'], 'e2e', round(c['end_to_end']['snps_per_s']/1e6,2), repr(c['sigma_total'][:2]), repr(c['sigma_se'][:2]))"
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --verify --h2 0.3 $cfg --opt dots_generic=1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('   generic:', 'pass2_s', c['end_to_end']['pass2_s'], repr(c['sigma_total'][:2]), repr(c['sigma_se'][:2]))"
done
