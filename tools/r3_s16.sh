#!/bin/bash
for cfg in "--n-bin 8 --n-cov 10" "--genie G+GxE" ""; do for o in "--opt dots_tiles=2" "--opt dots_tiles=1"; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --verify --h2 0.3 $o $cfg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('[$cfg] $o pass2_s', c['end_to_end']['pass2_s'], 'e2e', round(c['end_to_end']['snps_per_s']/1e6,2), repr(c['sigma_total'][:2]), repr(c['sigma_se'][:2]))"
done; done
