#!/bin/bash
# pass 2's one read of the block store on the float32 matrix instruction (PYRHE_OPT_DOTS_F32) against the float64 one: time and results
mkdir -p gpurun_out/r3
for cfg in "" "--n-bin 8 --n-cov 10" "--genie G+GxE" "--n-bin 8 --n-cov 10 --compat reference"; do for o in 0 1; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --verify --h2 0.3 --opt dots_f32=$o $cfg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('[$cfg] dots_f32=$o pass2_s', c['end_to_end']['pass2_s'], 'e2e', round(c['end_to_end']['snps_per_s']/1e6,2), repr(c['sigma_total']), repr(c['sigma_se']))"
done; done
