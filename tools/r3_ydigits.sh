#!/bin/bash
# round 3: what a 16-bit phenotype column (two int8 digits) would do to sigma^2 and the jackknife SE -- the price of a two-tile GEMM 1
# at the headline (10 probes x 3 digits + 2 = 32 digit rows, with the genotype sums cached at load instead of the ones row)
for N in 200000 500000; do for H in 0.5 0.1; do for D in 3 2; do
python bench.py --steps 1 --warmup 0 --n-indv $N --n-snp 200000 --jn 100 --no-cpu-baseline --no-alt --no-cross-check --verify --h2 $H --norm-digits $D 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$N h2=$H digits=$D', repr(c['sigma_total']), repr(c['sigma_se']))"
done; done; done
