#!/bin/bash
# Cross-check sweep on the GPU box: unusual shapes through bench.py --verify, the default i8x3 (packed int8 kernels,
# segmented flow) against f16x2 (fp16 kernels) -- two independent kernel families must agree to ~1e-6 on sigma^2 and SEs.
fmt='import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; print(sys.argv[1], [round(x,9) for x in c["sigma_total"][:4]], [round(x,9) for x in c["sigma_se"][:4]])'
run() { tag="$1"; shift; for p in i8x3 f16x2; do timeout -k 10 300 python bench.py --precision $p "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-alt --no-cross-check --verify 2>/dev/null | tail -1 | python -c "$fmt" "$tag/$p"; done; }
run "blocks>sub_snps" --n-indv 30000 --n-snp 200000 --jn 3
run "tiny-blocks" --n-indv 100000 --n-snp 100000 --jn 100
run "24bins" --n-indv 50000 --n-snp 120000 --jn 20 --n-bin 24
run "3bins+15cov" --n-indv 20000 --n-snp 50000 --jn 5 --n-bin 3 --n-cov 15
run "genie-nxe+cov" --n-indv 40000 --n-snp 80000 --jn 10 --genie G+GxE+NxE --n-cov 4
run "N500k-seg-cut" --n-snp 60000 --jn 2
run "k4" --n-indv 60000 --n-snp 60000 --jn 10 --k 4 --n-bin 2
run "k25" --n-indv 60000 --n-snp 60000 --jn 10 --k 25 --n-bin 2
run "k25+cov" --n-indv 60000 --n-snp 40000 --jn 10 --k 25 --n-cov 3
run "odd-N" --n-indv 33331 --n-snp 70001 --jn 7 --n-bin 5
run "k25+cov(2 engines)" --n-indv 60000 --n-snp 40000 --jn 10 --k 25 --n-cov 3
run "genie+cov(2 engines)" --n-indv 60000 --n-snp 40000 --jn 10 --genie G+GxE --n-cov 5 --n-bin 2
run "k40" --n-indv 30000 --n-snp 30000 --jn 6 --k 40
run "missing+dropped" --n-indv 80000 --n-snp 80000 --jn 10 --missing-rate 0.01 --drop-indv 777
run "missing+8bins+cov" --n-indv 60000 --n-snp 60000 --jn 10 --n-bin 8 --n-cov 6 --missing-rate 0.02 --drop-indv 5
run "missing+genie" --n-indv 50000 --n-snp 50000 --jn 5 --genie G+GxE --missing-rate 0.005
run "dom" --n-indv 60000 --n-snp 60000 --jn 10 --dom
run "dom+3bins+cov+missing" --n-indv 40000 --n-snp 50000 --jn 5 --dom --n-bin 3 --n-cov 4 --missing-rate 0.01
run "dom-long-blocks" --n-indv 20000 --n-snp 150000 --jn 2 --dom
run "dom-N200k-packed" --n-indv 200000 --n-snp 40000 --jn 4 --dom --missing-rate 0.01 --drop-indv 100
# round 3: covariates as cross columns (default; ONE int8 digit from N = 16 384 on) against the materialised U Z sketch columns --
# two forms of the covariate terms of the normal equations, both in i8x3
runm() { tag="$1"; shift; for m in "" "--materialize-xxuz"; do timeout -k 10 300 python bench.py $m "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-alt --no-cross-check --verify 2>/dev/null | tail -1 | python -c "$fmt" "$tag/${m:-cross}"; done; }
runm "cross-vs-UZ: 8bins+10cov N200k" --n-indv 200000 --n-snp 60000 --jn 10 --n-bin 8 --n-cov 10
runm "cross-vs-UZ: 3bins+15cov N20k" --n-indv 20000 --n-snp 50000 --jn 5 --n-bin 3 --n-cov 15
runm "cross-vs-UZ: 1bin+4cov N5k (3 digits)" --n-indv 5000 --n-snp 20000 --jn 10 --n-cov 4
runm "cross-vs-UZ: genie+cov" --n-indv 60000 --n-snp 40000 --jn 10 --genie G+GxE --n-cov 5 --n-bin 2
runm "cross-vs-UZ: genie-nxe+cov Q4" --n-indv 40000 --n-snp 40000 --jn 10 --genie G+GxE+NxE --n-cov 4 --compat reference
runm "cross-vs-UZ: dom+3bins+cov+missing" --n-indv 40000 --n-snp 50000 --jn 5 --dom --n-bin 3 --n-cov 4 --missing-rate 0.01
runm "cross-vs-UZ: k25+cov" --n-indv 60000 --n-snp 40000 --jn 10 --k 25 --n-cov 3
runm "cross-vs-UZ: Q4 streaming 4bins+6cov" --n-indv 100000 --n-snp 50000 --jn 10 --n-bin 4 --n-cov 6 --compat reference
