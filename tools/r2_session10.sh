#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
B="timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-alt --no-cpu-baseline"
show() { python -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; c=d['config'].get('cross_check') or {}; print('$2', round(d['ms_per_step'],2), r['gemm1_ms_per_step'], r['gemm2_ms_per_step'], c.get('sigma_rel'), c.get('se_rel'))"; }
for v in 1 0; do
  $B --opt g2_wide_loads=$v > $O/w2_c3_$v.json 2> $O/w2_c3_$v.err || { echo "c3 failed"; tail -n 5 $O/w2_c3_$v.err; exit 1; }
  show $O/w2_c3_$v.json "C3 w2=$v"
  $B --opt g2_wide_loads=$v --n-bin 8 --n-cov 10 > $O/w2_c4_$v.json 2> $O/w2_c4_$v.err && show $O/w2_c4_$v.json "C4 w2=$v"
  $B --opt g2_wide_loads=$v --genie G+GxE > $O/w2_c5_$v.json 2> $O/w2_c5_$v.err && show $O/w2_c5_$v.json "C5 w2=$v"
  $B --opt g2_wide_loads=$v --n-indv 50000 --n-snp 500000 > $O/w2_c2_$v.json 2> $O/w2_c2_$v.err && show $O/w2_c2_$v.json "C2 w2=$v"
  $B --opt g2_wide_loads=$v --missing-rate 0.005 > $O/w2_miss_$v.json 2> $O/w2_miss_$v.err && show $O/w2_miss_$v.json "C3 missing w2=$v"
done
timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=25 > $O/tests10.log 2>&1
rc=$?; echo "pytest rc $rc"; grep -E "passed|failed|^FAILED|^E  " $O/tests10.log | cut -c1-300 | tail -n 30
