#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
B="timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-alt --no-cpu-baseline"
show() { python -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', round(d['ms_per_step'],2), r['gemm1_ms_per_step'], r['gemm2_ms_per_step'], r['kernels'], d['config'].get('cross_check'))"; }
for v in 0 1; do
  $B --opt g1_deep=$v > $O/deep_c3_$v.json 2> $O/deep_c3_$v.err || { echo "c3 deep=$v failed"; tail -n 5 $O/deep_c3_$v.err; exit 1; }
  show $O/deep_c3_$v.json "C3 deep=$v"
  $B --opt g1_deep=$v --n-bin 8 --n-cov 10 > $O/deep_c4_$v.json 2> $O/deep_c4_$v.err || { echo "c4 deep=$v failed"; tail -n 5 $O/deep_c4_$v.err; exit 1; }
  show $O/deep_c4_$v.json "C4 deep=$v"
  $B --opt g1_deep=$v --genie G+GxE > $O/deep_c5_$v.json 2> $O/deep_c5_$v.err || { echo "c5 deep=$v failed"; tail -n 5 $O/deep_c5_$v.err; exit 1; }
  show $O/deep_c5_$v.json "C5 deep=$v"
  $B --opt g1_deep=$v --n-indv 50000 --n-snp 500000 > $O/deep_c2_$v.json 2> $O/deep_c2_$v.err || { echo "c2 deep=$v failed"; tail -n 5 $O/deep_c2_$v.err; exit 1; }
  show $O/deep_c2_$v.json "C2 deep=$v"
done
