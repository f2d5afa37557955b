#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
B="timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-alt --no-cpu-baseline"
show() { python -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; c=d['config'].get('cross_check') or {}; print('$2', round(d['ms_per_step'],2), r['gemm1_ms_per_step'], r['gemm2_ms_per_step'], r['kernels']['gemm1'], c.get('sigma_rel'), c.get('se_rel'))"; }
for g in 1536 768 1152 3072; do
  $B --opt g1_wide=1 --opt g1_workgroups=$g > $O/wide_$g.json 2> $O/wide_$g.err || { echo "wide wgs=$g failed"; tail -n 5 $O/wide_$g.err; exit 1; }
  show $O/wide_$g.json "C3 wide wgs=$g"
done
$B --opt g1_wide=1 --n-bin 8 --n-cov 10 > $O/wide_c4.json 2> $O/wide_c4.err && show $O/wide_c4.json "C4 wide"
$B --opt g1_wide=1 --n-indv 50000 --n-snp 500000 > $O/wide_c2.json 2> $O/wide_c2.err && show $O/wide_c2.json "C2 wide"
$B --opt g1_wide=1 --n-indv 20000 --n-snp 50000 --jn 5 --n-bin 3 --missing-rate 0.01 --verify > $O/wide_miss.json 2> $O/wide_miss.err && show $O/wide_miss.json "missing wide"
