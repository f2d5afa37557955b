#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
B="timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-alt --no-cpu-baseline"
show() { python -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; c=d['config'].get('cross_check') or {}; print('$2', round(d['ms_per_step'],2), r['gemm1_ms_per_step'], r['gemm2_ms_per_step'], c.get('sigma_rel'), c.get('se_rel'))"; }
for w in 4 6 12; do for g in 768 1536 3072; do
  $B --opt g1_waves=$w --opt g1_workgroups=$g > $O/waves_${w}_$g.json 2> $O/waves_${w}_$g.err || { echo "waves=$w wgs=$g failed"; tail -n 5 $O/waves_${w}_$g.err; exit 1; }
  show $O/waves_${w}_$g.json "C3 waves=$w wgs=$g"
done; done
for w in 6 12; do
  $B --opt g1_waves=$w --n-bin 8 --n-cov 10 > $O/waves_c4_$w.json 2> $O/waves_c4_$w.err && show $O/waves_c4_$w.json "C4 waves=$w"
  $B --opt g1_waves=$w --genie G+GxE > $O/waves_c5_$w.json 2> $O/waves_c5_$w.err && show $O/waves_c5_$w.json "C5 waves=$w"
  $B --opt g1_waves=$w --n-indv 50000 --n-snp 500000 > $O/waves_c2_$w.json 2> $O/waves_c2_$w.err && show $O/waves_c2_$w.json "C2 waves=$w"
done
export PYRHE_LIB=$PWD/pyrhe_amd/csrc/libpyrhe_sketch_diag.so; python tools/g1_phases.py g1_waves=12
