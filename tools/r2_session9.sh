#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=25 > $O/tests9.log 2>&1
rc=$?; echo "pytest rc $rc"; grep -E "passed|failed|^FAILED|^E  " $O/tests9.log | cut -c1-300 | tail -n 30
if [ $rc -ge 124 ]; then exit $rc; fi
B="timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-alt --no-cpu-baseline --no-cross-check"
show() { python -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', round(d['ms_per_step'],2), r['gemm1_ms_per_step'], r['gemm2_ms_per_step'], r['kernels']['gemm1'])"; }
for g in 1536 2304 4608; do $B --opt g1_workgroups=$g > $O/wide2_$g.json 2> $O/wide2_$g.err && show $O/wide2_$g.json "C3 wgs=$g"; done
for n in 200000 300000; do for w in 0 1; do $B --n-indv $n --n-snp 500000 --opt g1_wide=$w > $O/wide_n${n}_$w.json 2> $O/wide_n${n}_$w.err && show $O/wide_n${n}_$w.json "N=$n wide=$w"; done; done
$B --opt g1_wide=1 --missing-rate 0.005 > $O/wide_missing.json 2>$O/wide_missing.err && show $O/wide_missing.json "C3 missing wide=1"
$B --opt g1_wide=0 --missing-rate 0.005 > $O/narrow_missing.json 2>$O/narrow_missing.err && show $O/narrow_missing.json "C3 missing wide=0"
