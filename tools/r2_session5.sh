#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
B="timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-alt --no-cpu-baseline --no-cross-check"
export PYRHE_LIB=$PWD/pyrhe_amd/csrc/libpyrhe_sketch_diag.so
for v in "diag_g1_no_barrier=0" "diag_g1_no_barrier=1" "diag_g1_no_barrier=2" "diag_g1_no_barrier=3"; do
  $B --opt $v > $O/diag_$v.json 2> $O/diag_$v.err || { echo "$v failed"; tail -n 5 $O/diag_$v.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/diag_$v.json')); r=d['roofline']; print('$v', round(d['ms_per_step'],2), r['gemm1_ms_per_step'], r['gemm2_ms_per_step'])"
done
