#!/bin/bash
# Round-2 GPU session 1: the whole GPU suite, the default bench line, GEMM 1 launch-shape knobs.
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=25 -s > $O/tests1.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -n 40 $O/tests1.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 3 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err || { echo "bench failed"; tail -n 20 $O/bench_default.err; exit 1; }
cat $O/bench_default.json
B="timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-alt --no-cpu-baseline --no-cross-check"
for v in "g1_split_major=1" "g1_workgroups=708" "g1_workgroups=768" "g1_workgroups=1180" "g1_workgroups=2304" "g1_workgroups=3072"; do
  $B --opt $v > $O/knob_$v.json 2> $O/knob_$v.err || { echo "knob $v failed"; tail -n 5 $O/knob_$v.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/knob_$v.json')); r=d['roofline']; print('$v', round(d['ms_per_step'],2), r['gemm1_ms_per_step'], r['gemm2_ms_per_step'])"
done
PYRHE_LIB=$PWD/pyrhe_amd/csrc/libpyrhe_sketch_diag.so $B --opt diag_g1_plain_store=1 > $O/knob_plainstore.json 2> $O/knob_plainstore.err && \
  python -c "import json; d=json.load(open('$O/knob_plainstore.json')); r=d['roofline']; print('plain stores (diag, wrong results)', round(d['ms_per_step'],2), r['gemm1_ms_per_step'], r['gemm2_ms_per_step'])"
