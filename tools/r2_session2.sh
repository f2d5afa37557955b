#!/bin/bash
# Round-2 GPU session 2: suite again, BASELINE configurations with the kernel split, GEMM 1 priority knob, fused skeleton.
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out/r2 && O=gpurun_out/r2
timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=25 -s > $O/tests2.log 2>&1
rc=$?; echo "pytest rc $rc"; grep -E "^\[|passed|failed|^FAILED|^E  " $O/tests2.log | cut -c1-300 | tail -n 60
if [ $rc -ge 124 ]; then exit $rc; fi
B="timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-alt --no-cpu-baseline --no-cross-check"
for v in "g1_tune=1" "g1_workgroups=6144"; do
  $B --opt $v > $O/knob_$v.json 2> $O/knob_$v.err || { echo "knob $v failed"; tail -n 5 $O/knob_$v.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/knob_$v.json')); r=d['roofline']; print('$v', round(d['ms_per_step'],2), r['gemm1_ms_per_step'], r['gemm2_ms_per_step'])"
done
bash tools/run_configs.sh 2>&1 | tail -n 12
timeout -k 10 200 tools/bin/fused_skel2 > $O/fused_skel2.txt 2>&1; echo "skel rc $?"; cat $O/fused_skel2.txt
