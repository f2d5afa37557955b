#!/bin/bash
# round 3, session 4: RHE_DOM tests in the segmented flow, A/B against the per-list flow, C2 / C5 at HEAD
set -o pipefail
mkdir -p gpurun_out/r3
python -m pytest tests/test_headline_gpu.py tests/test_families_gpu.py tests/test_midsize_gpu.py tests/test_models_gpu.py -x -q -k "dom or DOM" > gpurun_out/r3/t4.log 2>&1; echo "dom tests rc=$?"; tail -5 gpurun_out/r3/t4.log
python bench.py --dom --no-cpu-baseline --no-alt --no-cross-check --steps 2 --warmup 1 --opt dom_seg=0 > gpurun_out/r3/dom_perlist.json 2> gpurun_out/r3/dom_perlist.err; echo "dom per-list rc=$?"
python bench.py --dom --no-cpu-baseline --no-alt --no-cross-check --steps 2 --warmup 1 --n-bin 8 --n-cov 10 --opt dom_seg=0 > gpurun_out/r3/dom_c4_perlist.json 2> gpurun_out/r3/dom_c4_perlist.err; echo "dom c4 per-list rc=$?"
python bench.py --n-indv 50000 --n-snp 500000 --no-cpu-baseline --no-alt --no-cross-check --steps 5 --warmup 2 > gpurun_out/r3/c2.json 2> gpurun_out/r3/c2.err; echo "c2 rc=$?"
python bench.py --genie G+GxE --no-cpu-baseline --no-alt --no-cross-check --steps 3 --warmup 2 > gpurun_out/r3/c5.json 2> gpurun_out/r3/c5.err; echo "c5 rc=$?"
python bench.py --genie G+GxE --n-cov 10 --no-cpu-baseline --no-alt --no-cross-check --steps 3 --warmup 2 > gpurun_out/r3/c5_cov.json 2> gpurun_out/r3/c5_cov.err; echo "c5 cov rc=$?"
python bench.py --genie G+GxE --n-cov 10 --materialize-xxuz --no-cpu-baseline --no-alt --no-cross-check --steps 3 --warmup 2 > gpurun_out/r3/c5_cov_mat.json 2> gpurun_out/r3/c5_cov_mat.err; echo "c5 cov mat rc=$?"
for f in dom dom_perlist dom_c4 dom_c4_perlist c2 c5 c5_cov c5_cov_mat; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3/$f.json")); r=d["roofline"]
    print("%-16s %7.2f M SNPs/s %7.2f ms  g1 %6.2f g2 %6.2f mid %5.2f  %s %s  e2e %s" % ("$f", d["value"]/1e6, d["ms_per_step"], r["gemm1_ms_per_step"], r["gemm2_ms_per_step"], r["mid_ms_per_step"], r["kernels"]["gemm1"], r["kernels"]["gemm2"], d["config"].get("end_to_end")))
except Exception as ex:
    print("$f", "failed", ex)
PY
done
