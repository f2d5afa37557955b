#!/bin/bash
# Round-2 evidence at HEAD (run through gpurun from the repo root); copies what is to be judged into profiles/ afterwards
# (tools/collect_r02_copy.py, on the builder's side).
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the driver's line)
#   2. PMC passes (own runs, one counter group each, --kernel-trace only) on a tenth of the cohort, same 10 000-SNP blocks
#   3. the BASELINE configurations on one GPU;  4. the cross-check sweep of unusual shapes
set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02_final
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python3 bench.py --steps 3 --warmup 2 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace done"
python3 bench.py --steps 3 --warmup 2 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
echo "plain bench done"
SMALL="--n-snp 100000 --jn 10 --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check"
run() { d=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$d -o p --output-format csv -- python3 bench.py $SMALL > $OUT/pmc_$d.json 2> $OUT/pmc_$d.err; python3 tools/pmc_summary.py $OUT/pmc_$d > $OUT/pmc_$d.summary.txt; echo "$d done"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES
run sq2 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE
bash tools/run_configs.sh > $OUT/configs.txt 2>&1; cp gpurun_out/configs/c*.json $OUT/
echo "configs done"; cat $OUT/configs.txt
bash tools/verify_sweep.sh > $OUT/sweep.log 2>&1; python3 tools/sweep_summary.py $OUT/sweep.log > $OUT/sweep_summary.txt; cat $OUT/sweep_summary.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
tail -c 2500 $OUT/bench_plain.json
