#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the default bench command (whole pass, N=500k M=1M)
#   2. PMC passes (own runs, one counter group each) on one-tenth of the cohort (same 10 000-SNP blocks)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_final
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace done"
SMALL="--n-snp 100000 --jn 10 --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- python3 bench.py $SMALL > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p --output-format csv -- python3 bench.py $SMALL > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES -d $OUT/pmc_sq1 -o p --output-format csv -- python3 bench.py $SMALL > $OUT/pmc_sq1.json 2> $OUT/pmc_sq1.err
echo "sq1 done"
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o p --output-format csv -- python3 bench.py $SMALL > $OUT/pmc_sq2.json 2> $OUT/pmc_sq2.err
echo "sq2 done"
for d in pmc_fetch pmc_write pmc_sq1 pmc_sq2; do python3 tools/pmc_summary.py $OUT/$d > $OUT/$d.summary.txt; done
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
tail -c 1500 $OUT/bench_trace.json
