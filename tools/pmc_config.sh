#!/bin/bash
# PMC passes (own runs, one counter group each, with --kernel-trace only) for one bench configuration on a tenth of the cohort.
# usage: tools/pmc_config.sh <name> <bench args...>      -> gpurun_out/pmc_<name>/*.summary.txt
# PMC_GROUPS="A B C;D E" adds counter groups of your own (one run per group; e.g. the LDS side of GEMM 1, DESIGN.md 3.4:
#   PMC_GROUPS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS;SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY")
# (rocprofv3 is given the program itself after `--`; --pmc never together with anything but --kernel-trace)
set -e
name=${1:?usage: tools/pmc_config.sh <name> <bench args...>}; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
[ -f bench.py ] || { echo "not at the repo root"; exit 1; }
OUT="gpurun_out/pmc_$name"
mkdir -p gpurun_out && rm -rf -- "./$OUT" && mkdir -p "$OUT"
SMALL="--n-snp 100000 --jn 10 --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check $*"
run() { d=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$d -o p --output-format csv -- python3 bench.py $SMALL > $OUT/$d.json 2> $OUT/$d.err; python3 tools/pmc_summary.py $OUT/$d > $OUT/$d.summary.txt; echo "$d done"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES
run sq2 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE
run sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS
if [ -n "$PMC_GROUPS" ]; then
  i=0; IFS=';' read -ra GR <<< "$PMC_GROUPS"
  for g in "${GR[@]}"; do i=$((i+1)); run "user$i" $g; done
fi
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 bench.py $SMALL > $OUT/trace.json 2> $OUT/trace.err
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
cat $OUT/*.summary.txt
