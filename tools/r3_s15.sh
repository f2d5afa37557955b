#!/bin/bash
# PMC counters of pass 2's block-store kernel (blk_multi_dots) at 8 bins + covariates: where do its waves spend their time?
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3/pmc_dots; rm -rf $OUT; mkdir -p $OUT
ARGS="--n-bin 8 --n-cov 10 --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check"
run() { d=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$d -o p --output-format csv -- python3 bench.py $ARGS > $OUT/$d.json 2> $OUT/$d.err; python3 tools/pmc_summary.py $OUT/$d | grep -A 12 "blk_multi_dots" > $OUT/$d.summary.txt; cat $OUT/$d.summary.txt; }
run sq1 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES
run sq2 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE
run sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
