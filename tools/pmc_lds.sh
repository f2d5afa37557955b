set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_lds
rm -rf $OUT && mkdir -p $OUT
SMALL="--n-snp 100000 --jn 10 --steps 1 --warmup 1 --no-cpu-baseline --no-alt"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL -d $OUT/a -o p --output-format csv -- python3 bench.py $SMALL > $OUT/a.json 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD -d $OUT/b -o p --output-format csv -- python3 bench.py $SMALL > $OUT/b.json 2> $OUT/b.err
for d in a b; do python3 tools/pmc_summary.py $OUT/$d > $OUT/$d.summary.txt; done
grep -A8 "gemm1_i8" $OUT/*.summary.txt
