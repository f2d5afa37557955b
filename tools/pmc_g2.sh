set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_g2p
rm -rf $OUT && mkdir -p $OUT
SMALL="--n-snp 100000 --jn 10 --steps 1 --warmup 1 --no-cpu-baseline --no-alt"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $OUT/a -o p --output-format csv -- python3 bench.py $SMALL > $OUT/a.json 2> $OUT/a.err
echo a done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $OUT/b -o p --output-format csv -- python3 bench.py $SMALL > $OUT/b.json 2> $OUT/b.err
echo b done
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE FETCH_SIZE -d $OUT/c -o p --output-format csv -- python3 bench.py $SMALL > $OUT/c.json 2> $OUT/c.err
echo c done
for d in a b c; do python3 tools/pmc_summary.py $OUT/$d > $OUT/$d.summary.txt; done
grep -A8 "gemm2_i8p\|gemm1_i8" $OUT/*.summary.txt
