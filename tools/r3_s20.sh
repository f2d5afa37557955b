#!/bin/bash
# where do the 1.40 fetched bytes per algorithmic byte of the five-tile GEMM 1 come from?  FETCH_SIZE + L2 hit / miss counts per variant
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3/pmc_g1v
rm -rf $OUT && mkdir -p $OUT
SMALL="--n-snp 100000 --jn 10 --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --genie G+GxE"
run() { d=$1; o=$2; shift; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$d -o p --output-format csv -- python3 bench.py $SMALL $o > $OUT/$d.json 2> $OUT/$d.err; echo "== $d"; python3 tools/pmc_summary.py $OUT/$d | grep -A6 "gemm1"; }
for v in "base:--opt g1_line=0" "line:--opt g1_line=1" "xcd:--opt g1_line=0 --opt g1_split_major=2" "xcd4k:--opt g1_line=0 --opt g1_split_major=2 --opt g1_workgroups=4096"; do
  n=${v%%:*}; o=${v#*:}
  run ${n}_fetch "$o" FETCH_SIZE
  run ${n}_tcc "$o" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
done
