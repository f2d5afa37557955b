#!/bin/bash
# Memory-path counters (TA / TCP / UTCL1 / TCC) of the headline pass on a tenth of the cohort, one SMALL --pmc group per run
# (five counters of one block exceed what the hardware collects at once: rocprofv3 then aborts and the process hangs).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_mem
rm -rf $OUT && mkdir -p $OUT
SMALL="--n-snp 100000 --jn 10 --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check $*"
run() { d=$1; shift; timeout -k 5 75 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$d -o p --output-format csv -- python3 bench.py $SMALL > $OUT/$d.json 2> $OUT/$d.err; echo "== $d rc $?"; python3 tools/pmc_summary.py $OUT/$d > $OUT/$d.summary.txt; grep -A6 "gemm1_i8\|gemm2_i8p" $OUT/$d.summary.txt; grep -m1 "exceeds" $OUT/$d.err; }
run ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
run utcl1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
run busy GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum
