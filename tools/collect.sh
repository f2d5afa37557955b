#!/bin/bash
# Evidence of one round at HEAD, run through gpurun from the repo root:  tools/collect.sh r04
# tools/collect_copy.py r04 <commit> then copies what is to be judged from gpurun_out/r04_final (scratch) into profiles/ (tracked).
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the driver's line) + the plain run
#   2. PMC passes (own runs, one counter group each, --kernel-trace only) on a tenth of the cohort: headline, config 4, config 5
#   3. the BASELINE configurations on one GPU + kernel traces of configs 2, 4, 5, RHE_DOM;  4. the cross-check sweep
# (rocprofv3 is always given the program itself after `--`: no env / bash -c hop behind the profiler's preloaded library)
set -e
TAG=${1:?usage: tools/collect.sh <round tag, e.g. r04>}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
[ -f bench.py ] || { echo "not at the repo root"; exit 1; }
OUT="gpurun_out/${TAG}_final"
mkdir -p gpurun_out && rm -rf -- "./$OUT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python3 bench.py --steps 3 --warmup 2 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace done"
python3 bench.py --steps 3 --warmup 2 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
echo "plain bench done"
SMALL="--n-snp 100000 --jn 10 --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check"
run() { d=$1; cfg=$2; shift 2; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$d -o p --output-format csv -- python3 bench.py $SMALL $cfg > $OUT/pmc_$d.json 2> $OUT/pmc_$d.err; python3 tools/pmc_summary.py $OUT/pmc_$d > $OUT/pmc_$d.summary.txt; echo "$d done"; }
run fetch "" FETCH_SIZE
run write "" WRITE_SIZE
run sq1 "" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES
run c4_fetch "--n-bin 8 --n-cov 10" FETCH_SIZE
run c4_write "--n-bin 8 --n-cov 10" WRITE_SIZE
run c4_sq1 "--n-bin 8 --n-cov 10" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES
run c5_fetch "--genie G+GxE" FETCH_SIZE
run c5_write "--genie G+GxE" WRITE_SIZE
bash tools/run_configs.sh > $OUT/configs.txt 2>&1; cp gpurun_out/configs/c*.json $OUT/
echo "configs done"; cat $OUT/configs.txt
prof() { tag=$1; shift; O=$OUT/prof_$tag; mkdir -p $O; rocprofv3 --kernel-trace --stats -d $O -o $tag --output-format csv -- python3 bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check > $O/bench.json 2> $O/bench.err; echo "prof $tag done"; }
prof c4 --n-bin 8 --n-cov 10
prof c5 --genie G+GxE
prof c2 --n-indv 50000 --n-snp 500000
prof dom --dom
python3 bench.py --dom --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --n-bin 8 --n-cov 10 > $OUT/dom_c4.json 2> $OUT/dom_c4.err || true
bash tools/verify_sweep.sh > $OUT/sweep.log 2>&1; python3 tools/sweep_summary.py $OUT/sweep.log > $OUT/sweep_summary.txt; cat $OUT/sweep_summary.txt
tail -c 3000 $OUT/bench_plain.json
