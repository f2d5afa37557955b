#!/bin/bash
# A/B runner: bench.py over {configurations} x {environments} x {option sets} on ONE box, one summary line per run.
# Every kernel / launch-shape experiment of DESIGN.md 3.4 is an invocation of this script (tools/README.md lists them).
#
#   tools/ab.sh [-c CONFIG]... [-o "OPTS"]... [-e "ENV=VAL ..."]... [-s STEPS] [-w WARMUP] [-r REPS] [-x "EXTRA"] [-t "PYTEST -k EXPR"] [-l TAG]
#
#   -c  a named configuration (C2 C3 C4 C5 C5cov DOM DOMC4 RANK8) or literal bench.py arguments ("--n-bin 32 --n-cov 5"); default C3
#   -o  one option set per flag, e.g. -o "" -o "--opt g1_compact=0"  (bench.py --opt name=value = pyrhe_engine_set_option); default ""
#   -e  one environment per flag, e.g. -e "" -e "PYRHE_WHOLE_TABLES=1 PYRHE_CAPACITY=196608"; default ""
#   -x  extra bench.py arguments for every run (e.g. "--verify --h2 0.3" prints sigma^2 / SE beside the times)
#   -t  run these GPU tests first and stop if they fail (the parity gate of an experiment)
#   -l  tag: the JSON lines are also kept under gpurun_out/ab/TAG/
# Prints: label | M SNPs/s | ms per pass | GEMM 1 / GEMM 2 / mid kernels ms per pass | kernels | pass 2 s | [sigma^2, SE with --verify]
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
CONFIGS=(); OPTS=(); ENVS=(); STEPS=4; WARM=2; REPS=1; EXTRA=""; TESTS=""; TAG=""
while getopts "c:o:e:s:w:r:x:t:l:" f; do
  case $f in
    c) CONFIGS+=("$OPTARG");; o) OPTS+=("$OPTARG");; e) ENVS+=("$OPTARG");; s) STEPS=$OPTARG;; w) WARM=$OPTARG;; r) REPS=$OPTARG;;
    x) EXTRA=$OPTARG;; t) TESTS=$OPTARG;; l) TAG=$OPTARG;; *) sed -n 2,14p "$0"; exit 2;;
  esac
done
[ ${#CONFIGS[@]} -eq 0 ] && CONFIGS=("C3"); [ ${#OPTS[@]} -eq 0 ] && OPTS=(""); [ ${#ENVS[@]} -eq 0 ] && ENVS=("")
named() {      # a configuration name, optionally followed by more bench.py arguments ("C2 --n-bin 8 --n-cov 10")
  local first=${1%% *} rest=""
  [ "$first" != "$1" ] && rest=" ${1#* }"
  case "$first" in
    C2) echo "--n-indv 50000 --n-snp 500000$rest";; C3) echo "$rest";; C4) echo "--n-bin 8 --n-cov 10$rest";; C5) echo "--genie G+GxE$rest";;
    C5cov) echo "--genie G+GxE --n-cov 10$rest";; DOM) echo "--dom$rest";; DOMC4) echo "--dom --n-bin 8 --n-cov 10$rest";;
    RANK8) echo "--n-snp 130000 --jn 13$rest";;      # rank 0's shard of an 8-GPU job: 13 of the 100 blocks
    *) echo "$1";;
  esac
}
if [ -n "$TESTS" ]; then
  python -m pytest tests -x -q -m gpu -k "$TESTS" 2>&1 | tail -3 || exit 1
fi
[ -n "$TAG" ] && mkdir -p "gpurun_out/ab/$TAG"
fmt='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; e=c.get("end_to_end") or {}
k=r.get("kernels") or {}
out=[sys.argv[1].ljust(64), "%7.2f M" % (d["value"]/1e6), "%7.2f ms" % d["ms_per_step"], "g1 %s g2 %s mid %s" % (r.get("gemm1_ms_per_step"), r.get("gemm2_ms_per_step"), r.get("mid_ms_per_step")),
     "%s + %s" % (k.get("gemm1"), k.get("gemm2")), "frac %s pass_frac %s" % (r.get("frac"), r.get("pass_frac")), "pass2 %s s" % e.get("pass2_s")]
if "sigma_total" in c: out += [repr(c["sigma_total"]), repr(c.get("sigma_se"))]
print(" | ".join(out))'
n=0
for rep in $(seq 1 "$REPS"); do for cfg in "${CONFIGS[@]}"; do for ev in "${ENVS[@]}"; do for opt in "${OPTS[@]}"; do
  args=$(named "$cfg"); n=$((n+1))
  line=$(env $ev timeout -k 10 400 python bench.py --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline --no-alt --no-cross-check $args $opt $EXTRA 2>/dev/null | tail -1) || { echo "$cfg [$ev] [$opt]: bench.py failed"; exit 1; }
  [ -n "$TAG" ] && echo "$line" > "gpurun_out/ab/$TAG/run$n.json"
  echo "$line" | python -c "$fmt" "$cfg${ev:+ {$ev\}}${opt:+ [$opt]}"
done; done; done; done
