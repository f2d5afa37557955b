#!/bin/bash
# rocprofv3 kernel trace of one BASELINE configuration: tools/r2_cfgprof.sh TAG <bench.py arguments>
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=$1; shift
O=gpurun_out/r2/prof_$tag; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o $tag --output-format csv -- python3 bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check > $O/bench.json 2> $O/bench.err
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:80], r["Calls"], round(int(r["TotalDurationNs"])/1e6,2), "ms total", round(float(r["AverageNs"])/1e3,1), "us avg")
PY
