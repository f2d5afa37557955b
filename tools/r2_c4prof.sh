#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2/c4prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o c4 --output-format csv -- python3 bench.py --n-bin 8 --n-cov 10 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check > $O/bench.json 2> $O/bench.err
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r["Name"][:90], r["Calls"], round(int(r["TotalDurationNs"])/1e6,2), "ms total", round(float(r["AverageNs"])/1e3,1), "us avg")
PY
