#!/bin/bash
# kernel trace of RHE_DOM with 8 bins + 10 covariates (pass 2 takes 32 ms there: where?)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3/prof_domc4; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o t --output-format csv -- python3 bench.py --dom --n-bin 8 --n-cov 10 --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check > $O/bench.json 2> $O/bench.err
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(5), f"{float(r['TotalDurationNs'])/1e6:9.2f} ms", f"{float(r['AverageNs'])/1e3:9.1f} us")
PY
