#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3/prof_c4_f32; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o c4 --output-format csv -- python3 bench.py --n-bin 8 --n-cov 10 --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-cross-check --opt dots_f32=1 > $O/bench.json 2> $O/bench.err
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(r["Name"][:80].ljust(82), r["Calls"].rjust(5), "%9.2f ms total" % (int(r["TotalDurationNs"])/1e6), "%9.1f us avg" % (float(r["AverageNs"])/1e3))
PY
