#!/bin/bash
# the slow FIRST run of N = 50k / 8 bins + covariates on a fresh box: where are the gaps?  (kernel trace of the first process)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3/first_run; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python3 bench.py --n-indv 50000 --n-snp 500000 --n-bin 8 --n-cov 10 --steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check > $O/bench.json 2> $O/bench.err
python3 - <<PY
import csv, glob, json
print("ms_per_step", json.loads(open("$O/bench.json").read().strip().splitlines()[-1])["ms_per_step"])
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
gaps = []
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps.append((g, a["Kernel_Name"][:50], b["Kernel_Name"][:50], int(b["Start_Timestamp"])))
t0 = int(rows[0]["Start_Timestamp"])
# the timed region = the last ~6 passes: look at gaps in the last third of the trace
tl = int(rows[-1]["End_Timestamp"])
late = [g for g in gaps if g[3] > t0 + (tl - t0) * 0.5]
for g in sorted(late, reverse=True)[:12]:
    print(f"gap {g[0]/1e6:8.2f} ms at {(g[3]-t0)/1e9:7.3f} s  after {g[1]}  before {g[2]}")
PY
python3 bench.py --n-indv 50000 --n-snp 500000 --n-bin 8 --n-cov 10 --steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-cross-check 2>/dev/null | tail -1 | python3 -c "import json,sys; print('second process ms_per_step', json.loads(sys.stdin.read())['ms_per_step'])"
