#!/usr/bin/env python3
"""Cycle account of GEMM 1's loop phases (diagnostic library: PYRHE_LIB=pyrhe_amd/csrc/libpyrhe_sketch_diag.so).
One sketch pass at the headline shape on a tenth of the cohort; every wave adds the s_memtime cycles it spent in
(0) waiting for its packed words, (1) issuing the next step's loads, (2) unpack + matrix instructions, (3) waiting for the
next R slice, (4) the step barrier."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrhe_amd import engine as E
from pyrhe_amd.streaming import BlockPlan, SketchRunner, engine_capacity

extra = [a for a in sys.argv[1:]]
N, M, jn, k = 500_000, 100_000, 10, 10
dev = torch.device("cuda", 0)
geno = E.recode_rows(E.synth_genotypes(M, N, 20251212, 0, dev), "value")
plan = BlockPlan(np.ones((M, 1), np.int8), jn)
eng = E.SketchEngine(N, N, max_entries=engine_capacity(plan), precision="i8x3", device=dev, sub_snps=engine_capacity(plan))
eng.set_coding("value")
for a in extra:
    name, val = a.split("=")
    eng.set_option(name, int(val))
np.random.seed(0)
Z = np.random.randn(N, k)
y = np.random.default_rng(1).normal(size=N); y -= y.mean()
r = SketchRunner(eng, geno, plan, (0, jn), n_orig=N, missing_indv=[], Z=Z, seed=0, streaming=True)
r.begin_trait(y, check_missing=False)
r.pass1()
torch.cuda.synchronize()
stats = torch.zeros(8, dtype=torch.int64, device=dev)
eng.set_option("diag_g1_stats", stats.data_ptr())
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); r.pass1(); t1.record(); torch.cuda.synchronize()
eng.set_option("diag_g1_stats", 0)
s = stats.cpu().numpy().astype(np.float64)
tot = s[:5].sum()
names = ["wait packed words", "issue next loads", "unpack + MFMA", "wait R slice", "step barrier"]
print(f"pass {t0.elapsed_time(t1):.2f} ms, {int(s[5])} waves, options {extra}")
for n_, v in zip(names, s[:5]):
    print(f"  {n_:20s} {100 * v / tot:5.1f} %   {v / s[5]:12.0f} cycles per wave")
