"""cProfile of SketchRunner.pass1 on a synthetic cohort (diagnostic; GPU box only): host-side time per pass."""
import cProfile, pstats, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrhe_amd import engine as eng_mod
from pyrhe_amd.streaming import BlockPlan, SketchRunner, engine_capacity

N, M, jn, k = int(os.environ.get("N", 500_000)), int(os.environ.get("M", 200_000)), int(os.environ.get("JN", 20)), 10
nbin, ncov = int(os.environ.get("NBIN", 8)), int(os.environ.get("NCOV", 10))
dev = torch.device("cuda:0")
geno = eng_mod.synth_genotypes(M, N, 1, 0, dev)
if os.environ.get("MISSING"):
    for a in range(0, M, 4096):
        part = geno[a:a + 4096]
        hit = torch.rand(part.shape, device=dev) < 0.02
        part[hit] = (part[hit] & 0xFC) | 0x01
    del part, hit
eng_mod.recode_rows(geno, "value")
annot = np.zeros((M, nbin), np.int8); annot[np.arange(M), np.arange(M) % nbin] = 1
plan = BlockPlan(annot, jn)
np.random.seed(0); Z = np.random.randn(N, k)
y = np.random.default_rng(1).normal(size=N); y -= y.mean()
cov = np.concatenate([np.ones((N, 1)), np.random.default_rng(2).normal(size=(N, ncov - 1))], axis=1) if ncov else None
engine = eng_mod.SketchEngine(N, N, max_entries=engine_capacity(plan), device=dev, sub_snps=engine_capacity(plan))
engine.set_coding("value")
runner = SketchRunner(engine, geno, plan, (0, jn), n_orig=N, missing_indv=[], Z=Z, cov=cov, seed=0, streaming=True,
                      compat="fixed" if cov is not None else "reference")
runner.begin_trait(y, check_missing=bool(os.environ.get("MISSING")))
for _ in range(2):
    runner.pass1(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); runner.pass1(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((round((t1 - t0) * 1e3, 1), round((t2 - t0) * 1e3, 1)))
print("pass1 (host enqueue ms, total ms):", ts)
pr = cProfile.Profile(); pr.enable(); runner.pass1(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
