"""cProfile of SketchRunner.pass2 on a synthetic cohort (diagnostic; GPU box only)."""
import cProfile, pstats, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrhe_amd import engine as eng_mod
from pyrhe_amd.streaming import BlockPlan, SketchRunner

N, M, jn, k = 500_000, int(os.environ.get("M", 200_000)), 100, 10
nbin, ncov = int(os.environ.get("NBIN", 1)), int(os.environ.get("NCOV", 0))
dev = torch.device("cuda:0")
geno = eng_mod.synth_genotypes(M, N, 1, 0, dev)
annot = np.zeros((M, nbin), np.int8); annot[np.arange(M), np.arange(M) % nbin] = 1
plan = BlockPlan(annot, jn)
np.random.seed(0); Z = np.random.randn(N, k)
y = np.random.default_rng(1).normal(size=N); y -= y.mean()
cov = None
if ncov:
    cov = np.concatenate([np.ones((N, 1)), np.random.default_rng(2).normal(size=(N, ncov - 1))], axis=1)
engine = eng_mod.SketchEngine(N, N, max_entries=plan.max_block, device=dev)
runner = SketchRunner(engine, geno, plan, (0, jn), n_orig=N, missing_indv=[], Z=Z, cov=cov, seed=0, streaming=True,
                      compat="fixed" if cov is not None else "reference")
runner.begin_trait(y, check_missing=False)
runner.pass1(); torch.cuda.synchronize()
t0 = time.perf_counter(); runner.pass1(); torch.cuda.synchronize(); print("pass1 s", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); runner.pass2(); torch.cuda.synchronize(); print("pass2 s", time.perf_counter() - t0)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)

if os.environ.get("TRACE_DOTS"):
    orig = runner._dots
    def traced(x, y, n_sub, batch):
        torch.cuda.synchronize(); t = time.perf_counter()
        ld = runner.eng.loo_dots
        def ld2(*a, **k):
            t1 = time.perf_counter(); r = ld(*a, **k); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
            print(f"    loo_dots enqueue {(t2-t1)*1e3:.2f} ms, gpu wait {(t3-t2)*1e3:.2f} ms")
            return r
        runner.eng.loo_dots = ld2
        out = orig(x, y, n_sub, batch)
        runner.eng.loo_dots = ld
        torch.cuda.synchronize()
        print(f"_dots nx={len(x)} ny={len(y)} n_sub={n_sub} batch={batch} blkx={sum(r is not None and r.blk is not None for r in x)} totx={sum(r is not None and r.tot is not None for r in x)}: {(time.perf_counter()-t)*1e3:.2f} ms")
        return out
    runner._dots = traced
    t0 = time.perf_counter(); runner.pass2(); torch.cuda.synchronize(); print("pass2 traced s", time.perf_counter() - t0)
