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
NT = int(os.environ.get('NT', 1))
y = np.random.default_rng(1).normal(size=(N, NT)) if NT > 1 else np.random.default_rng(1).normal(size=N); y -= y.mean(axis=0)
cov = None
if ncov:
    cov = np.concatenate([np.ones((N, 1)), np.random.default_rng(2).normal(size=(N, ncov - 1))], axis=1)
engine = eng_mod.SketchEngine(N, N, max_entries=plan.max_block, device=dev)
runner = SketchRunner(engine, geno, plan, (0, jn), n_orig=N, missing_indv=[], Z=Z, cov=cov, seed=0, streaming=True, dom=bool(os.environ.get("DOM")),
                      compat=os.environ.get("COMPAT", "fixed" if cov is not None else "reference"))
runner.begin_trait(y, check_missing=False)
runner.pass1(); torch.cuda.synchronize()
t0 = time.perf_counter(); runner.pass1(); torch.cuda.synchronize(); print("pass1 s", time.perf_counter() - t0)
t0 = time.perf_counter(); runner.pass2(); torch.cuda.synchronize(); print("pass2 (first call) s", time.perf_counter() - t0)
import gc
if os.environ.get("GC_OFF"):
    gc.disable()
_gc_t = {}
def _gc_cb(phase, info):
    if phase == "start": _gc_t["t"] = time.perf_counter()
    else:
        dt = (time.perf_counter() - _gc_t["t"]) * 1e3
        if dt > 5: print(f"   gc gen{info['generation']} took {dt:.1f} ms")
gc.callbacks.append(_gc_cb)
ts = []
for _ in range(12):
    t0 = time.perf_counter(); runner.pass2(); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
print("pass2 x12 ms:", ts)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); runner.pass2(); torch.cuda.synchronize(); print("pass2 s", time.perf_counter() - t0)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

if os.environ.get("TRACE_DOTS"):
    sync = os.environ["TRACE_DOTS"] == "1"
    orig = runner._dots
    def traced(x, y, n_sub, batch):
        if sync: torch.cuda.synchronize()
        t = time.perf_counter()
        out = orig(x, y, n_sub, batch)
        print(f"_dots nx={len(x)} ny={len(y)} n_sub={n_sub} batch={batch} blkx={sum(r is not None and r.blk is not None for r in x)} totx={sum(r is not None and r.tot is not None for r in x)}: {(time.perf_counter()-t)*1e3:.2f} ms")
        return out
    runner._dots = traced
    import pyrhe_amd.engine as E
    lib = runner.eng.lib
    raw = lib.pyrhe_engine_loo_dots
    class W:
        def __call__(self, *a):
            t = time.perf_counter(); r = raw(*a); dt = (time.perf_counter() - t) * 1e3
            if dt > 2: print(f"   C call pyrhe_engine_loo_dots took {dt:.2f} ms")
            return r
    lib.pyrhe_engine_loo_dots = W()
    _cpu = torch.Tensor.cpu
    def cpu2(self, *a, **k):
        t = time.perf_counter(); r = _cpu(self, *a, **k); dt = (time.perf_counter() - t) * 1e3
        if dt > 8: print(f"   .cpu() took {dt:.2f} ms shape {tuple(self.shape)}")
        return r
    torch.Tensor.cpu = cpu2
    for rep in range(2):
        t0 = time.perf_counter(); runner.pass2(); torch.cuda.synchronize(); print("pass2 traced s", time.perf_counter() - t0)
