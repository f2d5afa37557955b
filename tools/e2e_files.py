"""File -> result wall time of the model classes at a moderate scale (GPU box only): writes a synthetic cohort in the
reference's input formats under /tmp (genotypes from the device generator), then runs StreamingRHE with 8 bins and
covariates and prints the phase timings.  Exercises the host side: parsers, memory-mapped upload, recode, statistics,
both passes, estimators (incl. the overlapping-annotation h2), log."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrhe_amd import engine as E, models
from pyrhe_amd.logger import Logger

N, M, nbin, ncov = int(os.environ.get("N", 100_000)), int(os.environ.get("M", 200_000)), 8, 5
out = "/tmp/e2e"; os.makedirs(out, exist_ok=True); prefix = os.path.join(out, "cohort")
dev = torch.device("cuda:0")
t0 = time.perf_counter()
g = E.synth_genotypes(M, N, 7, 0, dev)
if os.environ.get("MISSING"):                 # plant PLINK missing codes (01): the imputation path then runs in every pass
    for a in range(0, M, 4096):
        part = g[a:a + 4096]
        hit = torch.rand(part.shape, device=dev) < float(os.environ["MISSING"]) * 4
        part[hit] = (part[hit] & 0xFC) | 0x01
    del part, hit
rb = (N + 3) // 4
with open(prefix + ".bed", "wb") as f:
    f.write(bytes([0x6C, 0x1B, 0x01]))
    for r0 in range(0, M, 20000):
        f.write(g[r0:r0 + 20000, :rb].contiguous().cpu().numpy().tobytes())
del g
rng = np.random.default_rng(0)
with open(prefix + ".fam", "w") as f:
    f.write("".join(f"{i} {i} 0 0 0 -9\n" for i in range(N)))
with open(prefix + ".bim", "w") as f:
    f.write("".join(f"1\trs{s}\t0\t{s}\tA\tG\n" for s in range(M)))
y = rng.normal(size=N)
with open(prefix + ".pheno", "w") as f:
    f.write("FID IID pheno0\n" + "".join(f"{i} {i} {float(y[i])!r}\n" for i in range(N)))
bins = rng.integers(0, nbin, M)
with open(prefix + ".annot", "w") as f:
    f.write("".join(" ".join("1" if b == bins[s] else "0" for b in range(nbin)) + "\n" for s in range(M)))
cov = rng.normal(size=(N, ncov))
with open(prefix + ".cov", "w") as f:
    f.write("FID IID " + " ".join(f"cov{j}" for j in range(ncov)) + "\n" + "".join(f"{i} {i} " + " ".join(repr(float(v)) for v in cov[i]) + "\n" for i in range(N)))
print(f"wrote cohort N={N} M={M} ({os.path.getsize(prefix + '.bed') / 1e9:.2f} GB .bed) in {time.perf_counter() - t0:.1f} s")
log = Logger(suppress=True, debug_mode=False)
t0 = time.perf_counter()
m = models.StreamingRHE(geno_file=prefix, annot_file=prefix + ".annot", pheno_file=prefix + ".pheno", cov_file=prefix + ".cov",
                        num_jack=100, num_random_vec=10, seed=0, device="cuda", log=log)
torch.cuda.synchronize(); t1 = time.perf_counter()
import cProfile, pstats
pr0 = cProfile.Profile(); pr0.enable()
res = m(trait=0)
torch.cuda.synchronize(); pr0.disable(); t2 = time.perf_counter()
if os.environ.get("PROFILE_FIRST"):
    pstats.Stats(pr0).sort_stats("cumulative").print_stats(28)
print(f"construct (parse + upload + recode + engines): {t1 - t0:.2f} s;  trait (pass 1 + pass 2 + estimates): {t2 - t1:.2f} s")
print("sigma", np.round(res["sigma_ests_total"], 5), "h2", np.round(res["h2_total"], 5))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); res = m(trait=0); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
