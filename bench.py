#!/usr/bin/env python3
"""bench.py -- SNPs/sec through the streaming trace sketch on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is ONE streaming sketch pass over the whole synthetic cohort: every jackknife block goes through
decode -> per-SNP statistics -> X^T R -> X (X^T R) for k probe vectors + the phenotype norm, the per-block
sketches are accumulated and the per-bin accumulators are all-reduced over the GPUs (RCCL over xGMI).
Workload = BASELINE.json configs[2] ("Synthetic N=500k M=1M single-bin StreamingRHE, k=10 jn=100"), the
configuration the metric is quoted on; genotypes are generated on the device (resident in HBM before the timed
region) by the counter-based generator that oracle/synth.py reproduces on the CPU.  With N GPUs the SAME cohort
is sharded by jackknife block (strong scaling).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
SYNTH_SEED = 20251212


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-indv", type=int, default=500_000)
    ap.add_argument("--n-snp", type=int, default=1_000_000)
    ap.add_argument("--jn", type=int, default=100)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--n-bin", type=int, default=1, help="annotation bins (bin(s) = hash(s) mod n_bin; BASELINE configs[3] uses 8)")
    ap.add_argument("--n-cov", type=int, default=0, help="covariate columns incl. intercept (configs[3]: 11)")
    ap.add_argument("--genie", default=None, choices=["G+GxE", "G+GxE+NxE"],
                    help="StreamingGENIE sketch pass with one Bernoulli(0.5) environment (BASELINE configs[4])")
    ap.add_argument("--precision", default=os.environ.get("PYRHE_PRECISION", "i8x3"), choices=["f16", "f16x2", "f16e", "i8x3", "i8f16"])
    ap.add_argument("--sub-snps", type=int, default=int(os.environ.get("PYRHE_SUB_SNPS", "0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-snps", type=int, default=0, help="SNPs in the CPU baseline sample (0 = auto)")
    ap.add_argument("--verify", action="store_true", help="also run pass 2 + solve and print sigma^2")
    ap.add_argument("--h2", type=float, default=0.5, help="heritable share of the --verify phenotype")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra timed passes in the single-rounding f16 mode")
    ap.add_argument("--missing-rate", type=float, default=0.0,
                    help="cross-check aid (tools/verify_sweep.sh): plant missing calls at this rate -- the imputation path "
                         "(block statistics, replayed NumPy draws, fill values in both contractions) then runs in every pass")
    ap.add_argument("--drop-indv", type=int, default=0, help="cross-check aid: individuals without a phenotype (keep mask)")
    ap.add_argument("--dom", action="store_true", help="RHE_DOM: additive + dominance sketches (pyrhe_engine_sketch_block_dom)")
    return ap.parse_args()


def cpu_baseline(n_indv: int, k: int, n_snps: int) -> dict:
    """The oracle's reference-order pass-1 body (decode -> flip -> per-SNP impute loop -> standardize -> k float32
    GEMV pairs + yXXy, float64 accumulate; oracle/rhe_oracle.py restating base_streaming.py:85-104) on a bounded
    sample of the same synthetic cohort, on this box's host cores."""
    from oracle import plink, synth
    from oracle import rhe_oracle as O
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    g = synth.synth_rows(SYNTH_SEED, 0, n_snps, n_indv)
    packed = plink.encode_bed_rows(g)
    rng = np.random.default_rng(1)
    y = rng.normal(size=(n_indv, 1)); y -= y.mean()
    p = O.Problem(packed=packed, n_indv_orig=n_indv, annot=np.ones((n_snps, 1), np.int64), pheno=y, n_jack=1, k=k, seed=0)
    p.Z = O.make_probe_vectors(0, n_indv, k)
    t0 = time.perf_counter()
    O.sketch_block(p, 0, reseed=False)
    dt = time.perf_counter() - t0
    return {"value": n_snps / dt, "unit": "SNPs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"pass-1 body of the reference-order oracle on {n_snps} SNPs x N={n_indv}, k={k}, one block, "
                      f"torch CPU float32 GEMV pairs ({dt:.1f} s)"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU over RCCL ("nccl").  PYRHE_DIST_BACKEND=gloo is a rehearsal aid only: several ranks may then share
    # one GPU (local_rank wraps around), which RCCL refuses.
    backend = os.environ.get("PYRHE_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(dev)

    from pyrhe_amd import engine as eng_mod
    from pyrhe_amd.streaming import n_engines_needed, BlockPlan, SketchRunner, engine_capacity, shard_ranges
    from pyrhe_amd import estimate as est

    N, M, jn, k = args.n_indv, args.n_snp, args.jn, args.k
    annot = np.ones((M, 1), np.int8)
    if args.n_bin > 1:
        h = (np.arange(M, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)
        annot = np.zeros((M, args.n_bin), np.int8)
        annot[np.arange(M), (h % np.uint64(args.n_bin)).astype(np.int64)] = 1
    plan = BlockPlan(annot, jn)
    j0, j1 = shard_ranges(jn, world)[rank]
    r0 = plan.range(j0)[0] if j1 > j0 else 0
    r1 = plan.range(j1 - 1)[1] if j1 > j0 else 0
    t_gen = time.perf_counter()
    geno = eng_mod.synth_genotypes(r1 - r0, N, SYNTH_SEED, r0, dev)          # resident in HBM before timing
    if args.missing_rate > 0:                    # PLINK missing code 01 in the first call of random bytes, before the recode
        gen = torch.Generator(device=dev)
        for a in range(r0 - r0 % 4096, r1, 4096):            # the pattern is a function of the GLOBAL row: same cohort on any rank count
            gen.manual_seed(99 + a // 4096)
            hit = torch.rand((4096, geno.shape[1]), device=dev, generator=gen) < args.missing_rate * 4
            lo, hi = max(a, r0), min(a + 4096, r1)
            part = geno[lo - r0:hi - r0]
            hit = hit[lo - a:hi - a]
            part[hit] = (part[hit] & 0xFC) | 0x01
        del part, hit
    coding = os.environ.get("PYRHE_CODING", "value")
    if coding == "value":
        eng_mod.recode_rows(geno, "value")       # the load-time recode of the product path (models.py), outside the timing
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    dropped = sorted(np.random.default_rng(5).choice(N, args.drop_indv, replace=False).tolist()) if args.drop_indv else []
    keep_idx = np.setdiff1d(np.arange(N), np.asarray(dropped, dtype=np.int64))
    Nk = len(keep_idx)                           # individuals with a phenotype: every per-individual input has this length
    check_missing = args.missing_rate > 0
    engine = eng_mod.SketchEngine(N, Nk, max_entries=engine_capacity(plan), precision=args.precision, device=dev,
                                  sub_snps=args.sub_snps or engine_capacity(plan))
    engine.set_coding(coding)
    np.random.seed(0)
    Z = np.random.randn(Nk, k)                                              # NumPy legacy stream, seed 0
    y = np.random.default_rng(1).normal(size=Nk); y -= y.mean()
    if args.verify and args.h2 > 0:
        # heritable phenotype for the parity check between precision modes: y = a * (K r) + b * noise, K r = X X^T r / M
        r = np.random.default_rng(7).normal(size=(Nk, 1))
        # always built in the parity-safe mode, so that every --precision is verified on the SAME phenotype
        eng_y = eng_mod.SketchEngine(N, Nk, max_entries=engine_capacity(plan), precision="i8x3", device=dev, sub_snps=args.sub_snps or engine_capacity(plan))
        eng_y.set_coding(coding)
        rr = SketchRunner(eng_y, geno, plan, (j0, j1), n_orig=N, missing_indv=dropped, Z=r, seed=0, streaming=True)
        rr.begin_trait(y, check_missing=check_missing)
        rr.pass1()
        g = rr.tot_sk[0, 0, :N].cpu().numpy()[keep_idx]
        del rr
        eng_y.close()
        g = (g - g.mean()) / g.std()
        y = np.sqrt(args.h2) * g + np.sqrt(1 - args.h2) * y / y.std()
        y -= y.mean()
    cov = None
    if args.n_cov > 0:
        cov = np.concatenate([np.ones((Nk, 1)), np.random.default_rng(2).normal(size=(Nk, args.n_cov - 1))], axis=1)
    env = None
    if args.genie:
        env = np.random.default_rng(3).integers(0, 2, size=Nk).astype(np.float64)     # Bernoulli(0.5)
    # one engine per pass over the genotypes: the column groups (Z, UZ, e.Z, e.UZ of k columns each) that do not fit one
    # engine's 30 right-hand columns go to further engines (k = 10: one, or two for GxE + covariates)
    engines = [engine]
    for _ in range(n_engines_needed(k, cov is not None, bool(args.genie)) - 1):
        engines.append(eng_mod.SketchEngine(N, Nk, max_entries=engine_capacity(plan), precision=args.precision, device=dev,
                                            sub_snps=args.sub_snps or engine_capacity(plan)))
        engines[-1].set_coding(coding)
    runner = SketchRunner(engines, geno, plan, (j0, j1), n_orig=N, missing_indv=dropped, Z=Z, cov=cov, env=env,
                          genie_model=args.genie, seed=0, streaming=True, dom=args.dom,
                          compat="fixed" if (cov is not None or args.genie) else "reference")
    runner.begin_trait(y, check_missing=check_missing)      # the generator itself emits no missing calls (contract of synth_kernel)

    def barrier():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[dev.index])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        runner.pass1()
    barrier()
    engine.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.pass1()
    barrier()
    dt = time.perf_counter() - t0
    prof = engine.profile_read()
    engine.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = M * args.steps / dt

    row_bytes = (N + 3) // 4
    local_snps = (r1 - r0) * args.steps
    launches = max(prof["launches"], 1)
    g1, g2 = prof["gemm1_ms"], prof["gemm2_ms"]
    dom = "gemm2_kernel" if g2 >= g1 else "gemm1_kernel"
    dom_ms = max(g1, g2)
    # which kernels ran (pyrhe_amd/csrc/sketch_engine.hip): int8 GEMM 1 under i8x3 / i8f16; the limb-packed int8 GEMM 2
    # under i8x3 with <= 20 sketch columns, the fp16 GEMM 2 otherwise
    g1_name = "gemm1_i8_kernel" if args.precision in ("i8x3", "i8f16") else "gemm1_kernel"
    g2_packed = args.precision == "i8x3" and 1 <= engine.n_sketch <= 20 and os.environ.get("PYRHE_G2_PACKED", "1") != "0"
    g2_name = "gemm2_i8p_kernel" if g2_packed else "gemm2_kernel"
    dom_name = g2_name if dom == "gemm2_kernel" else g1_name
    bytes_per_launch = local_snps * row_bytes / launches                    # algorithmic bytes: ceil(N/4) per SNP
    achieved = bytes_per_launch / (dom_ms / launches * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md): the measured
    # bytes per algorithmic byte of that kernel times this run's algorithmic bytes per launch, only when this run has the
    # shape those passes were taken on (10 000-SNP blocks at N = 500 000, one bin)
    traffic = None
    try:
        with open(os.path.join(REPO, "profiles", "r01_traffic.json")) as fh:
            tj = json.load(fh)
        key = {"gemm1_kernel": "gemm1_i8_kernel" if g1_name == "gemm1_i8_kernel" else "gemm1_kernel_f16",
               "gemm2_kernel": "gemm2_i8p_kernel" if g2_packed else
                               ("gemm2_kernel_f16" if args.precision in ("f16", "i8f16") else "gemm2_kernel_f16x2")}[dom]
        if N == tj["n_indv"] and M // jn == tj["block_snps"] and args.n_bin == 1:
            traffic = int(tj["kernels"][key]["bytes_per_algorithmic_byte"] * bytes_per_launch)
    except Exception:
        traffic = None
    # matrix-core work of the dominant kernel as issued (16-column tiles, all limbs / parts), beside the byte roofline:
    # (the fp16 GEMM 2 is bound by instruction issue; the packed int8 GEMM 2 and GEMM 1 by HBM)
    n_pad = engine.n_pad
    tiles_in = (engine.n_sketch + engine.n_norm + 1 + 15) // 16              # + the ones column
    tiles_out = max(1, (engine.n_sketch + 15) // 16)
    rows_per_launch = local_snps / launches
    if dom == "gemm2_kernel" and g2_packed:
        mfma_ops = rows_per_launch * 2.0 * n_pad * 16 * (2 if engine.n_sketch <= 10 else 4)   # packed tiles of (digit, column) rows
        mfma_peak = 5000.0                                                                     # int8 dense TOP/s
    elif dom == "gemm2_kernel":
        parts = 1 if args.precision in ("f16", "f16e", "i8f16") else 2
        mfma_ops, mfma_peak = rows_per_launch * 2.0 * n_pad * 16 * tiles_out * parts, 2500.0       # fp16 dense TFLOP/s
    else:
        if g1_name == "gemm1_i8_kernel":       # packed (digit, column) rows: 3 per user column + the ones column
            g1_tiles = (3 * (engine.n_sketch + engine.n_norm) + 1 + 15) // 16
        else:
            g1_tiles = tiles_in * (2 if args.precision == "f16x2" else 1)
        mfma_ops = rows_per_launch * 2.0 * n_pad * 16 * g1_tiles
        mfma_peak = 5000.0 if g1_name == "gemm1_i8_kernel" else 2500.0                             # int8 dense TOP/s
    mfma_tops = mfma_ops / (dom_ms / launches * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "mfma_tops_issued": round(mfma_tops, 1), "mfma_frac": round(mfma_tops / mfma_peak, 4),
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "bytes_per_launch": int(bytes_per_launch), "avg_launch_ms": round(dom_ms / launches, 4),
                "gemm1_ms_per_step": round(g1 / args.steps, 3), "gemm2_ms_per_step": round(g2 / args.steps, 3),
                "mid_ms_per_step": round(prof["mid_ms"] / args.steps, 3),
                "pass_frac": round(value * row_bytes / 1e9 / HBM_PEAK_GBS, 4)}

    out = {"metric": "SNPs/sec through streaming trace sketch (N=500k, k=10, jn=100)", "value": value, "unit": "SNPs/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": {"f16x2": "f16x2", "i8x3": "i8x3" if g2_packed else "i8x3+f16x2", "f16": "f16", "f16e": "f16", "i8f16": "i8x3+f16"}[args.precision], "data": "synthetic",
           "config": {"workload": f"synthetic N={N} M={M} {args.n_bin}-bin "
                                  + (f"StreamingGENIE {args.genie} (1 environment)" if args.genie else "StreamingRHE")
                                  + f" sketch pass, k={k} jn={jn}" + (f", {args.n_cov} covariates" if args.n_cov else ""),
                      "precision": args.precision, "coding": coding, "sub_snps": engine.sub_snps, "parallelism": f"blocks/{world}gpu",
                      "gen_seconds": round(t_gen, 2)},
           "roofline": roofline}

    if True:     # pass 2 + solves are cheap next to the sketch pass: always timed, reported beside the metric
        w = runner.pass2()                                                  # untimed: scratch allocation, BLAS thread start
        est.solve_all(w.T[:1], w.q[:1])
        barrier()
        t0 = time.perf_counter()
        sysm = runner.pass2()
        barrier()
        t_pass2 = time.perf_counter() - t0
        sig = est.solve_all(sysm.T, sysm.q)
        t_solve = time.perf_counter() - t0 - t_pass2
        # end to end = one sketch pass (incl. aggregate + all-reduce) + pass 2 (LOO systems) + the jn+1 solves
        out["config"]["end_to_end"] = {"pass1_s": round(ms_per_step / 1e3, 4), "pass2_s": round(t_pass2, 4), "solve_s": round(t_solve, 4),
                                       "snps_per_s": M / (ms_per_step / 1e3 + t_pass2 + t_solve)}
        if args.verify:
            out["config"]["sigma_total"] = [float(v) for v in sig[-1]]
            out["config"]["sigma_se"] = [float(v) for v in est.jackknife_se(sig[:-1], jn)]

    if not args.no_alt and not args.dom and args.precision in ("f16x2", "i8x3"):
        # the same pass in the single-rounding mode (1 MFMA per fragment, subnormal unpack): reported beside the
        # headline, never as `value` -- its sigma^2 error is ~2^-12/sqrt(N k), not bounded by 1e-5 at every size
        del runner
        engine.close()
        engine = eng_mod.SketchEngine(N, Nk, max_entries=engine_capacity(plan), precision="f16", device=dev, sub_snps=args.sub_snps or engine_capacity(plan))
        engine.set_coding(coding)
        runner = SketchRunner(engine, geno, plan, (j0, j1), n_orig=N, missing_indv=dropped, Z=Z, cov=cov, seed=0, streaming=True,
                              compat="fixed" if cov is not None else "reference")
        runner.begin_trait(y, check_missing=check_missing)
        runner.pass1()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            runner.pass1()
        barrier()
        dt2 = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt2], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        out["config"]["f16_single_rounding"] = {"value": M * args.steps / dt2, "ms_per_step": dt2 / args.steps * 1e3,
                                                "pass_frac": round(M * args.steps / dt2 * row_bytes / 1e9 / HBM_PEAK_GBS, 4)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        n_s = args.cpu_baseline_snps or int(min(20000, max(50, 400 * 500_000 / max(N, 1))))
        try:
            out["cpu_baseline"] = cpu_baseline(N, k, n_s)
        except Exception as ex:                                             # the baseline must never sink the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "SNPs/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
