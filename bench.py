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
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="launch-shape knob of pyrhe_engine_set_option (pyrhe_amd.engine.OPTIONS), e.g. g1_workgroups=708; "
                         "every value computes the same results")
    ap.add_argument("--pipeline", action="store_true",
                    help="pyrhe_engine_set_pipeline: GEMM 1 of call j+1 beside GEMM 2 of call j on two private streams (off by "
                         "default: overlapped kernels blur the per-kernel durations the roofline object is built from)")
    ap.add_argument("--materialize-xxuz", action="store_true",
                    help="covariates as the k sketch columns U Z (the reference's XXUz arrays) instead of c cross columns whose "
                         "Gram matrix W^T X X^T W replaces them (SketchRunner.cross)")
    ap.add_argument("--norm-digits", type=int, default=3, choices=[2, 3],
                    help="precision experiment: int8 digits of the phenotype column in GEMM 1 (PYRHE_COL_DIGITS; 3 = all 24 bits)")
    ap.add_argument("--compat", default=None, choices=["reference", "fixed"],
                    help="UXXz semantics with covariates (SketchRunner compat); default: fixed with covariates / GENIE")
    ap.add_argument("--force-collectives", action="store_true",
                    help="rehearsal aid: initialise the process group and run every collective of the multi-GPU path (in-place "
                         "float64 all-reduce per bucket, packed norm all-reduce, tensor all-gather of the systems, barriers) even "
                         "with ONE rank -- RCCL then executes on a one-GPU box")
    ap.add_argument("--no-cross-check", action="store_true", help="skip the in-run i8x3 vs f16x2 comparison on three blocks")
    return ap.parse_args()


def cpu_baseline(n_indv: int, k: int, n_blocks: int, block_snps: int, gpu_check=None) -> dict:
    """The oracle's reference-order pass-1 body (decode -> flip -> per-SNP impute loop -> standardize -> k float32
    GEMV pairs + yXXy, float64 accumulate; oracle/rhe_oracle.py restating base_streaming.py:85-104) on a bounded
    sample of the same synthetic cohort -- its first ``n_blocks`` x ``block_snps`` SNPs as ``n_blocks`` jackknife
    blocks -- on this box's host cores.  SURVEY 8(d) planned 100 x 200 SNPs; at the ~85 SNPs/s this port reaches at
    N = 500k that is four minutes, so the default sample is 10 x 200 (the metric is linear in the SNP count).
    ``gpu_check(j, oracle_block)``: the caller's comparison of the HIP engine's sketch of the same block with the
    oracle's (the oracle is the checker here, never the thing measured on the GPU side); its worst relative
    deviations are reported beside the baseline."""
    from oracle import plink, synth
    from oracle import rhe_oracle as O
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    n_snps = n_blocks * block_snps
    g = synth.synth_rows(SYNTH_SEED, 0, n_snps, n_indv)
    packed = plink.encode_bed_rows(g)
    rng = np.random.default_rng(1)
    y = rng.normal(size=(n_indv, 1)); y -= y.mean()
    p = O.Problem(packed=packed, n_indv_orig=n_indv, annot=np.ones((n_snps, 1), np.int64), pheno=y, n_jack=n_blocks, k=k, seed=0)
    p.Z = O.make_probe_vectors(0, n_indv, k)
    dt, worst = 0.0, {}
    for j in range(n_blocks):
        t0 = time.perf_counter()
        blk = O.sketch_block(p, j, reseed=False)
        dt += time.perf_counter() - t0
        if gpu_check is not None:
            blk_x = None
            if j < 2:          # untimed: the same block with the float32 products summed exactly (the reference's own
                try:           # accumulation noise is |reference order - exact|; at N = 500k it is ~6e-6 of the block maximum)
                    O.ACCUM = "exact"
                    blk_x = O.sketch_block(p, j, reseed=False)
                finally:
                    O.ACCUM = "reference"
            for key, v in gpu_check(j, blk, blk_x).items():
                worst[key] = max(worst.get(key, 0.0), float(v))
    out = {"value": n_snps / dt, "unit": "SNPs/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"pass-1 body of the reference-order oracle on the first {n_blocks} x {block_snps} SNPs x N={n_indv}, k={k} "
                     f"(SURVEY 8d planned 100 x 200; bounded to keep the default run short), torch CPU float32 GEMV pairs ({dt:.1f} s)"}
    if worst:
        out["gpu_vs_oracle"] = worst
    return out


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU over RCCL ("nccl").  PYRHE_DIST_BACKEND=gloo is a rehearsal aid only: several ranks may then share
    # one GPU (local_rank wraps around), which RCCL refuses.
    backend = os.environ.get("PYRHE_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    use_dist = world > 1 or args.force_collectives
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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
    cap = args.sub_snps or engine_capacity(plan, N)
    opts = [(o.split("=", 1)[0], int(o.split("=", 1)[1])) for o in args.opt]

    def make_engine(precision, capacity=None):
        e_ = eng_mod.SketchEngine(N, Nk, max_entries=capacity or engine_capacity(plan, N), precision=precision, device=dev,
                                  sub_snps=capacity or cap)
        e_.set_coding(coding)
        for name, val in opts:
            e_.set_option(name, val)
        if args.pipeline:
            e_.set_pipeline(True)
        return e_

    engine = make_engine(args.precision)
    np.random.seed(0)
    Z = np.random.randn(Nk, k)                                              # NumPy legacy stream, seed 0
    y = np.random.default_rng(1).normal(size=Nk); y -= y.mean()
    if args.verify and args.h2 > 0:
        # heritable phenotype for the parity check between precision modes: y = a * (K r) + b * noise, K r = X X^T r / M
        r = np.random.default_rng(7).normal(size=(Nk, 1))
        # always built in the parity-safe mode, so that every --precision is verified on the SAME phenotype
        eng_y = make_engine("i8x3")
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
    n_eng = n_engines_needed(k, cov is not None, bool(args.genie))
    engines = [engine] + [make_engine(args.precision) for _ in range(n_eng - 1)]
    compat = args.compat or ("fixed" if (cov is not None or args.genie) else "reference")
    runner = SketchRunner(engines, geno, plan, (j0, j1), n_orig=N, missing_indv=dropped, Z=Z, cov=cov, env=env,
                          genie_model=args.genie, seed=0, streaming=True, dom=args.dom, compat=compat,
                          materialize_xxuz=args.materialize_xxuz, force_collectives=args.force_collectives)
    runner.norm_digits = args.norm_digits
    runner.begin_trait(y, check_missing=check_missing)      # the generator itself emits no missing calls (contract of synth_kernel)

    def barrier():
        if use_dist:
            if backend == "nccl":
                dist.barrier(device_ids=[dev.index])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    def timed_passes(rn):
        barrier()
        t0_ = time.perf_counter()
        for _ in range(args.steps):
            rn.pass1()
        barrier()
        dt_ = time.perf_counter() - t0_
        if use_dist:
            t_ = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            dt_ = float(t_.item())
        return dt_

    for _ in range(args.warmup):
        runner.pass1()
    barrier()
    engine.profile(True)
    dt = timed_passes(runner)
    prof = engine.profile_read()
    engine.profile(False)
    path = engine.last_path()                    # the kernels the timed passes launched (pyrhe_engine_info 6..8)
    ms_per_step = dt / args.steps * 1e3
    value = M * args.steps / dt

    row_bytes = (N + 3) // 4
    local_snps = (r1 - r0) * args.steps
    launches = max(prof["launches"], 1)
    g1, g2 = prof["gemm1_ms"], prof["gemm2_ms"]
    dom = "gemm2_kernel" if g2 >= g1 else "gemm1_kernel"
    dom_ms = max(g1, g2)
    g1_i8 = path["gemm1"] >= 100
    g1_tiles_i8 = path["gemm1"] % 100
    g1_name = (f"gemm1_i8w_kernel<{g1_tiles_i8}>" if path["gemm1"] >= 200 else f"gemm1_i8_kernel<{g1_tiles_i8}>") if g1_i8 \
        else f"gemm1_kernel<{path['gemm1']}>"
    g2_packed = path["gemm2"] in (11, 12, 13)
    g2_name = ("gemm2_i8p_kernel<2, GV=2>" if path["gemm2"] == 13 else f"gemm2_i8p_kernel<{path['gemm2'] - 10}>") if g2_packed else "gemm2_kernel"
    dom_name = g2_name if dom == "gemm2_kernel" else g1_name
    bytes_per_launch = local_snps * row_bytes / launches                    # algorithmic bytes: ceil(N/4) per SNP
    achieved = bytes_per_launch / (dom_ms / launches * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # HBM bytes per launch: NOT a counter of this run.  It is the bytes-per-algorithmic-byte ratio of that kernel from the
    # committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md; separate rocprofv3 --pmc runs) times this run's
    # algorithmic bytes per launch, and only when this run has the shape those passes were taken on; `traffic_source` says
    # which file, which commit and which shape.
    traffic, traffic_source = None, None
    shape = None
    if not args.dom and not args.materialize_xxuz:
        if args.n_bin == 1 and not args.n_cov and not args.genie:
            shape = "1bin"
        elif args.n_bin == 8 and args.n_cov == 10 and not args.genie:
            shape = "8bin+10cov"
        elif args.n_bin == 1 and not args.n_cov and args.genie == "G+GxE":
            shape = "G+GxE"
    try:
        import glob
        tfile = sorted(f for f in glob.glob(os.path.join(REPO, "profiles", "r*_traffic.json")) if "shapes" in json.load(open(f)))[-1]   # the latest round's
        with open(tfile) as fh:
            tj = json.load(fh)
        kern = tj["shapes"][shape]["kernels"][dom_name] if shape else None
        if kern and N == tj["n_indv"] and M // jn == tj["block_snps"] and k == 10:
            traffic = int(kern["bytes_per_algorithmic_byte"] * bytes_per_launch)
            traffic_source = (f"profiles/{os.path.basename(tfile)}@{tj.get('commit', '?')} [{shape}]: separate --pmc passes at N={tj['n_indv']}, "
                              f"M={tj.get('n_snp', '?')}, jn={tj.get('jn', '?')}, ratio x this run's algorithmic bytes")
    except Exception:
        pass
    # matrix-core work of the dominant kernel as issued (16-column tiles, all limbs / parts), beside the byte roofline:
    # (the fp16 GEMM 2 is bound by instruction issue; the packed int8 GEMM 2 and GEMM 1 by HBM)
    n_pad = engine.n_pad
    tiles_in = (engine.n_sketch + engine.n_norm + engine.n_cross + 1 + 15) // 16              # + the ones column
    tiles_out = max(1, (engine.n_sketch + 15) // 16)
    rows_per_launch = local_snps / launches
    if dom == "gemm2_kernel" and g2_packed:
        mfma_ops = rows_per_launch * 2.0 * n_pad * 16 * {11: 2, 12: 4, 13: 6}[path["gemm2"]]      # packed tiles of (digit, column) rows
        mfma_peak = 5000.0                                                                     # int8 dense TOP/s
    elif dom == "gemm2_kernel":
        parts = 1 if args.precision in ("f16", "f16e", "i8f16") else 2
        mfma_ops, mfma_peak = rows_per_launch * 2.0 * n_pad * 16 * tiles_out * parts, 2500.0       # fp16 dense TFLOP/s
    else:
        g1_tiles = g1_tiles_i8 if g1_i8 else tiles_in * (2 if args.precision == "f16x2" else 1)
        mfma_ops = rows_per_launch * 2.0 * n_pad * 16 * g1_tiles
        mfma_peak = 5000.0 if g1_i8 else 2500.0                                                    # int8 dense TOP/s
    mfma_tops = mfma_ops / (dom_ms / launches * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "mfma_tops_issued": round(mfma_tops, 1), "mfma_frac": round(mfma_tops / mfma_peak, 4),
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "bytes_per_launch": int(bytes_per_launch), "avg_launch_ms": round(dom_ms / launches, 4),
                "kernels": {"gemm1": g1_name, "gemm2": g2_name, "segmented_flow": path["segmented"]},
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
                      "options": dict(opts), "gen_seconds": round(t_gen, 2),
                      "covariates_as": None if cov is None else ("cross columns (W^T X X^T W per block and bin)" if runner.cross else "U Z sketch columns")},
           "roofline": roofline}

    if use_dist:
        # what the collectives saw: backend, ranks, blocks per rank (an all-gather of this rank's count)
        mine = torch.tensor([j1 - j0], dtype=torch.int64, device=dev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        out["config"]["dist"] = {"backend": dist.get_backend(), "world": dist.get_world_size(),
                                 "blocks_per_rank": [int(p_.item()) for p_ in parts], "collectives_forced": bool(args.force_collectives and world == 1)}

    def extra(name, fn):
        """A leg beside the headline: its failure is reported in the line, it never sinks the measured number."""
        try:
            fn()
        except Exception as ex:
            out["config"][name] = {"failed": f"{type(ex).__name__}: {ex}"}

    def leg_stream_read():
        """the box's MEASURED stream bandwidth beside the nominal peak -- a plain 16-bytes-per-lane read of this rank's
        genotype rows (pyrhe_stream_read, up to 32 GB of them), and the dominant kernel's rate as a fraction of it."""
        n_rows_s = int(min(geno.shape[0], (32 << 30) // geno.shape[1]))
        gbs = eng_mod.stream_read_gbs(geno[:n_rows_s])
        roofline["stream_read"] = {"measured_GBps": round(gbs, 1), "bytes": int(n_rows_s * geno.shape[1]),
                                   "frac_of_measured": round(achieved / gbs, 4) if gbs > 0 else None,
                                   "what": "plain linear read, 16 B per lane, four loads in flight (pyrhe_stream_read); best of 3"}
    if not args.no_alt:
        extra("stream_read", leg_stream_read)

    def leg_end_to_end():     # pass 2 + solves are cheap next to the sketch pass: always timed, reported beside the metric
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
    extra("end_to_end", leg_end_to_end)

    def leg_plink_coded():
        """The same pass on PLINK-coded rows (the .bed body as it is; the C ABI's default coding): what the one-time value-code
        transcode buys per pass, and what it costs (timed here on this rank's rows; the product hides it behind the upload,
        engine.upload_packed(recode="value"))."""
        if coding != "value" or args.dom:
            return
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng_mod.recode_rows(geno, "plink")
        torch.cuda.synchronize()
        t_rec = time.perf_counter() - t0
        try:
            for e_ in engines:
                e_.set_coding("plink")
            runner.pass1()
            dt3 = timed_passes(runner)
            out["config"]["plink_coded"] = {"value": M * args.steps / dt3, "ms_per_step": dt3 / args.steps * 1e3,
                                            "pass_frac": round(M * args.steps / dt3 * row_bytes / 1e9 / HBM_PEAK_GBS, 4)}
            out["config"]["recode_ms"] = round(t_rec * 1e3, 2)
        finally:
            eng_mod.recode_rows(geno, "value")
            for e_ in engines:
                e_.set_coding("value")
            torch.cuda.synchronize()
    if not args.no_alt:
        extra("plink_coded", leg_plink_coded)

    def leg_cross_check():
        """The kernels behind `value` against the fp16 hi+lo kernels (other code in every stage) on the first three local
        blocks: sigma^2 and jackknife SE of a 3-block run, same inputs."""
        nb3 = min(3, j1 - j0)
        if nb3 < 2 or args.dom:
            return
        s3, e3 = plan.range(j0)[0], plan.range(j0 + nb3 - 1)[1]
        sub = BlockPlan(annot[s3:e3], nb3)
        res = {}
        for prec in (args.precision, "f16x2"):
            engs = [make_engine(prec, engine_capacity(sub, N)) for _ in range(n_eng)]
            rn = SketchRunner(engs, geno[s3 - r0:e3 - r0], sub, (0, nb3), n_orig=N, missing_indv=dropped, Z=Z, cov=cov, env=env,
                              genie_model=args.genie, seed=0, streaming=True, compat=compat, solo=True,
                              materialize_xxuz=args.materialize_xxuz)
            sysm = rn.run_trait(y, check_missing=check_missing)
            sig = est.solve_all(sysm.T, sysm.q)
            res[prec] = (sig[-1], np.asarray(est.jackknife_se(sig[:-1], nb3)), engs[0].last_path())
            del rn
            for e_ in engs:
                e_.close()
        a, b = res[args.precision], res["f16x2"]
        out["config"]["cross_check"] = {"blocks": nb3, "against": "f16x2 kernels", "kernels": a[2],
                                        "sigma_rel": float(np.max(np.abs(a[0] - b[0]) / np.abs(b[0]))),
                                        "se_rel": float(np.max(np.abs(a[1] - b[1]) / np.abs(b[1])))}
    if not args.no_cross_check and args.precision != "f16x2":
        extra("cross_check", leg_cross_check)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the CPU baseline leg; while the oracle has each block in hand it also checks the HIP engine's sketch of the same
        # block (the headline kernels at the headline sample count) -- the oracle is the checker, never the product
        n_s = args.cpu_baseline_snps or int(min(20000, max(200, 2000 * 500_000 / max(N, 1))))
        bs = 200 if n_s >= 400 else n_s
        nblk = max(1, n_s // bs)
        chk_eng = make_engine(args.precision, max(bs, 1))
        keep = chk_eng.keep_vector(dropped)
        kt = torch.as_tensor(keep_idx, device=dev)

        def padded(a2):                                      # [n, Nk] float64 -> [n, n_pad] float32 at the kept positions
            t_ = torch.zeros((a2.shape[0], chk_eng.n_pad), dtype=torch.float32, device=dev)
            t_[:, kt] = torch.from_numpy(np.ascontiguousarray(a2)).to(dev).to(torch.float32)
            return t_
        y_b = np.random.default_rng(1).normal(size=(N, 1)); y_b -= y_b.mean()       # the baseline's own phenotype (cpu_baseline)
        can_check = not dropped and not args.missing_rate and r0 == 0 and (r1 - r0) >= nblk * bs
        if can_check:
            chk_eng.set_rhs(padded(make_probe(N, k).T), padded(y_b.T), keep)
        ent = torch.arange(bs, dtype=torch.int32, device=dev)

        def gpu_check(j, blk, blk_x):
            """Deviations of the engine's block sketch, relative to the block's largest entry: from the oracle in the
            reference's summation order, and (first two blocks) from the exactly summed one, beside the distance between the
            two oracles = the reference's own float32 accumulation noise."""
            sk, nm, _ = chk_eng.sketch_block(geno[j * bs:(j + 1) * bs], ent, [0, bs])
            got = sk[0, :, :N].double().cpu().numpy()
            ref = blk.XXz[0]
            sc = np.abs(ref).max()
            out_ = {"XXz_engine_vs_reference_order": np.abs(got - ref).max() / sc,
                    "yXXy_engine_vs_reference_order": abs(float(nm[0, 0]) - blk.yXXy[0]) / abs(blk.yXXy[0])}
            if blk_x is not None:
                out_["XXz_engine_vs_exact_sum"] = np.abs(got - blk_x.XXz[0]).max() / sc
                out_["XXz_reference_order_vs_exact_sum"] = np.abs(ref - blk_x.XXz[0]).max() / sc
                out_["yXXy_engine_vs_exact_sum"] = abs(float(nm[0, 0]) - blk_x.yXXy[0]) / abs(blk_x.yXXy[0])
            return out_
        try:
            out["cpu_baseline"] = cpu_baseline(N, k, nblk, bs, gpu_check if can_check else None)
            if can_check:
                out["cpu_baseline"]["gpu_kernels_checked"] = chk_eng.last_path()
        except Exception as ex:                                             # the baseline must never sink the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "SNPs/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
        chk_eng.close()

    def leg_alt():
        # the same pass in the single-rounding mode (1 MFMA per fragment, subnormal unpack): reported beside the
        # headline, never as `value` -- its sigma^2 error is ~2^-12/sqrt(N k), not bounded by 1e-5 at every size
        nonlocal runner, engines
        del runner
        for e_ in engines:
            e_.close()
        engines = [make_engine("f16") for _ in range(n_eng)]
        rn = SketchRunner(engines, geno, plan, (j0, j1), n_orig=N, missing_indv=dropped, Z=Z, cov=cov, env=env,
                          genie_model=args.genie, seed=0, streaming=True, compat=compat, materialize_xxuz=args.materialize_xxuz)
        rn.begin_trait(y, check_missing=check_missing)
        rn.pass1()
        dt2 = timed_passes(rn)
        out["config"]["f16_single_rounding"] = {"value": M * args.steps / dt2, "ms_per_step": dt2 / args.steps * 1e3,
                                                "pass_frac": round(M * args.steps / dt2 * row_bytes / 1e9 / HBM_PEAK_GBS, 4)}
        del rn
    if not args.no_alt and not args.dom and args.precision in ("f16x2", "i8x3"):
        extra("f16_single_rounding", leg_alt)
    for e_ in engines:
        e_.close()
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def make_probe(n, k):
    """The CPU baseline's probe vectors (oracle.make_probe_vectors: NumPy legacy stream, seed 0), for the GPU side of its check."""
    np.random.seed(0)
    return np.random.randn(n, k)


if __name__ == "__main__":
    main()
