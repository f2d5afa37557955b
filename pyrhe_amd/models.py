"""Model classes with the reference's user surface -- constructor kwargs, ``model(trait, method="QR") -> dict``,
result-dict keys and log lines -- running on the MI355X sketch engine.

Reference surface being mirrored: ``pyrhe/src/base/base.py:24-193,874-886`` (constructor, ``__call__``),
``pyrhe/src/models/rhe/rhe.py:28-101`` (``RHE.run`` log grammar + dict), ``pyrhe/src/models/rhe/streaming_rhe.py``.

Semantics per class:
  * ``RHE``          -- non-streaming semantics: per-block reseeded imputation stream, ``UXXz = U(block XXz)``.
  * ``StreamingRHE`` -- the reference's one-worker streaming semantics, including quirk Q4
                        (``UXXz`` accumulated from the running sum) unless ``compat="fixed"``.
Both keep the per-block sketches in HBM; ``recompute_pass2=True`` re-sketches blocks in pass 2 instead.
There is no CPU path: a missing GPU or HIP library raises.
"""
from __future__ import annotations

import os
import time
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import estimate as est
from . import io as rio
from .engine import SketchEngine, upload_packed
from .logger import Logger
from .streaming import engine_capacity, n_engines_needed, BlockPlan, SketchRunner, shard_ranges


class Base:
    streaming = False

    def __init__(self, model: str = "rhe", geno_file: str = None, annot_file: str = None, pheno_file: str = None,
                 cov_file: str = None, num_bin: int = 8, num_jack: int = 1, num_random_vec: int = 10,
                 geno_impute_method: str = "binary", cov_impute_method: str = "ignore",
                 cov_one_hot_conversion: Optional[bool] = False, categorical_threshhold: int = 100,
                 device: str = "cuda", cuda_num: Optional[int] = None, num_workers: Optional[int] = None,
                 multiprocessing: bool = True, seed: Optional[int] = None, get_trace: bool = False,
                 trace_dir: Optional[str] = None, samp_prev: Optional[float] = None, pop_prev: Optional[float] = None,
                 log: Optional[Logger] = None, precision: str = "i8x3", compat: str = "reference",
                 recompute_pass2: bool = False, sub_snps: int = 0, materialize_xxuz: bool = False,
                 geno_resident: Optional[bool] = None):
        self.model = model
        self.materialize_xxuz = bool(materialize_xxuz)
        self.num_jack = self.num_blocks = int(num_jack)
        self.num_random_vec = int(num_random_vec)
        self.geno_impute_methods = geno_impute_method
        self.log = log if log is not None else Logger(debug_mode=False)
        self.precision, self.compat, self.recompute_pass2 = precision, compat, recompute_pass2
        if self.num_jack < 1:
            raise ValueError("num_jack must be >= 1")
        # device: one process per GPU; LOCAL_RANK picks the card under torchrun
        if str(device) == "cpu":
            raise RuntimeError("pyrhe_amd runs the sketch on an MI355X only (no CPU path); use device='cuda'")
        if not torch.cuda.is_available():
            raise RuntimeError("pyrhe_amd: no ROCm GPU visible")
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        idx = cuda_num if (cuda_num is not None and int(cuda_num) > -1) else int(os.environ.get("LOCAL_RANK", 0))
        self.device = torch.device("cuda", int(idx))
        torch.cuda.set_device(self.device)
        self.log._debug(f"Using device {self.device}")

        # seed -> private legacy stream replaying the reference's global one (base.py:72-73).  Every rank must draw the
        # SAME probe vectors, generated annotation and imputation stream: an unset seed is chosen on rank 0 and broadcast.
        self.seed = int(time.process_time()) if seed is None else int(seed)
        if self.world > 1:
            given = [None] * self.world              # every rank sees every rank's seed, so that ALL of them raise on a mismatch
            dist.all_gather_object(given, None if seed is None else int(seed))      # (one raising alone leaves the others in a collective)
            box = [self.seed]
            dist.broadcast_object_list(box, src=0)
            bad = [(r, s_) for r, s_ in enumerate(given) if s_ is not None and s_ != int(box[0])]
            if bad:
                raise ValueError(f"ranks were given different seeds (rank 0 uses {box[0]}; differing: {bad}): all ranks need the same seed")
            self.seed = int(box[0])
        rs = np.random.RandomState(self.seed)

        self.num_indv_original, fam_df = rio.read_fam(geno_file + ".fam")
        self.num_snp = rio.count_bim(geno_file + ".bim")
        self._rng_prefix, self._rng_high = 0, 2
        if annot_file is None:
            if num_bin is None:
                raise ValueError("Must specify number of bins if annot file is not provided")
            annot_file = "generated_annot"
            draws = rs.randint(0, num_bin, self.num_snp)          # file_processing.py:102-111: one randint per SNP, same stream
            if self.rank == 0:                                    # one writer; the other ranks read it after the barrier
                with open(annot_file, "w") as f:
                    for d_ in draws:
                        row = [0] * num_bin
                        row[int(d_)] = 1
                        f.write(" ".join(str(v) for v in row) + "\n")
            if self.world > 1:
                dist.barrier()
            self._rng_prefix, self._rng_high = self.num_snp, int(num_bin)
        self.num_bin, self.annot_matrix, self.len_bin = rio.read_annot(annot_file)

        self.pheno_file = pheno_file
        self.pheno, missing_indv, self.binary_pheno = rio.read_pheno(pheno_file)
        self.num_traits = self.pheno.shape[1]
        self.log._log(f"Number of traits: {self.num_traits}")
        if cov_file is None:
            self.use_cov, self.cov_matrix, self.Q = False, None, None
            self.missing_indv = missing_indv
        else:
            self.use_cov = True
            self.cov_matrix, self.missing_indv = rio.read_cov(cov_file, missing_indvs=missing_indv, impute=cov_impute_method)
            self.log._log(f"Rank of the covariate matrix: {np.linalg.matrix_rank(self.cov_matrix)}")
            self.Q = np.linalg.pinv(self.cov_matrix.T @ self.cov_matrix)
        self.pheno = np.delete(self.pheno, self.missing_indv, axis=0)
        self.pheno = self.pheno - np.mean(self.pheno, axis=0)
        self.num_indv = self.num_indv_original - len(self.missing_indv)
        for i, m in enumerate(self.missing_indv, start=1):
            self.log._log(f"missing individual {i}: FID:{fam_df.iloc[m, 0]} IID:{fam_df.iloc[m, 1]}")
        self.log._log(f"Number of individuals after filtering: {self.num_indv}")
        if self.cov_matrix is not None:
            self.log._log(f"Number of covariates: {self.cov_matrix.shape[1]}")
        self.log._log("*****")
        for i, n in enumerate(self.len_bin):
            self.log._log(f"Number of features in bin {i} : {n}")

        self.all_zb = rs.randn(self.num_indv, self.num_random_vec)          # base.py:176
        if self.use_cov:
            self.all_Uzb = self.cov_matrix @ self.Q @ (self.cov_matrix.T @ self.all_zb)
        self.get_trace, self.trace_dir = get_trace, trace_dir
        self.samp_prev, self.pop_prev = samp_prev, pop_prev
        self.pheno_cp = self.pheno.copy()
        self.num_env, self.num_gen_env_bin = 0, 0
        self.env, self.genie_model = None, None

        # ---- genotype rows of this rank's blocks -> HBM (packed, never decoded on the host)
        self.plan = BlockPlan(self.annot_matrix, self.num_jack)
        self.local_blocks = shard_ranges(self.num_jack, self.world)[self.rank]
        j0, j1 = self.local_blocks
        packed = rio.read_bed_packed(geno_file + ".bed", self.num_indv_original, self.num_snp)
        r0 = self.plan.range(j0)[0] if j1 > j0 else 0
        r1 = self.plan.range(j1 - 1)[1] if j1 > j0 else 0
        # memmap slice -> HBM in pinned 256 MiB pieces; PLINK code -> allele counts in place, piece by piece on a second
        # stream while the next piece crosses PCIe: the transcode is hidden behind the load it belongs to
        # A shard that does not fit HBM (beside the sketch store and workspaces: more than 60 % of the free memory; N = 500k:
        # beyond ~1.3 M SNPs per GPU) stays on the host and reaches the device one engine call's rows at a time
        # (engine.HostRows): the pass is then bound by the PCIe link, but the cohort runs.  ``geno_resident`` /
        # PYRHE_GENO_RESIDENT=0|1 force either form.
        pitch = (packed.shape[1] + 255) // 256 * 256
        if geno_resident is None and os.environ.get("PYRHE_GENO_RESIDENT") in ("0", "1"):
            geno_resident = os.environ["PYRHE_GENO_RESIDENT"] == "1"
        if geno_resident is None:
            geno_resident = (r1 - r0) * pitch <= 0.6 * torch.cuda.mem_get_info(self.device)[0]
        self.geno_resident = bool(geno_resident)
        if self.geno_resident:
            self.geno_dev = upload_packed(packed[r0:r1], self.device, recode="value")
        else:
            from .engine import HostRows
            self.geno_dev = HostRows(packed[r0:r1], self.device, recode="value")
            self.log._log(f"Genotypes stay on the host ({(r1 - r0) * pitch / 1e9:.1f} GB for this rank): streamed to the GPU per engine call")
        self._sub_snps = sub_snps
        self.engine = self._new_engine()
        self._engines = [self.engine]
        self._grow_engines(has_gxe=False)
        self._runner = None

    def _new_engine(self) -> SketchEngine:
        cap = engine_capacity(self.plan, self.num_indv_original)
        eng = SketchEngine(self.num_indv_original, self.num_indv, max_entries=cap,
                           precision=self.precision, device=self.device, sub_snps=self._sub_snps or cap)
        eng.set_coding("value")
        return eng

    def _grow_engines(self, has_gxe: bool):
        """One engine per pass of at most 30 right-hand columns (num_random_vec, covariates and G x E decide how many)."""
        while len(self._engines) < n_engines_needed(self.num_random_vec, self.use_cov, has_gxe):
            self._engines.append(self._new_engine())

    # ---- plugin hooks of the reference (base.py:453-463)
    def get_num_estimates(self):
        return self.num_bin

    def get_M_last_row(self):
        return self.len_bin

    def _make_runner(self) -> SketchRunner:
        return SketchRunner(self._engines, self.geno_dev, self.plan, self.local_blocks, n_orig=self.num_indv_original,
                            missing_indv=self.missing_indv, Z=self.all_zb, cov=self.cov_matrix, env=self.env,
                            genie_model=self.genie_model, impute=self.geno_impute_methods, seed=self.seed,
                            streaming=self.streaming, compat=self.compat, recompute_pass2=self.recompute_pass2,
                            rng_prefix_draws=self._rng_prefix, rng_prefix_high=self._rng_high,
                            materialize_xxuz=self.materialize_xxuz)

    def pre_compute(self):
        t0 = time.time()
        if self._runner is None:
            self._runner = self._make_runner()
        self.num_estimates = self.get_num_estimates()
        trait = getattr(self, "_cur_trait", None)
        cache = self.__dict__.setdefault("_trait_systems", {})
        if trait is None:
            self.systems = self._runner.run_trait(self.pheno[:, 0])
        else:
            if trait not in cache:
                # The sketches do not depend on the phenotype: this phenotype and the following ones (as many as the spare
                # right-hand columns of one pass take) share both passes; their systems differ in q only.  The reference
                # repeats the whole computation per phenotype with identical intermediate results.
                group = list(range(trait, min(self.pheno_cp.shape[1], trait + self._runner.max_traits())))
                from .streaming import Systems
                sysm = self._runner.run_trait(self.pheno_cp[:, group])
                for i, t_ in enumerate(group):
                    cache[t_] = Systems(sysm.T, sysm.q_all[i], sysm.M, sysm.q_all[i:i + 1])
            self.systems = cache[trait]
        self.M = self.systems.M
        self.log._debug(f"Precompute total time: {time.time() - t0}")

    def estimate(self, method: str = "lstsq"):
        rows = est.solve_all(self.systems.T[:self.num_jack + 1], self.systems.q[:self.num_jack + 1], method)
        if self.get_trace and self.rank == 0:
            self.get_trace_summary()
        return rows[:-1, :], rows[-1, :]

    def get_trace_summary(self):
        """``--trace``: stochastic-trace summary statistics, files ``run_<pheno>.MN`` / ``.tr`` in the reference's
        format (base.py:598-599,827-855): LD_SUM[a,b] = (T[a,b] - N) * M_a * M_b / N^2 per jackknife row."""
        T, M, N = self.systems.T, self.M, self.num_indv
        ne, jn = self.num_estimates, self.num_jack
        name = f"run_{os.path.basename(self.pheno_file) if self.pheno_file is not None else None}"
        prefix = os.path.join(self.trace_dir, name) if (self.trace_dir and os.path.isdir(self.trace_dir)) else name
        with open(prefix + ".MN", "w") as fd:
            fd.write("NSAMPLE,NSNPS,NBLKS,NBINS,K\n")
            fd.write(f"{N:.0f},{self.num_snp:.0f},{jn:.0f},{self.num_bin:.0f},{self.num_random_vec:.0f}")
        with open(prefix + ".tr", "w") as fd:
            fd.write(",".join(f"LD_SUM_{i:d}" for i in range(self.num_bin)) + ",NSNPS_JACKKNIFE\n")
            for j in range(jn + 1):
                for a in range(self.num_bin):
                    vals = []
                    for b in range(self.num_bin):
                        mm = M[j, a] * M[j, b]
                        vals.append((T[j][a, b] - N) * mm / pow(N, 2) if mm != 0 else 0)
                    fd.write(",".join(f"{v:.3f}" for v in vals) + f",{M[j, a]:.0f}\n")
        self.log._log(f"Saved trace summary into {prefix}(.tr/.MN)")

    def estimate_error(self, ests):
        return est.jackknife_se(ests, self.num_jack)

    def __call__(self, trait, method: str = "QR"):
        self.pheno = self.pheno_cp[:, trait].reshape(-1, 1)
        self.log._log("*****")
        self.log._log(f"OUTPUT FOR TRAIT {trait}: ")
        self._cur_trait = int(trait)
        try:
            self.pre_compute()
        finally:
            self._cur_trait = None
        return self.run(method=method)

    def run(self, method):
        raise NotImplementedError


class RHE(Base):
    """Additive multi-component RHE (``pyrhe/src/models/rhe/rhe.py``)."""

    def run(self, method):
        log = self.log._log
        sigma_jack, sigma_total = self.estimate(method=method)
        sig_errs = self.estimate_error(sigma_jack)
        ne = self.num_estimates
        log("Variance components: ")
        for i, v in enumerate(sigma_total):
            if i == len(sigma_total) - 1:
                log(f"Sigma^2_e : {v}  SE : {sig_errs[i]}")
            else:
                log(f"Sigma^2_g[{i}] : {v}  SE : {sig_errs[i]}")
        h2_jack, h2_total = est.h2_nonoverlapping(sigma_jack, sigma_total, ne)
        h2_errs = self.estimate_error(h2_jack)
        log("*****")
        log("Heritabilities:")
        for i, v in enumerate(h2_total):
            if i == len(h2_total) - 1:
                log(f"Total h2 : {v} SE: {h2_errs[i]}")
            else:
                log(f"h2_g[{i}] : {v} : {h2_errs[i]}")
        log("*****")
        log("Enrichments: ")
        enr_jack, enr_total = est.enrichment(h2_jack, h2_total, self.M, ne)
        enr_errs = self.estimate_error(enr_jack)
        for i, v in enumerate(enr_total):
            log(f"Enrichment g[{i}] : {v} SE : {enr_errs[i]}")
        log("*****\n*****\nHeritabilities and enrichments computed based on overlapping setting")
        h2o_jack, h2o_total = est.h2_overlapping(sigma_jack, sigma_total, ne, self.annot_matrix, self.M,
                                                 self.num_snp, self.num_jack)
        h2o_errs = self.estimate_error(h2o_jack)
        log("Heritabilities:")
        for i, v in enumerate(h2o_total):
            if i == len(h2_total) - 1:
                log(f"Total h2 : {v} SE: {h2o_errs[i]}")
            else:
                log(f"h2_g[{i}] : {v} : {h2o_errs[i]}")
        log("Enrichments (overlapping def):")
        enro_jack, enro_total = est.enrichment(h2o_jack, h2o_total, self.M, ne)
        enro_errs = self.estimate_error(enro_jack)
        for i, v in enumerate(enro_total):
            log(f"Enrichment g[{i}] : {v} SE : {enro_errs[i]}")
        if self.binary_pheno:
            # the reference calls a method that does not exist here (rhe.py:84,87 vs base.py:857);
            # this build computes the liability transform it intended when prevalences are given.
            if self.samp_prev is not None and self.pop_prev is not None:
                log("*****")
                log("Liability Scale h2 for binary phenotype:")
                for i, v in enumerate(h2_total):
                    out = est.liability_scale(v, h2_errs[i], float(self.pop_prev), float(self.samp_prev))
                    name = "Total Liability-scale h2" if i == len(h2_total) - 1 else f"Liability-scale h2_g[{i}]"
                    log(f"{name} : {out[0]}, SE: {out[1]}, p-value: {out[2]}")
        self.sigma_est_jackknife = sigma_jack
        return {
            "sigma_ests_total": sigma_total, "sig_errs": sig_errs,
            "h2_total": h2_total, "h2_errs": h2_errs,
            "enrichment_total": enr_total, "enrichment_errs": enr_errs,
            "h2_total_overlap": h2o_total, "h2_errs_overlap": h2o_errs,
            "enrichment_total_overlap": enro_total, "enrichment_errs_overlap": enro_errs,
        }


class StreamingRHE(RHE):
    """``pyrhe/src/models/rhe/streaming_rhe.py`` semantics (one-worker streaming; quirk Q4 unless compat='fixed')."""
    streaming = True


class RHE_DOM(Base):
    """Additive + dominance components per bin (``pyrhe/src/models/rhe_dom/rhe_dom.py``): estimates 0..nbin-1 use the
    standardized allele counts, estimates nbin..2nbin-1 the dominance encoding (0, 2p, 4p-2) scaled by 1/(2p(1-p))."""

    def get_num_estimates(self):
        return self.num_bin * 2

    def get_M_last_row(self):
        return np.concatenate([self.len_bin, self.len_bin])

    def _make_runner(self) -> SketchRunner:
        return SketchRunner(self._engines, self.geno_dev, self.plan, self.local_blocks, n_orig=self.num_indv_original,
                            missing_indv=self.missing_indv, Z=self.all_zb, cov=self.cov_matrix, impute=self.geno_impute_methods,
                            seed=self.seed, streaming=self.streaming, compat=self.compat, recompute_pass2=self.recompute_pass2,
                            rng_prefix_draws=self._rng_prefix, rng_prefix_high=self._rng_high,
                            materialize_xxuz=self.materialize_xxuz, dom=True)

    def run(self, method):
        log = self.log._log
        sigma_jack, sigma_total = self.estimate(method=method)
        sig_errs = self.estimate_error(sigma_jack)
        ne = self.num_estimates
        log("Variance components: ")
        for i, v in enumerate(sigma_total):
            if i == len(sigma_total) - 1:
                log(f"Sigma^2_e : {v}  SE : {sig_errs[i]}")
            else:
                log(f"Sigma^2_g[{i}] : {v}  SE : {sig_errs[i]}")
        h2_jack, h2_total = est.h2_nonoverlapping(sigma_jack, sigma_total, ne)
        h2_errs = self.estimate_error(h2_jack)
        log("*****")
        log("Heritabilities:")
        for i, v in enumerate(h2_total):
            if i == len(h2_total) - 1:
                log(f"Total h2 : {v} SE: {h2_errs[i]}")
            else:
                log(f"h2_g[{i}] : {v} : {h2_errs[i]}")
        log("*****")
        log("Enrichments: ")
        enr_jack, enr_total = est.enrichment(h2_jack, h2_total, self.M, ne)
        enr_errs = self.estimate_error(enr_jack)
        for i, v in enumerate(enr_total):
            log(f"Enrichment g[{i}] : {v} SE : {enr_errs[i]}")
        self.sigma_est_jackknife = sigma_jack
        return {"sigma_ests_total": sigma_total, "sig_errs": sig_errs, "h2_total": h2_total, "h2_errs": h2_errs,
                "enrichment_total": enr_total, "enrichment_errs": enr_errs}


class StreamingRHE_DOM(RHE_DOM):
    """``pyrhe/src/models/rhe_dom/streaming_rhe_dom.py`` semantics (one-worker streaming)."""
    streaming = True


class GENIE(Base):
    """Gene-environment model (``pyrhe/src/models/genie/genie.py``): G, G+GxE or G+GxE+NxE with ONE environment
    column named ``env``.  Non-streaming semantics, including the reference's treatment of the last estimate of
    "G+GxE" as the heteroscedastic-noise term (genie.py:18 with base.py:467-481) unless ``compat="fixed"``.
    ``genie_model="G"`` raises IndexError in the reference's aggregate; here it is the plain G model."""

    def __init__(self, env_file: str = None, genie_model: str = "G+GxE+NxE", **kwargs):
        super().__init__(**kwargs)
        if genie_model not in ("G", "G+GxE", "G+GxE+NxE"):
            raise ValueError("Unsupported GENIE genie_model type")
        self.num_env, env = rio.read_env(env_file)
        # the reference keeps the env rows of dropped individuals (and then fails to broadcast); this build drops them
        self.env = np.delete(np.asarray(env, dtype=np.float64), self.missing_indv, axis=0)
        self.num_gen_env_bin = self.num_bin * self.num_env
        self.genie_model = genie_model
        self.log._log(f"Number of environments: {self.num_env}")
        self.log._log(f"GENIE model: {self.genie_model}")
        self._grow_engines(has_gxe=genie_model in ("G+GxE", "G+GxE+NxE"))   # e.g. [Z, UZ | y, e.y] and [e.Z, e.UZ]

    def get_num_estimates(self):
        if self.genie_model == "G":
            return self.num_bin
        if self.genie_model == "G+GxE":
            return self.num_bin + self.num_gen_env_bin
        return self.num_bin + self.num_gen_env_bin + self.num_env

    def get_M_last_row(self):
        if self.genie_model == "G":
            return self.len_bin
        if self.genie_model == "G+GxE":
            return np.concatenate((self.len_bin, self.len_bin * self.num_env))
        return np.concatenate((self.len_bin, self.len_bin * self.num_env, [1] * self.num_env))

    def _make_runner(self) -> SketchRunner:
        return SketchRunner(self._engines, self.geno_dev, self.plan, self.local_blocks, n_orig=self.num_indv_original,
                            missing_indv=self.missing_indv, Z=self.all_zb, cov=self.cov_matrix, env=self.env,
                            genie_model=self.genie_model, impute=self.geno_impute_methods, seed=self.seed,
                            streaming=self.streaming, compat=self.compat, recompute_pass2=self.recompute_pass2,
                            rng_prefix_draws=self._rng_prefix, rng_prefix_high=self._rng_high,
                            materialize_xxuz=self.materialize_xxuz)

    def estimate(self, method: str = "lstsq"):
        """Variance components and their trace-adjusted copies sigma_i * T[i, ne] (genie.py:97-144)."""
        ne = self.num_estimates
        T = self.systems.T[:self.num_jack + 1]
        rows = est.solve_all(T, self.systems.q[:self.num_jack + 1], method)
        adj = rows * T[:, :rows.shape[1], ne]
        return rows[:-1, :], rows[-1, :], adj[:-1, :], adj[-1, :]

    def run(self, method):
        log = self.log._log
        nb, ng, nv = self.num_bin, self.num_gen_env_bin, self.num_env
        has_gxe = self.genie_model in ("G+GxE", "G+GxE+NxE")
        has_nxe = self.genie_model == "G+GxE+NxE"
        sigma_jack, sigma_total, adj_jack, adj_total = self.estimate(method=method)
        sig_errs = self.estimate_error(sigma_jack)
        log("Variance components: ")
        for i, v in enumerate(sigma_total):
            last = i == len(sigma_total) - 1
            if self.genie_model == "G":
                if not last:
                    log(f"Sigma^2_g[{i}] : {v}  SE : {sig_errs[i]}")
            elif self.genie_model == "G+GxE":
                # the reference's loop has no upper bound here, so sigma_e is ALSO echoed as the last "gxe" line
                # (genie.py:232-236); kept for log compatibility
                if i < nb:
                    log(f"Sigma^2_g[{i}] : {v}  SE : {sig_errs[i]}")
                else:
                    log(f"Sigma^2_gxe[{i - nb}] : {v}  SE : {sig_errs[i]}")
            else:
                if i < nb:
                    log(f"Sigma^2_g[{i}] : {v}  SE : {sig_errs[i]}")
                elif i < nb + ng:
                    log(f"Sigma^2_gxe[{i - nb}] : {v}  SE : {sig_errs[i]}")
                elif i < nb + ng + nv:
                    log(f"Sigma^2_nxe[{i - nb - ng}] : {v}  SE : {sig_errs[i]}")
        log(f"Sigma^2_e : {sigma_total[-1]}  SE : {sig_errs[-1]}")
        h2_jack, h2_total = est.genie_h2(adj_jack, adj_total, nb, ng if has_gxe else 0, nv if has_nxe else 0)
        h2_errs = self.estimate_error(h2_jack)
        log("*****")
        log("Heritabilities:")
        n_comp = nb + (ng if has_gxe else 0) + (nv if has_nxe else 0)
        for i, v in enumerate(h2_total):
            if i < nb:
                log(f"h2_g[{i}] : {v} SE : {h2_errs[i]}")
            elif has_gxe and i < nb + ng:
                log(f"h2_gxe[{i - nb}] : {v} SE : {h2_errs[i]}")
            elif has_nxe and i < n_comp:
                log(f"h2_nxe[{i - nb - ng}] : {v} SE : {h2_errs[i]}")
            elif i == n_comp:
                log(f"Total h2 : {v} SE: {h2_errs[i]}")
            elif i == n_comp + 1:
                log(f"Total h2_g : {v} SE: {h2_errs[i]}")
            elif i == n_comp + 2:
                log(f"Total h2_gxe : {v} SE: {h2_errs[i]}")
        log("*****")
        log("Enrichments:")
        enr_jack, enr_total = est.genie_enrichment(h2_jack, h2_total, self.M[-1], nb)
        enr_errs = self.estimate_error(enr_jack)
        for i, v in enumerate(enr_total):
            log(f"Enrichment g[{i}] : {v} SE : {enr_errs[i]}")
        self.sigma_est_jackknife = sigma_jack
        return {"sigma_ests_total": sigma_total, "sig_errs": sig_errs, "h2_total": h2_total, "h2_errs": h2_errs,
                "enrichment_total": enr_total, "enrichment_errs": enr_errs}


class StreamingGENIE(GENIE):
    """``pyrhe/src/models/genie/streaming_genie.py`` semantics (one-worker streaming).  The reference's streaming
    aggregate raises for "G+GxE" (base_streaming.py:45-46 indexes one estimate too far); here that model simply has
    no N x E term."""
    streaming = True
