"""Streaming driver: the host-side mirror of ``StreamingBase`` around the HIP sketch engine.

Reference being mirrored (``/root/reference``): ``pyrhe/src/base/base_streaming.py:21-174`` (two-pass
streaming, aggregate, per-jackknife solve), ``pyrhe/src/base/base.py:362-379`` (block indexing),
``:315-336`` (bin partition), ``:265-289`` (imputation), ``:530-533`` (work split), ``:568-628``
(normal equations) and ``pyrhe/src/models/rhe/streaming_rhe.py:13-43`` (per-block bodies).

MI355X-first differences (results identical within the stated tolerance):
  * a jackknife block is sketched by ONE engine call on the packed 2-bit rows (all k probe vectors,
    covariate-projected vectors and the phenotype at once) instead of 2k float32 GEMVs on a decoded matrix;
  * per-block sketches stay in HBM (2 GB at N=500k, jn=100, k=10), so "pass 2" is a subtraction;
    ``recompute_pass2=True`` re-sketches every block instead, like the reference's second pass;
  * blocks are sharded over ranks (one process per GPU) by the reference's own contiguous rule and the
    per-bin accumulators are summed with ONE all-reduce (RCCL over xGMI); leave-one-out systems are built
    where the block lives and only ``(ne+1)^2 + ne + 1`` numbers per block are gathered;
  * the Gram matrices of the normal equations and the covariate projector U = W Q W^T come from the engine
    (``pyrhe_engine_loo_dots*``, ``pyrhe_engine_apply_projector``); torch holds memory and does [blocks, ne, ne] algebra;
  * with covariates, ``X X^T (U z)`` (base.py:410-412) is kept as its projection ``W^T XXUz = S (Q W^T z)`` only -- all that
    setup_lhs_rhs_jackknife reads of it (base.py:589-591; UXXz lies in span(W)): the covariates go through the first
    contraction as CROSS columns and the engine returns ``S = W^T X X^T W`` per (block, bin) instead of k more sketch columns
    (``SketchRunner.cross``; ``materialize_xxuz=True`` keeps the reference's arrays).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import os

import numpy as np
import torch
import torch.distributed as dist



# ----------------------------------------------------------------------------- block plan
def block_range(j: int, n_snp: int, n_jack: int) -> Tuple[int, int]:
    """SNP rows of jackknife block j: ``M // jn`` each, the last block also takes ``M % jn`` (base.py:362-371)."""
    step = n_snp // n_jack
    start = j * step
    return start, start + (step if j < n_jack - 1 else step + n_snp % n_jack)


def shard_ranges(n_jobs: int, n_workers: int) -> List[Tuple[int, int]]:
    """Contiguous ``ceil(jobs/workers)`` ranges per worker/GPU (base.py:530-533)."""
    per = int(np.ceil(n_jobs / n_workers))
    return [(min(i * per, n_jobs), min((i + 1) * per, n_jobs)) for i in range(n_workers)]


class BlockPlan:
    """Block indexing and bin membership: which SNP rows every (block, bin) sketch covers."""

    def __init__(self, annot: np.ndarray, n_jack: int):
        self.annot = np.asarray(annot)
        self.n_snp, self.n_bin = self.annot.shape
        self.n_jack = int(n_jack)
        self.len_bin = (self.annot == 1).sum(axis=0).astype(np.int64)
        self.max_block = (self.n_snp // self.n_jack) + self.n_snp % self.n_jack

    def range(self, j: int) -> Tuple[int, int]:
        return block_range(j, self.n_snp, self.n_jack)

    def entries(self, j: int) -> Tuple[np.ndarray, np.ndarray]:
        """Row indices (relative to the block) per bin, concatenated, and offsets (partition_bins, base.py:315-336:
        a SNP may sit in several bins or in none)."""
        s, e = self.range(j)
        sub = self.annot[s:e]
        lists = [np.nonzero(sub[:, b])[0].astype(np.int32) for b in range(self.n_bin)]
        offs = np.zeros(self.n_bin + 1, np.int64)
        offs[1:] = np.cumsum([len(x) for x in lists])
        cat = np.concatenate(lists) if offs[-1] else np.zeros(0, np.int32)
        return cat, offs

    def block_counts(self, j: int) -> np.ndarray:
        """SNPs of block j in every bin (what partition_bins would select, base.py:315-336)."""
        if not hasattr(self, "_counts"):
            member = (self.annot != 0).astype(np.int64)
            self._counts = np.stack([member[slice(*self.range(jj))].sum(axis=0) for jj in range(self.n_jack)])
        return self._counts[j]


# ----------------------------------------------------------------------------- imputation replay
def hwe_fill_values(sum_obs: np.ndarray, n_obs: np.ndarray, uniforms: np.ndarray) -> np.ndarray:
    """Genotype substituted for the missing calls of each SNP ("binary" mode, base.py:265-289): the observed mean
    is float32 (``nanmean`` of a float32 column: float64 quotient rounded to float32), ``p = mean * 0.5`` and the HWE
    thresholds are float32 arithmetic; one uniform per SNP decides 0/1/2."""
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = (sum_obs.astype(np.float64) / n_obs.astype(np.float64)).astype(np.float32)
    p = mean * np.float32(0.5)
    one = np.float32(1)
    p0 = (one - p) * (one - p)
    p1 = np.float32(2) * p * (one - p)
    u = np.asarray(uniforms, np.float64)
    return np.where(u < p0.astype(np.float64), 0, np.where(u < (p0 + p1).astype(np.float64), 1, 2)).astype(np.float32)


# ----------------------------------------------------------------------------- the runner
@dataclass
class Systems:
    """Normal equations of every jackknife slot: rows 0..jn-1 leave block j out, row jn is the whole data."""
    T: np.ndarray        # [jn+1, ne+1, ne+1]   (does not depend on the phenotype)
    q: np.ndarray        # [jn+1, ne+1]         first phenotype of the run
    M: np.ndarray        # [jn+1, ne]
    q_all: Optional[np.ndarray] = None   # [n_traits, jn+1, ne+1]: every phenotype sketched in the same pass


MAX_ENGINE_COLS = 30            # sketch + norm columns of one engine pass (PYRHE_MAX_COLS_FP16 of include/pyrhe_sketch.h)
MAX_ENGINE_COLS_I8 = 46         # PYRHE_MAX_COLS: an int8 engine takes cross columns beyond the 30, while its packed digit rows
MAX_DIGIT_ROWS_I8 = 96          # (3 per sketch / norm column, `cross_digits` per cross column, 1 for the ones column) fit six tiles
MAX_SEGMENTS = 64               # kMaxSeg of the engine: (block, bin) lists one kernel chunk can hold
SEG_CAP = 15104                 # kG2pSegCap of the engine: entries of ONE list a kernel chunk takes (GEMM 2 keeps a segment's rows in LDS);
                                # a longer list ends the chunk -- and with it the GEMM 1 launch -- at every 15 104 entries


def engine_capacity(plan: "BlockPlan", n_samples: Optional[int] = None) -> int:
    """``max_entries`` / ``sub_snps`` to create the engines with: at least one block, and room (131 072 entries per kernel
    chunk) to sketch several blocks in one call -- see SketchRunner.sketch_local_blocks.  Measured on MI355X: round 1
    (16 384 -> 65 536) N=50k 71 -> 122 M SNPs/s, N=500k 15.3 -> 16.2 M; round 2, once a call may pass the 32-bit row
    offsets (PYRHE_OPT_LIST_SPAN) and GEMM 1 picks its split count by how the workgroups fill the chip: 65 536 / 131 072 /
    262 144 entries = 6 / 13 / 26 blocks per call at N=500k: 21.4 / 21.9 / 21.3 M SNPs/s, N=50k 144 / 162 / 165 M.  Round 3
    (with GEMM 1's 32-steps-per-split rule): N=50k 131 072 / 262 144 / 524 288 entries: 167.7 / 174.8 / 169.2 M SNPs/s; N=100k
    90.3 / 89.3 M -- so cohorts of up to 65 536 samples (``n_samples``, when the caller gives it) take 262 144."""
    default = 262144 if (n_samples is not None and n_samples <= 65536) else 131072
    return max(int(plan.max_block), int(os.environ.get("PYRHE_CAPACITY", str(default))))


def column_digits(cols: np.ndarray, base: int) -> np.ndarray:
    """int8 digits for each cross column ``cols [N, c]`` in the first contraction, given the count ``base`` that a well-behaved
    column gets (SketchRunner.cross_digits: 1 from N = 16 384 on).  One digit scales a column by 126 / max|w|: its quantisation
    error is 0.0023 max|w| per sample (uniform within half a step), i.e. 0.0023 x (max|w| / rms(w)) of the column's own scale --
    1 % for a Gaussian covariate (max / rms = 5 at N = 500k), but a skewed or squared covariate with max / rms in the hundreds
    would be quantised to zeros almost everywhere (advisor, round 3).  So: one digit only up to max / rms = 8 (<= 2 % of the
    rms), two digits (error x 1 / 256) up to 2048, three beyond.  A column that one digit holds EXACTLY (an intercept, 0 / 1
    indicators however rare, counts 0 / 1 / 2 / 3) keeps one digit whatever its ratio."""
    cols = np.asarray(cols, np.float64)
    out = np.full(cols.shape[1], int(base), np.int64)
    if base >= 3:
        return out
    mx = np.abs(cols).max(axis=0)
    rms = np.sqrt(np.mean(cols * cols, axis=0))
    for j in range(cols.shape[1]):
        if not (mx[j] > 0) or not np.isfinite(mx[j]):
            continue
        q = cols[:, j].astype(np.float32) * np.float32(126.0 / mx[j])
        if np.array_equal(q, np.rint(q)):
            continue                                     # exact in one digit
        ratio = mx[j] / max(rms[j], 1e-300)
        need = 1 if ratio <= 8.0 else (2 if ratio <= 2048.0 else 3)
        out[j] = max(int(base), need)
    return out


def plan_passes(k: int, n_groups: int, n_norm: int, cross=(), cross_digits: int = 3, wide: bool = False, cross_rows=None):
    """Distribute ``n_groups`` groups of ``k`` sketch columns (Z, UZ, e.Z, e.UZ) and ``n_norm`` norm columns (all in
    pass 0) over engine passes of at most 30 right-hand columns.  A group stays whole when it fits into the current
    pass or into a fresh one; only groups wider than a pass (num_random_vec > 30 - n_norm) are split.
    Returns, per pass, a list of chunks ``(group index, first column s0, number of columns)``.
    ``cross``: sizes of cross-column groups (the covariates W, e.W: each stays whole, its Gram matrix needs all pairs in one
    pass); then the result is ``(passes, cross_at)`` with ``cross_at[i] = (pass, first cross column of that pass)`` -- the
    first pass with room, or a pass of its own (one GEMM 1 over the genotypes, no GEMM 2).  ``wide``: int8 engines, which
    take up to 46 columns as long as the packed digit rows of the pass fit six 16-row tiles (``cross_digits`` rows per cross
    column: GENIE G x E + 10 one-digit covariates = 20 + 2 + 20 columns = 87 rows in ONE pass).  ``cross_rows``: the packed digit
    rows of each cross group where its columns carry different digit counts (column_digits; default ``cross_digits`` x size)."""
    if cross:
        passes = plan_passes(k, n_groups, n_norm)
        cap = MAX_ENGINE_COLS_I8 if wide else MAX_ENGINE_COLS
        cols3 = [sum(nc for _, _, nc in ch) + (n_norm if pi == 0 else 0) for pi, ch in enumerate(passes)]    # three-digit columns
        used, used_rows, cross_at = [0] * len(passes), [0] * len(passes), []
        rows_of = [int(r) for r in cross_rows] if cross_rows is not None else [cross_digits * c for c in cross]

        def fits(pi, c, r):
            rows = 3 * cols3[pi] + used_rows[pi] + r + 1
            return cols3[pi] + used[pi] + c <= cap and (not wide or rows <= MAX_DIGIT_ROWS_I8)
        for c, r in zip(cross, rows_of):
            if c > MAX_ENGINE_COLS or (wide and r + 1 > MAX_DIGIT_ROWS_I8):
                raise ValueError(f"a cross group of {c} columns does not fit one engine pass")
            pi = next((i for i in range(len(passes)) if fits(i, c, r)), None)
            if pi is None:
                passes.append([]); cols3.append(0); used.append(0); used_rows.append(0)
                pi = len(passes) - 1
            cross_at.append((pi, used[pi]))
            used[pi] += c
            used_rows[pi] += r
        return passes, cross_at
    passes, free = [[]], [MAX_ENGINE_COLS - n_norm]
    for g in range(n_groups):
        if k <= free[-1]:
            passes[-1].append((g, 0, k)); free[-1] -= k
        elif k <= MAX_ENGINE_COLS:
            passes.append([(g, 0, k)]); free.append(MAX_ENGINE_COLS - k)
        else:
            s0 = 0
            while s0 < k:
                if free[-1] == 0:
                    passes.append([]); free.append(MAX_ENGINE_COLS)
                nc = min(free[-1], k - s0)
                passes[-1].append((g, s0, nc)); free[-1] -= nc
                s0 += nc
    return passes


def n_engines_needed(k: int, use_cov: bool, has_gxe: bool) -> int:
    """Engines (= passes over the genotypes) with the U Z sketch columns materialised; the covariates-as-cross-columns
    form is only chosen when it needs no more (SketchRunner.cross)."""
    return len(plan_passes(k, (2 if use_cov else 1) * (2 if has_gxe else 1), 2 if has_gxe else 1))


def cross_columns_pay(k: int, c: int, has_gxe: bool, cross_digits: int = 3, wide: bool = False, cross_rows=None) -> bool:
    """Whether the covariates go in as CROSS columns (c more columns in GEMM 1, their c x c Gram per (block, bin) out)
    instead of the k sketch columns U Z (k more columns in both contractions, [k, N] rows per (block, bin) out): when that
    takes no more passes over the genotypes than the materialised form."""
    if c < 1 or c > MAX_ENGINE_COLS:
        return False
    g = 2 if has_gxe else 1
    passes, _ = plan_passes(k, g, g, cross=[c] * g, cross_digits=cross_digits, wide=wide, cross_rows=cross_rows)
    return len(passes) <= n_engines_needed(k, True, has_gxe)


class SketchRunner:
    """Pass 1 (sketch every local block, accumulate, all-reduce) and pass 2 (leave-one-out systems).

    ``engines``: one :class:`pyrhe_amd.engine.SketchEngine`, or two when the right-hand columns do not fit one
    engine (GENIE G x E with covariates: [Z, UZ | y] and [e.Z, e.UZ | e.y]).  Anything with the engine's
    ``set_rhs/sketch_block/block_stats/keep_vector`` interface works: the multi-process CPU tests inject a
    stand-in, the product never does.  ``geno`` holds the packed rows of THIS rank's blocks ``[j0, j1)`` on the
    device, first row = ``range(j0)[0]``.

    Estimates (rows of the normal equations), in the reference's order (genie.py:26-44): one per annotation bin
    (G), then one per bin for G x E, then one N x E term.
    """

    def __init__(self, engines, geno: torch.Tensor, plan: BlockPlan, local_blocks: Tuple[int, int], *,
                 n_orig: int, missing_indv: Sequence[int], Z: np.ndarray, cov: Optional[np.ndarray] = None,
                 env: Optional[np.ndarray] = None, genie_model: Optional[str] = None,
                 impute: str = "binary", seed: int = 0, streaming: bool = True, compat: str = "reference",
                 recompute_pass2: bool = False, rng_prefix_draws: int = 0, group=None, dom: bool = False,
                 solo: bool = False, rng_prefix_high: int = 2, materialize_xxuz: bool = False,
                 force_collectives: bool = False, cross_digits: Optional[int] = None):
        self.engs = list(engines) if isinstance(engines, (list, tuple)) else [engines]
        self.eng = self.engs[0]
        self.geno = geno
        self.plan = plan
        self.j0, self.j1 = local_blocks
        self.n_orig = int(n_orig)
        self.missing = [int(i) for i in missing_indv]
        self.keep_idx = np.setdiff1d(np.arange(self.n_orig), np.asarray(self.missing, dtype=np.int64))
        self.N = len(self.keep_idx)
        self.Z = np.asarray(Z, np.float64)
        self.k = self.Z.shape[1]
        self.cov = None if cov is None else np.asarray(cov, np.float64)
        self.env = None if env is None else np.asarray(env, np.float64).reshape(-1)
        if genie_model not in (None, "G", "G+GxE", "G+GxE+NxE"):
            raise ValueError("Unsupported GENIE genie_model type")
        self.genie_model = genie_model
        self.impute = impute
        self.seed = int(seed)
        self.streaming = bool(streaming)
        if compat not in ("reference", "fixed"):
            raise ValueError("compat must be 'reference' or 'fixed'")
        self.compat = compat
        self.recompute_pass2 = bool(recompute_pass2)
        self.rng_prefix_draws = int(rng_prefix_draws)
        self.rng_prefix_high = int(rng_prefix_high)     # generate_annot drew randint(0, num_bin) per SNP (rejection sampling: the bound matters)
        self.group = group
        # solo: a runner over data every rank holds by itself (a cross-check beside a sharded job): no collectives
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized() and not solo) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # the cross-rank exchange of pass 1 / pass 2 runs when there is more than one rank -- or, for the one-GPU rehearsal of
        # the RCCL path (tests/test_distributed_gpu.py), on a one-rank group when asked to
        self.collective = self.world > 1 or bool(force_collectives and dist.is_available() and dist.is_initialized() and not solo)
        self.device = geno.device
        self.n_pad = self.eng.n_pad
        self.row0 = plan.range(self.j0)[0] if self.j1 > self.j0 else 0
        self._keep_t = torch.as_tensor(self.keep_idx, device=self.device)
        self.keep = self.eng.keep_vector(self.missing)
        nb = plan.n_bin
        self.n_gxe = nb if genie_model in ("G+GxE", "G+GxE+NxE") else 0
        self.n_nxe = 1 if genie_model == "G+GxE+NxE" else 0
        self.dom = bool(dom)          # RHE_DOM: a dominance estimate per bin after the additive ones (rhe_dom.py:7-13)
        if self.dom and genie_model is not None:
            raise ValueError("dominance and GENIE terms are separate models")
        self.n_est = 2 * nb if self.dom else nb + self.n_gxe + self.n_nxe
        # Reference quirk (base.py:467,472-481): the NON-streaming GENIE treats its LAST estimate as the
        # heteroscedastic-noise term for every genie_model (GENIE sets num_env = 1, genie.py:18), so for "G+GxE" the
        # whole-data slot of the last GxE bin is overwritten by the N x E sketch.  ("G" raises IndexError there and
        # the streaming aggregate raises for "G+GxE" (base_streaming.py:45-46), so those have no reference behaviour.)
        self.quirk_last_total = genie_model == "G+GxE" and (not self.streaming) and compat == "reference"
        # covariate projector pieces (base.py:151,177-178)
        if self.cov is not None:
            W = self.cov
            self.Q = np.linalg.pinv(W.T @ W)
            self.UZ = W @ self.Q @ (W.T @ self.Z)
            self.W64 = self._pad_rows(torch.from_numpy(np.ascontiguousarray(W)))          # [n_pad, c] f64
            self.W32 = self.W64.to(torch.float32)
            self.Q64 = torch.from_numpy(self.Q).to(self.device)
            self.Q32 = self.Q64.to(torch.float32)
            self.UZ64 = self._pad_cols(torch.from_numpy(np.ascontiguousarray(self.UZ.T)))  # [k, n_pad]
            self.WT64 = self.W64.T.contiguous()                                              # [c, n_pad]
            self.WT32 = self.WT64.to(torch.float32)
            self.G64 = torch.from_numpy(self.Q @ (W.T @ self.Z)).to(self.device)             # [c, k]: U Z = W G
        # Covariates: XXUz = X X^T (U Z) enters the normal equations only through sum(XXUz * UXXz) (base.py:589-591) and
        # UXXz lies in span(W), so only W^T XXUz = (W^T X X^T W)(Q W^T Z) is needed: the c covariate columns go through the
        # first contraction as CROSS columns and the engine returns S = W^T X X^T W per (block, bin) -- no U Z sketch
        # columns, half the second contraction, half the stored rows.  ``materialize_xxuz``: the k U Z sketch columns
        # instead (the reference's own arrays; block-wise XXUz tests, or more covariates than a pass has room for).
        self.cross_digits = int(cross_digits) if cross_digits else (1 if self.N >= 16384 else 3)      # see below
        self.wide_cols = getattr(self.eng, "precision", "") in ("i8x3", "i8f16")       # int8 engines: up to 46 columns per pass
        # per column: a covariate whose largest value is far above its rms takes more digits than the rest (column_digits); a
        # count given by the caller applies to every column
        self.cross_col_digits = {}
        if self.cov is not None:
            forced = bool(cross_digits)
            dg = lambda cols: (np.full(cols.shape[1], self.cross_digits, np.int64) if forced else column_digits(cols, self.cross_digits))
            self.cross_col_digits["W"] = dg(self.cov)
            if self.n_gxe:
                self.cross_col_digits["eW"] = dg(self.cov * self.env[:, None])
        self.cross = (self.cov is not None and not materialize_xxuz and
                      cross_columns_pay(self.k, self.cov.shape[1], self.n_gxe > 0, self.cross_digits, self.wide_cols, self._cross_rows()))
        # int8 digits of the covariate (cross) columns in the first contraction (PYRHE_COL_DIGITS): S only enters the O(c / N) terms
        # of T.  Measured on the CPU stand-in (N = 500 / 2000, 6 covariates, 3 bins): two digits move sigma^2 / SE by 4e-8 /
        # 8e-8 at N = 500; ONE digit by 6e-6 / 1e-5 at N = 500, 2e-7 / 1.6e-6 at N = 2000 (it falls as 1 / N^2: below 3e-8
        # from N = 16 384 on).  One digit packs 10 probes + phenotype + 10 covariates into 44 digit rows = three tiles (the
        # 128-row GEMM 1) instead of four: 8 bins + 10 covariates at N = 500k 30.5 -> 23.7 ms per pass in GEMM 1.  S itself
        # then carries the 1 % quantisation noise of an 8-bit W (exact for an intercept or 0/1 covariates); small cohorts,
        # where a tile more costs nothing, keep all three digits.  ``cross_digits`` overrides (tests hold S to 2e-5 at 3).
        self.Z64 = self._pad_cols(torch.from_numpy(np.ascontiguousarray(self.Z.T)))        # [k, n_pad] f64
        if self.env is not None:
            self.env_pad = self._pad_cols(torch.from_numpy(self.env[None, :]))[0]            # [n_pad] f64
            self.env32 = self.env_pad.to(torch.float32)
        self.pass2_impl = "hip"       # "torch": materialised cross-check path (tests)
        # long blocks (lists beyond SEG_CAP entries, rows beyond the 32-bit span) go into their call as row pieces = list windows
        # of their own where the engine allows (_windows); PYRHE_BATCH_PIECES=0: the engine's own chunking / one call per piece
        self.batch_pieces = os.environ.get("PYRHE_BATCH_PIECES", "1") != "0"
        self.force_tk_proj = False    # tests: <XXz, U Z> through the covariate projections even where the U Z row has room (_loo_systems)
        self.norm_digits = 3          # int8 digits of the phenotype columns in GEMM 1 (precision experiments: tools/ab.sh [ydigits])
        self.one_slot = False         # this trait group keeps ONE block slot and re-sketches in pass 2 (begin_trait decides)
        self.row_limit = None         # rows one engine call may address; None = from the engine's 32-bit byte offsets
        self._span_ok = None          # see _lists_may_span_calls
        self._entries_cache = {}
        self._stats = None
        self._check_missing = True
        self.timings = {}

    def _cross_rows(self):
        """Packed digit rows of the cross groups W [, e.W] (plan_passes)."""
        return [int(self.cross_col_digits[n_].sum()) for n_ in (("W", "eW") if self.n_gxe else ("W",))]

    # ---- padding helpers: kept samples -> their original positions, zeros elsewhere
    def _pad_cols(self, t: torch.Tensor) -> torch.Tensor:
        out = torch.zeros((t.shape[0], self.n_pad), dtype=t.dtype, device=self.device)
        out[:, self._keep_t] = t.to(self.device)
        return out

    def _pad_rows(self, t: torch.Tensor) -> torch.Tensor:
        out = torch.zeros((self.n_pad, t.shape[1]), dtype=t.dtype, device=self.device)
        out[self._keep_t] = t.to(self.device)
        return out

    # ---- right-hand sides ---------------------------------------------------------------------
    def _install_rhs(self, y: np.ndarray):
        """Columns handed to the engine(s).  Sketch groups of k columns: Z, [UZ] and for G x E e.Z, [e.UZ];
        norm columns: y~ and e.y~  (X_e^T r = X^T (e . r), X_e v = e . (X v); streaming_genie.py:33-49)."""
        use_cov = self.cov is not None
        y64 = np.asarray(y, np.float64)
        y64 = y64.reshape(-1, 1) if y64.ndim == 1 else y64                       # [N, n_traits]
        if use_cov:          # regress_pheno (base.py:396-401), float64 on the host, phenotype by phenotype
            W = self.cov
            y64 = np.stack([y64[:, t] - W @ (np.linalg.pinv(W.T @ W) @ (W.T @ y64[:, t])) for t in range(y64.shape[1])], axis=1)
        self.y_res = y64
        self.yy = np.einsum("ij,ij->j", y64, y64)                                # y~^T y~ per phenotype (base.py:625-626)
        self.n_traits = nt = y64.shape[1]
        if nt > self.max_traits():
            raise ValueError(f"{nt} phenotypes do not fit the spare right-hand columns of one pass ({self.max_traits()})")
        ypad = self._pad_cols(torch.from_numpy(np.ascontiguousarray(y64.T)))     # [n_traits, n_pad]
        names, groups, norms = ["Z"], [self.Z64], [ypad]
        cnames, cgroups = [], []                 # cross columns: the covariates W (and e.W for the G x E estimates)
        if use_cov and self.cross:
            cnames.append("W"); cgroups.append(self.WT64)
        elif use_cov:
            names.append("UZ"); groups.append(self.UZ64)
        if self.n_gxe:
            e = self.env_pad
            names.append("eZ"); groups.append(self.Z64 * e)
            if use_cov and self.cross:           # X_e X_e^T (U Z) = e.(X X^T (e.W)) G: W^T of it is the Gram of the columns e.W
                cnames.append("eW"); cgroups.append(self.WT64 * e)
            elif use_cov:
                names.append("eUZ"); groups.append(self.UZ64 * e)
            norms.append(ypad * e)
        k = self.k
        n_norm_all = sum(n.shape[0] for n in norms)
        if cgroups:
            passes, cross_at = plan_passes(k, len(groups), n_norm_all, cross=[g.shape[0] for g in cgroups],
                                           cross_digits=self.cross_digits, wide=self.wide_cols, cross_rows=self._cross_rows())
        else:
            passes, cross_at = plan_passes(k, len(groups), n_norm_all), []
        if len(passes) > len(self.engs):
            raise RuntimeError(f"{len(groups)} groups of {k} right-hand columns need {len(passes)} engines, "
                               f"{len(self.engs)} given")
        # where each logical group lives: a list of chunks (pass, first engine column, first group column s0, count);
        # norm columns: (pass, column).  `breaks`: the common refinement of all chunk boundaries (pass 2 walks it).
        self.loc = {n: [] for n in names}
        # cross columns of a pass in the order the engine wants them: digit counts must not grow with the column index, so the
        # columns of all cross groups of the pass are sorted by their digits (stable); cross_loc[name] = (pass, where the
        # group's columns sit in that order)
        self.cross_loc, cross_order = {}, {}
        for pi in range(len(passes)):
            mine = [i for i, (cp, _) in enumerate(cross_at) if cp == pi]
            if not mine:
                continue
            dg = np.concatenate([self.cross_col_digits[cnames[i]] for i in mine])
            order = np.argsort(-dg, kind="stable")
            pos = np.empty_like(order)
            pos[order] = np.arange(len(order))
            cross_order[pi] = (order, dg[order])
            o = 0
            for i in mine:
                c_ = cgroups[i].shape[0]
                self.cross_loc[cnames[i]] = (pi, torch.from_numpy(pos[o:o + c_].astype(np.int64)).to(self.device))
                o += c_
        self.pass_shapes, self.pass_cross = [], []
        breaks = {0, k}
        for pi, chunks in enumerate(passes):
            col, parts = 0, []
            for g, s0, nc in chunks:
                self.loc[names[g]].append((pi, col, s0, nc))
                parts.append(groups[g][s0:s0 + nc])
                breaks.update((s0, s0 + nc))
                col += nc
            sk = torch.cat(parts, dim=0).to(torch.float32) if parts else None
            nm = torch.cat(norms, dim=0).to(torch.float32) if pi == 0 else None
            cr = [cgroups[i] for i, (cp, _) in enumerate(cross_at) if cp == pi]
            if cr:
                order, dg = cross_order[pi]
                uniform = int(dg.min()) == int(dg.max())
                self.engs[pi].set_rhs(sk, nm, self.keep, cross_cols=torch.cat(cr, dim=0)[torch.from_numpy(order).to(self.device)].to(torch.float32),
                                      cross_digits=int(dg[0]) if uniform else [int(d_) for d_ in dg],
                                      **({"norm_digits": self.norm_digits} if self.norm_digits != 3 else {}))
            else:
                self.engs[pi].set_rhs(sk, nm, self.keep, **({"norm_digits": self.norm_digits} if self.norm_digits != 3 else {}))
            self.pass_shapes.append((col, n_norm_all if pi == 0 else 0))
            self.pass_cross.append(sum(g.shape[0] for g in cr))
        self.loc["y"] = (0, 0)                 # norm columns of pass 0: y of every phenotype, then e.y of every phenotype
        if self.n_gxe:
            self.loc["ey"] = (0, nt)
        self.n_pass = len(passes)
        bl = sorted(breaks)
        self.intervals = [(bl[i], bl[i + 1] - bl[i]) for i in range(len(bl) - 1)]

    def _loc_at(self, name: str, s0: int):
        """(pass, engine column) holding column ``s0`` of group ``name``."""
        for p_, c0, g0, nc in self.loc[name]:
            if g0 <= s0 < g0 + nc:
                return p_, c0 + (s0 - g0)
        raise KeyError((name, s0))

    # ---- imputation fill values ---------------------------------------------------------------
    def _rows(self, a: int, b: int) -> torch.Tensor:
        """Local rows a .. b - 1 on the device: a slice of the resident buffer, or a window uploaded from the host
        (engine.HostRows: a shard that does not fit HBM)."""
        g = self.geno
        return g.window(a, b) if hasattr(g, "window") else g[a:b]

    def _block_rows(self, j: int) -> torch.Tensor:
        s, e = self.plan.range(j)
        return self._rows(s - self.row0, e - self.row0)

    def _ingest_stats(self):
        """Per-SNP observed sums / missing counts of the local rows (one extra read, once per cohort; with host-resident
        rows one extra pass over the link, in windows of one engine call's rows)."""
        if self._stats is None:
            m = self.geno.shape[0]
            if m == 0:          # a rank without local blocks (world > useful ranks)
                self._stats = (np.zeros(0, np.int64), np.zeros(0, np.int64))
                return self._stats
            if hasattr(self.geno, "window"):
                step = max(1, int(getattr(self.eng, "sub_snps", 0)) or self.plan.max_block)
                parts = [self.eng.block_stats(self._rows(a, min(a + step, m))) for a in range(0, m, step)]
                s, nm = torch.cat([p_[0] for p_ in parts]), torch.cat([p_[1] for p_ in parts])
            else:
                s, nm = self.eng.block_stats(self.geno)
            self._stats = (s.cpu().numpy().astype(np.int64), nm.cpu().numpy().astype(np.int64))
        return self._stats

    # ---- host-resident rows: statistics and fill values window by window, inside the first sketch pass -----------------
    def _lazy_begin(self, pass_index: int):
        m = self.geno.shape[0]
        self._stats_lazy = (np.zeros(m, np.int64), np.zeros(m, np.int64))
        self._lazy_draw = self._uniform_stream(pass_index)
        if not hasattr(self, "_fills"):
            self._fills = {}
        self._fills[pass_index] = [None] * (self.j1 - self.j0)

    def _lazy_window_stats(self, pass_index: int, js, rows: torch.Tensor):
        """Statistics of one call's window ``rows`` (the blocks ``js``) and the fill values of those blocks."""
        a = self.plan.range(js[0])[0] - self.row0
        s_, nm_ = self.eng.block_stats(rows)
        self._stats_lazy[0][a:a + rows.shape[0]] = s_.cpu().numpy()
        self._stats_lazy[1][a:a + rows.shape[0]] = nm_.cpu().numpy()
        self._stats, keep = self._stats_lazy, self._stats          # (_fill_for_block reads the arrays through _ingest_stats)
        try:
            for j in js:
                sj, ej = self.plan.range(j)
                self._fills[pass_index][j - self.j0] = self._fill_for_block(j, self._lazy_draw(j, ej - sj))
        finally:
            self._stats = keep

    def _lazy_end(self):
        self._stats = self._stats_lazy
        del self._stats_lazy, self._lazy_draw

    def _fill_for_block(self, j: int, uniforms: Optional[np.ndarray]) -> Optional[torch.Tensor]:
        sums, nmiss = self._ingest_stats()
        s, e = self.plan.range(j)
        sl = slice(s - self.row0, e - self.row0)
        if not nmiss[sl].any():
            return None
        if self.impute == "binary":
            f = hwe_fill_values(sums[sl], self.N - nmiss[sl], uniforms)
        else:
            f = np.zeros(e - s, np.float32)       # "mean": missing -> 0 before standardisation (base.py:287)
        return torch.from_numpy(f).to(self.device)

    def _uniform_stream(self, pass_index: int):
        """Replay NumPy's legacy stream the way the reference consumes it (base.py:73,176,266,510;
        base_streaming.py:85-104,110-118): non-streaming reseeds per block, streaming keeps drawing.  ``pass_index`` 1
        is the in-process (multiprocessing=False) pass 2, whose stream continues behind pass 1; the one-worker
        multiprocessing mode this package mirrors forks both workers from the same state (pass_index 0 twice)."""
        if self.impute != "binary":
            return lambda j, m: None
        if not self.streaming:
            return lambda j, m: np.random.RandomState(self.seed).random_sample(m)
        rs = np.random.RandomState(self.seed)
        if self.rng_prefix_draws:
            rs.randint(0, self.rng_prefix_high, self.rng_prefix_draws)
        rs.randn(self.N, self.k)
        skip = 0
        for p in range(pass_index + 1):
            upto = self.j0 if p == pass_index else self.plan.n_jack
            for j in range(upto):
                s, e = self.plan.range(j)
                skip += e - s
        if skip:
            rs.random_sample(skip)
        return lambda j, m: rs.random_sample(m)

    # ---- pass 1 --------------------------------------------------------------------------------
    def _entries(self, j: int):
        if j not in self._entries_cache:
            cat, offs = self.plan.entries(j)
            self._entries_cache[j] = (torch.from_numpy(cat).to(self.device), offs)
        return self._entries_cache[j]

    def _blocks_per_call(self) -> int:
        """Consecutive blocks sketched by ONE engine call.  A call takes any number of entry lists, so the (block, bin)
        lists of several blocks can go in together: for small cohorts (N = 50k: 62 MB per block) that means one GEMM 1 /
        GEMM 2 launch per ~3 blocks instead of per block.  Bounds: the lists of a batch fit one kernel chunk (entries and
        segment table), the rows of a batch stay within the engine's 32-bit byte offsets."""
        if self.dom and not self._dom_batches():
            return 1
        cap = int(getattr(self.eng, "sub_snps", 0)) or self.plan.max_block
        by_entries = cap // max(self.plan.max_block, 1)
        by_bytes = self._row_limit() // max(self.plan.max_block, 1)
        if by_bytes < by_entries and self._lists_may_span_calls() and (by_bytes >= 1 or self._window_calls_ok()):
            # the call addresses every list relative to its own lowest row (PYRHE_OPT_LIST_SPAN); blocks beyond the row span of
            # ONE list go in as row pieces, each a window of lists of its own (_windows)
            by_bytes = by_entries
        # Lists per call.  The engine takes more lists than ONE segment table holds (further chunks of the call share its GEMM 1
        # launch).  Cohorts below 196 608 samples gain from that -- GEMM 1 gets rows enough for the 128-row kernel and long launches:
        # N = 50k with 8 bins 5.62 -> 4.86 ms per pass, with 4 bins 4.07 -> 3.62, N = 100k with 8 bins 7.88 -> 7.35 -- and take WHOLE
        # tables (a last table of fewer segments is a short GEMM 2 launch); at N = 500k GEMM 1 gains less than GEMM 2 loses (8 bins +
        # covariates 56.9 -> 57.6 ms, 32 bins 87.7 -> 88.2, 1000-SNP blocks 57.5 -> 57.6; tools/ab.sh [s47]): one table per call there.
        by_lists = max(1, MAX_SEGMENTS // max(self.plan.n_bin, 1))
        if (self.n_pad < 192 * 1024 or os.environ.get("PYRHE_WHOLE_TABLES") == "1") and os.environ.get("PYRHE_ONE_TABLE") != "1":
            by_lists *= max(1, min(by_entries, max(by_bytes, 1)) // by_lists)
        # (Choosing the batch so that GEMM 2's launch fills whole rounds of resident workgroups -- N = 50k: 10 blocks = 490
        # workgroups in one round of 512 instead of 13 = 637 in two -- was measured: GEMM 2 1.34 -> 1.21 ms per pass, GEMM 1
        # 1.66 -> 1.77 ms with two more launches: no gain; the largest batch stays.)
        return max(1, min(by_entries, by_bytes, by_lists))

    def _dom_batches(self) -> bool:
        """RHE_DOM through the engine's segmented flow (several blocks per call; one fused GEMM 2 launch per call -- three with
        dom_fuse = 0 -- instead of three per list): the int8 engine with both genotype views in one GEMM 1 launch (<= 4 tiles) and, for single-bin cohorts, the
        packed GEMM 2 (<= 10 sketch columns, >= 192k samples) -- the engine's own rule (sketch_impl: dom_seg)."""
        if not hasattr(self, "n_pass"):
            return False
        for e_ in self.engs[:self.n_pass]:
            if getattr(e_, "precision", "") != "i8x3" or not (1 <= getattr(e_, "n_sketch", 0)) or not getattr(e_, "options", {}).get("dom_seg", 1):
                return False
            slots = int(e_.lib.pyrhe_engine_info(e_._h, 11)) if hasattr(e_, "lib") else 99
            packed = e_.n_sketch <= 10 and self.n_pad >= 192 * 1024 and getattr(e_, "options", {}).get("g2_packed", 1)
            if slots > 64 or not (self.plan.n_bin > 1 or packed):
                return False
        return True

    def _lists_may_span_calls(self) -> bool:
        """Whether one engine call may address more rows than its 32-bit byte offsets reach (N = 500k: more than three
        10 000-SNP blocks): the int8 path with <= 20 sketch columns in the segmented flow addresses every entry list relative
        to its own lowest row, if the caller vouches that no LIST spans more (here: every list lies inside its block, and a
        block fits the limit).  GEMM 1 pays a fixed cost per launch -- 29.5 / 26.0 / 24.5 ms per pass at 1 / 2 / 3 blocks per
        call at N = 500k -- so larger calls are worth having."""
        if self._span_ok is None:
            if not hasattr(self, "n_pass") or any(getattr(e_, "n_sketch", 0) + getattr(e_, "n_norm", 0) == 0
                                                  for e_ in self.engs[:self.n_pass]):
                return False               # right-hand sides not installed yet (begin_trait): decide later
            ok = (self.row_limit is None and all(hasattr(e_, "set_option") for e_ in self.engs[:self.n_pass])
                  and (not self.dom or (self._dom_batches() and self.n_pad >= 192 * 1024 and all(e_.n_sketch <= 10 for e_ in self.engs[:self.n_pass])))
                  and all(getattr(e_, "precision", "") == "i8x3" and 1 <= e_.n_sketch <= 20 and
                          getattr(e_, "options", {}).get("g2_packed", 1) for e_ in self.engs[:self.n_pass])
                  and (self.plan.n_bin > 1 or self.n_pad >= 192 * 1024))
            if ok:
                from .engine import OPTIONS
                ok = "list_span" in OPTIONS
            if ok:
                # the engine may let lists share a GEMM 2 workgroup (addressed from their common lowest row) only inside one
                # window of n_bin consecutive lists = the bins of ONE block: every block fits the 32-bit span (_pieces
                # otherwise), two neighbouring blocks need not (N = 500k: blocks of more than 17 154 SNPs)
                for e_ in self.engs[:self.n_pass]:
                    e_.set_option("list_span", self.plan.n_bin)
            self._span_ok = bool(ok)
        return self._span_ok

    def _window_calls_ok(self) -> bool:
        """Whether the row pieces of blocks may go into one engine call as list windows of their own (outputs added afterwards):
        the engine addresses every window relative to its own lowest row (PYRHE_OPT_LIST_SPAN)."""
        return bool(self.batch_pieces and self._lists_may_span_calls())

    def _windows(self, js):
        """The blocks ``js`` (consecutive) of ONE engine call as list windows: every block whole, or in its row pieces (_pieces).
        Returns ``(entries, offsets, window -> index in js)``; rows are relative to the first row of the batch."""
        key = ("windows", js[0], js[-1], self._row_limit())
        if key not in self._entries_cache:
            first = self.plan.range(js[0])[0]
            ents, offs, wb, pos = [], [0], [], 0
            for bi, j in enumerate(js):
                base = self.plan.range(j)[0] - first
                pcs = self._pieces(j)
                if pcs is None:
                    ent_j, offs_j = self._entries(j)
                    pcs = [(0, None, ent_j, np.asarray(offs_j, np.int64))]
                for a, _, ent_p, offs_p in pcs:
                    ents.append(ent_p + int(base + a))
                    offs.extend((pos + np.asarray(offs_p[1:], np.int64)).tolist())
                    pos += int(offs_p[-1])
                    wb.append(bi)
            self._entries_cache[key] = (torch.cat(ents).to(torch.int32), np.asarray(offs, np.int64), wb)
        return self._entries_cache[key]

    def _row_limit(self) -> int:
        """Rows one engine call can address: its row byte offsets are 32-bit (rows * pitch < 2^32, rows < 2^24)."""
        if self.row_limit is not None:
            return int(self.row_limit)
        pitch = int(getattr(self.eng, "pitch", self.n_pad // 4))
        return max(1, min(((1 << 32) - 1) // max(pitch, 1), (1 << 24) - 1))

    def _pieces(self, j: int):
        """A block with more rows than one engine call addresses (N = 500k: over 34 308 SNPs, e.g. M = 1M with jn < 30) as
        consecutive row sub-ranges ``(a, b, entries, offsets)`` with the bin lists rebased to each piece; the later pieces
        ACCUMULATE into the block's outputs.  Exact in the reference's sense: standardisation is per SNP, so the block
        sketch is the sum over its SNPs in any grouping (float32 additions in list order either way).  None: one call."""
        s, e = self.plan.range(j)
        lim = self._row_limit()
        # Where the pieces of a call go in together as list WINDOWS (_window_calls_ok), no list of a piece exceeds what one kernel
        # chunk takes of a list (SEG_CAP): a longer list would end the chunk -- one GEMM 1 launch per 15 104 entries, N = 500k with
        # blocks of 16 667 ... 33 333 SNPs: GEMM 1 25.6 - 27.1 ms per pass against 23.1 (tools/ab.sh [s38]).
        lim_list = SEG_CAP if self._window_calls_ok() else None
        if e - s <= lim and (lim_list is None or e - s <= lim_list):
            return None                                             # (the common case, asked for every block of every call: no work here)
        key = ("pieces", j, lim, lim_list)
        if key not in self._entries_cache:
            cat, offs = self.plan.entries(j)
            if e - s <= lim and max(int(offs[b_ + 1] - offs[b_]) for b_ in range(self.plan.n_bin)) <= lim_list:
                self._entries_cache[key] = None                     # more rows than a list may have, but no single list is that long
                return None
            out = []
            bounds, a = [], 0
            n_p = -(-(e - s) // lim)                                # pieces of about equal size, as few as the limits allow
            if lim_list is not None:
                n_p = max(n_p, max(-(-int(offs[b_ + 1] - offs[b_]) // lim_list) for b_ in range(self.plan.n_bin)))
            step = min(lim, -(-(e - s) // n_p))
            while a < e - s:
                b = min(a + step, e - s)
                if lim_list is not None:
                    for bn in range(self.plan.n_bin):
                        l = cat[offs[bn]:offs[bn + 1]]
                        ia = int(np.searchsorted(l, a))
                        if ia + lim_list < len(l):
                            b = min(b, int(l[ia + lim_list]))      # the row of the first entry beyond the limit: exclusive bound
                bounds.append((a, b))
                a = b
            for a, b in bounds:
                lists = []
                for bn in range(self.plan.n_bin):
                    l = cat[offs[bn]:offs[bn + 1]]
                    lists.append((l[(l >= a) & (l < b)] - a).astype(np.int32))
                po = np.zeros(self.plan.n_bin + 1, np.int64)
                po[1:] = np.cumsum([len(x) for x in lists])
                ent = np.concatenate(lists) if po[-1] else np.zeros(0, np.int32)
                out.append((a, b, torch.from_numpy(ent).to(self.device), po))
            self._entries_cache[key] = out
        return self._entries_cache[key]

    def _sketch_call(self, p: int, rows, ent, offs, fill, osk, onm, oskd=None, onmd=None, accumulate=False, ocr=None, ocrd=None):
        """One engine call of pass ``p`` (additive, or additive + dominance for RHE_DOM) into the given outputs (``ocr`` /
        ``ocrd``: the Gram matrices of the pass's cross columns, when it has any)."""
        kw = {} if ocr is None else {"out_cross": ocr}
        if self.dom:
            if ocrd is not None:
                kw["out_cross_dom"] = ocrd
            self.engs[p].sketch_block_dom(rows, ent, offs, fill=fill, out_sketch=osk, out_norm=onm, out_sketch_dom=oskd,
                                          out_norm_dom=onmd, accumulate=accumulate, **kw)
        else:
            self.engs[p].sketch_block(rows, ent, offs, fill=fill, out_sketch=osk, out_norm=onm, accumulate=accumulate, **kw)

    def _batch_entries(self, js):
        """Entry lists of the blocks ``js`` (consecutive) as ONE call: row indices relative to the first row of the batch,
        offsets over the (block, bin) lists in block-major order."""
        key = (js[0], js[-1])
        if key not in self._entries_cache:
            first = self.plan.range(js[0])[0]
            ents, offs, pos = [], [0], 0
            for j in js:
                cat, o = self.plan.entries(j)
                ents.append(cat + np.int32(self.plan.range(j)[0] - first))
                offs.extend((pos + o[1:]).tolist())
                pos += int(o[-1])
            self._entries_cache[key] = (torch.from_numpy(np.concatenate(ents).astype(np.int32)).to(self.device), np.asarray(offs, np.int64))
        return self._entries_cache[key]

    def _block_fills(self, pass_index: int):
        """Fill vectors (device float32, or None for a block without a missing call) of every local block for the draws of
        ``pass_index`` (see _uniform_stream).  They depend on the seed, the keep mask and the genotypes only -- not on the
        phenotype, not on how often a pass runs -- so the replay of NumPy's stream (it starts by re-drawing the N x k probe
        matrix: 50 ms at N = 500k) is paid once per runner, not once per pass."""
        if not hasattr(self, "_fills"):
            self._fills = {}
        if pass_index not in self._fills:
            draw = self._uniform_stream(pass_index)
            out = []
            for j in range(self.j0, self.j1):
                sj, ej = self.plan.range(j)
                out.append(self._fill_for_block(j, draw(j, ej - sj)))      # one uniform per SNP, missing or not (base.py:266,285)
            self._fills[pass_index] = out
        return self._fills[pass_index]

    def sketch_local_blocks(self, out_sk, out_norm, pass_index: int = 0, after_block=None, after_call=None):
        """Enqueue the sketch of every local block: ``out_sk[p][jl]`` / ``out_norm[p][jl]`` receive the per-block
        outputs of engine pass p (``[nbin, n_sketch_p, n_pad]`` float32 / ``[nbin, n_norm_p]`` float64).  With a
        one-slot store (``recompute_pass2``) every block lands in slot 0 and ``after_block(jl_index, j)`` consumes it;
        ``after_call(jl0, n)`` runs behind every engine call of the stored form (local blocks jl0 .. jl0 + n - 1 are enqueued)."""
        # Host-resident rows (engine.HostRows) whose statistics are not known yet: no pass over the link of their own -- every
        # call's window gets its statistics (one HBM-speed kernel) and, from them, the fill values of its blocks right before
        # its sketch; the imputation stream is consumed in block order either way (_lazy_window_stats)
        lazy = bool(self._check_missing and self._stats is None and hasattr(self.geno, "window") and self.geno.shape[0] > 0)
        if lazy:
            self._lazy_begin(pass_index)
            any_missing, block_fills = True, self._fills[pass_index]
        else:
            any_missing = bool(self._check_missing and self._ingest_stats()[1].any())
            block_fills = self._block_fills(pass_index) if any_missing else None
        one_slot = after_block is not None
        pipelined = [] if one_slot else [eng for eng in self.engs if hasattr(eng, "set_async")]
        for eng in pipelined:
            eng.set_async(True)       # blocks write distinct outputs: let GEMM 1 of block j+1 overlap GEMM 2 of block j
        fills = []                    # keep fill tensors alive until the join
        B = 1 if one_slot else self._blocks_per_call()
        nb = self.plan.n_bin
        try:
            local = list(range(self.j0, self.j1))
            for jl0 in range(0, len(local), B):
                js = local[jl0:jl0 + B]
                jl = 0 if one_slot else jl0
                if len(js) == 1:
                    ent, offs = self._entries(js[0])
                else:
                    ent, offs = self._batch_entries(js)
                s = self.plan.range(js[0])[0]
                e = self.plan.range(js[-1])[1]
                rows = self._rows(s - self.row0, e - self.row0)
                if lazy:
                    self._lazy_window_stats(pass_index, js, rows)
                fill = None
                if any_missing:
                    parts = []
                    for j in js:
                        sj, ej = self.plan.range(j)
                        parts.append((block_fills[j - self.j0], ej - sj))
                    if any(f is not None for f, _ in parts):      # blocks without a missing call need no fill path
                        fill = torch.cat([f if f is not None else torch.zeros(m_, dtype=torch.float32, device=self.device)
                                          for f, m_ in parts]) if len(parts) > 1 else parts[0][0]
                        fills.append(fill)
                # Blocks whose lists are longer than a kernel chunk takes (or whose rows pass the 32-bit span): their row pieces go
                # into the SAME call as list windows of their own, the windows' outputs are added per block afterwards -- one
                # GEMM 1 launch over all rows of the call instead of one per 15 104 entries (N = 500k, 20 000-SNP blocks: GEMM 1
                # 25.9 -> 24.4 ms per pass, the mid kernels 2.0 -> 0.4).  Blocks this long are what M of several millions gives at jn = 100.
                win = None
                if self._window_calls_ok() and any(self._pieces(j) is not None for j in js):
                    win = self._windows(js)
                pieces = self._pieces(js[0]) if (len(js) == 1 and win is None) else None
                for p in range(self.n_pass):
                    has_cr = self.pass_cross[p] > 0
                    ocr = self.blocks_cr[p][jl] if has_cr else None
                    ocrd = self.blocks_crd[p][jl] if (has_cr and self.dom) else None
                    if win is not None:
                        ent_w, offs_w, wb = win
                        n_w = len(wb)
                        sl = slice(jl, jl + len(js))
                        # targets: additive sketch / norm / cross Gram, then RHE_DOM's dominance counterparts (None: not in this run)
                        tgts = [out_sk[p][sl], out_norm[p][sl], self.blocks_cr[p][sl] if has_cr else None,
                                self.blocks_skd[p][sl] if self.dom else None, self.blocks_nmd[p][sl] if self.dom else None,
                                self.blocks_crd[p][sl] if (has_cr and self.dom) else None]
                        tmps = [None if t is None else torch.empty((n_w * nb,) + tuple(t.shape[2:]), dtype=t.dtype, device=self.device)
                                for t in tgts]
                        self._sketch_call(p, rows, ent_w, offs_w, fill, tmps[0], tmps[1], tmps[3], tmps[4], ocr=tmps[2], ocrd=tmps[5])
                        if hasattr(self.engs[p], "join"):
                            self.engs[p].join()       # (a no-op unless the engine runs its private two-stream pipeline)
                        w0 = 0
                        for bi in range(len(js)):
                            w1 = w0
                            while w1 < n_w and wb[w1] == bi:
                                w1 += 1
                            for tmp, tgt in zip(tmps, tgts):
                                if tmp is not None:
                                    torch.sum(tmp[w0 * nb:w1 * nb].view(w1 - w0, nb, *tgt.shape[2:]), dim=0, out=tgt[bi])
                            w0 = w1
                        del tmps
                    elif pieces is not None:        # a block beyond the 32-bit row offsets of one call: row sub-ranges
                        for i_, (a, b, ent_p, offs_p) in enumerate(pieces):
                            self._sketch_call(p, rows[a:b], ent_p, offs_p, None if fill is None else fill[a:b].contiguous(),
                                              out_sk[p][jl], out_norm[p][jl], self.blocks_skd[p][jl] if self.dom else None,
                                              self.blocks_nmd[p][jl] if self.dom else None, accumulate=i_ > 0, ocr=ocr, ocrd=ocrd)
                    else:
                        n = len(js)
                        flat = lambda t: t[jl:jl + n].reshape(n * nb, *t.shape[2:])      # [blocks, bins, ...] -> the call's lists
                        self._sketch_call(p, rows, ent, offs, fill, flat(out_sk[p]), flat(out_norm[p]),
                                          flat(self.blocks_skd[p]) if self.dom else None, flat(self.blocks_nmd[p]) if self.dom else None,
                                          ocr=flat(self.blocks_cr[p]) if has_cr else None,
                                          ocrd=flat(self.blocks_crd[p]) if (has_cr and self.dom) else None)
                if one_slot:
                    after_block(jl0, js[0])
                elif after_call is not None:
                    for eng in pipelined:
                        eng.join()         # (a no-op unless the engine runs its private two-stream pipeline)
                    after_call(jl0, len(js))
            if lazy:
                self._lazy_end()
        finally:
            for eng in pipelined:
                eng.join()
                eng.set_async(False)

    def _apply_U(self, v: torch.Tensor, f32: bool) -> torch.Tensor:
        """``W (Q (W^T v))`` for rows of v ``[..., n_pad]``; float32 like _compute_UXXz (base.py:407-408) or float64
        like setup_lhs_rhs_jackknife (base.py:584-586).  Result float64."""
        proj = getattr(self.eng, "apply_projector", None)
        n_rows = v.numel() // max(v.shape[-1], 1)
        # the HIP kernels (pyrhe_engine_apply_projector: <= 64 covariates, < 65 536 rows per call; wider covariate sets take the
        # torch float32 chain below); stand-in engines of the CPU tests lack them
        if f32 and proj is not None and self.WT32.shape[0] <= 64 and n_rows <= 65535:
            return proj(v.to(torch.float64).contiguous(), self.WT32, self.Q32)
        W, Q = (self.W32, self.Q32) if f32 else (self.W64, self.Q64)
        x = v.to(W.dtype)
        shp = x.shape
        x = x.reshape(-1, shp[-1])
        out = ((x @ W) @ Q.T) @ W.T
        return out.reshape(shp).to(torch.float64)

    def max_traits(self) -> int:
        """Phenotypes one pass can take: the sketches X X^T Z do not depend on the phenotype, a phenotype only adds a norm
        column (two with G x E), so the spare right-hand columns of the first engine pass carry further phenotypes for free
        (k = 10: 20 without covariates, 10 with; the reference re-runs everything per phenotype)."""
        per = 2 if self.n_gxe else 1
        if self.cov is not None and self.cross:
            # the covariate (cross) columns share the passes with the phenotypes: as many phenotypes as leave the plan within
            # the engines at hand -- a few fit beside the cross columns in pass 0; more push the cross columns into a later pass
            # where there is one (G x E + covariates: a pass of GEMM 1 alone, far cheaper than a second trait group)
            c = self.cov.shape[1]
            used = sum(nc for _, _, nc in plan_passes(self.k, per, per)[0])
            best = 1
            for nt in range(1, (MAX_ENGINE_COLS - used) // per + 1):
                if len(plan_passes(self.k, per, per * nt, cross=[c] * per, cross_digits=self.cross_digits, wide=self.wide_cols,
                                   cross_rows=self._cross_rows())[0]) <= len(self.engs):
                    best = nt
            return best
        n_groups = (2 if self.cov is not None else 1) * (2 if self.n_gxe else 1)
        used = sum(nc for _, _, nc in plan_passes(self.k, n_groups, per)[0])
        return max(1, (MAX_ENGINE_COLS - used) // per)

    def run_trait(self, y: np.ndarray, check_missing: bool = True) -> Systems:
        """Pass 1 + pass 2 for one centred phenotype ``y [N]`` -- or several, ``y [N, n_traits <= max_traits()]``, which
        share both passes; returns the jn+1 normal-equation systems (every rank), ``q_all`` holding every phenotype's."""
        self.begin_trait(y, check_missing)
        self.pass1()
        return self.pass2()

    def begin_trait(self, y: np.ndarray, check_missing: bool = True):
        """Install the right-hand columns and allocate the per-block sketch store."""
        nb, jl = self.plan.n_bin, self.j1 - self.j0
        self._check_missing = check_missing
        self._install_rhs(y)
        self._span_ok = None          # decided per trait group: the passes (engines, columns) may differ from the last one's
        dev = self.device
        # The per-block sketches stay in HBM for pass 2 when they fit (C3: 2 GB, C4: 32 GB of the 288); otherwise pass 2
        # re-sketches every block like the reference's second pass and only one block's outputs are ever resident.  The
        # decision is per trait group: a store of the right shape left by the previous group is reused as it is, any other
        # one is released (and torch's cached blocks returned) BEFORE the free memory is read.
        shapes = [(ns, nn) for ns, nn in self.pass_shapes]
        full = [(jl, nb, ns, self.n_pad) for ns, _ in shapes]
        old = getattr(self, "blocks_sk", None)
        if old is not None and not self.one_slot and [tuple(b.shape) for b in old] == full and \
                [b.shape[2] for b in self.blocks_nm] == [nn for _, nn in shapes] and \
                [b.shape[2] for b in self.blocks_cr] == list(self.pass_cross):
            for b in self.blocks_nm:
                b.zero_()
            if self.dom:
                for b in self.blocks_nmd:
                    b.zero_()
            return
        self.blocks_sk = self.blocks_nm = self.blocks_skd = self.blocks_nmd = self.blocks_cr = self.blocks_crd = None
        self.one_slot = bool(self.recompute_pass2)
        store_bytes = jl * nb * sum(ns for ns, _ in shapes) * self.n_pad * 4 * (2 if self.dom else 1)
        # beside the store: the float64 totals, and with several ranks the exchange's accumulator + two rounds in flight
        # (_pass1_exchange_rounds), each once more for the (jn - j)-weighted totals of quirk Q4
        tot_bytes = nb * sum(ns for ns, _ in shapes) * self.n_pad * 8 * (2 if self.dom else 1)
        q4 = self.cov is not None and self.streaming and self.compat == "reference"
        store_bytes += tot_bytes * (3 if self.collective else 1) * (2 if q4 else 1)
        if not self.one_slot and dev.type == "cuda":
            torch.cuda.empty_cache()
            free = torch.cuda.mem_get_info(dev)[0]
            if store_bytes > 0.8 * free:
                self.one_slot = True
                self.timings["recompute_pass2"] = f"auto: {store_bytes / 1e9:.1f} GB of block sketches vs {free / 1e9:.1f} GB free"
        if self.one_slot:
            jl = min(jl, 1)
        self.blocks_sk = [torch.empty((jl, nb, ns, self.n_pad), dtype=torch.float32, device=dev) for ns, _ in shapes]
        self.blocks_nm = [torch.zeros((max(jl, 1), nb, nn), dtype=torch.float64, device=dev) for _, nn in shapes]
        self.blocks_skd = [torch.empty_like(b) for b in self.blocks_sk] if self.dom else None
        self.blocks_nmd = [torch.zeros_like(b) for b in self.blocks_nm] if self.dom else None
        # Gram matrices of the cross columns (the covariates): [block][bin][c][c] float64 -- all that is kept of XXUz
        self.blocks_cr = [torch.zeros((max(jl, 1), nb, nx, nx), dtype=torch.float64, device=dev) for nx in self.pass_cross]
        self.blocks_crd = [torch.zeros_like(b) for b in self.blocks_cr] if self.dom else None

    # ---- estimate-level views of one block's (or the total's) engine outputs ----------------------
    def _group(self, sk_list, name):
        """All k columns of a group ``[..., k, n_pad]`` from per-pass engine outputs (a view when the group is whole)."""
        parts = [sk_list[p_][:, c0:c0 + nc] for p_, c0, _, nc in self.loc[name]]
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)

    def _cross_grams(self, cr_list, crd_list=None):
        """S_a = W^T X_a X_a^T W ``[..., ne_g, c, c]`` of the genotype-derived estimates (G bins, then dominance or G x E
        bins) from the per-pass Gram matrices of the cross columns ``[..., nbin, n_cross_p, n_cross_p]``."""
        def grp(lst, name):
            p_, idx = self.cross_loc[name]
            return lst[p_][..., idx[:, None], idx[None, :]]
        S = grp(cr_list, "W")
        if self.dom:
            S = torch.cat([S, grp(crd_list, "W")], dim=-3)
        if self.n_gxe:
            S = torch.cat([S, grp(cr_list, "eW")], dim=-3)
        return S

    def _pc_of(self, S):
        """W^T XXUz_a[s] = S_a (Q W^T z_s) for every probe vector: ``[..., ne_g, c, c]`` -> ``[..., ne_g, k, c]``."""
        return torch.einsum("...amn,ns->...asm", S, self.G64)

    def _block_estimates(self, sk_list, nm_list, skd_list=None, nmd_list=None, cr_list=None, crd_list=None):
        """A = XXz, C = XXUz ``[ne_g, k, n_pad]`` and r = yXXy ``[ne_g, n_traits]`` of the genotype-derived estimates (G bins,
        then dominance bins or G x E bins) from engine outputs of one block; float64.  With the covariates as cross columns
        (``self.cross``) C is the projection ``W^T XXUz`` ``[ne_g, k, c]`` instead -- all the normal equations use of it."""
        f64 = torch.float64
        A = self._group(sk_list, "Z").to(f64)
        if self.cov is not None and self.cross:
            C = self._pc_of(self._cross_grams(cr_list, crd_list))
            py, cy = self.loc["y"]
            nt = self.n_traits
            r = nm_list[py][:, cy:cy + nt]
            if self.dom:
                A = torch.cat([A, self._group(skd_list, "Z").to(f64)], dim=0)
                r = torch.cat([r, nmd_list[py][:, cy:cy + nt]], dim=0)
            if self.n_gxe:
                gz = self._group(sk_list, "eZ")
                A = torch.cat([A, (gz * self.env_pad.to(gz.dtype)).to(f64)], dim=0)
                pe, ce = self.loc["ey"]
                r = torch.cat([r, nm_list[pe][:, ce:ce + nt]], dim=0)
            return A, C, r
        C = self._group(sk_list, "UZ").to(f64) if self.cov is not None else None
        py, cy = self.loc["y"]
        nt = self.n_traits
        r = nm_list[py][:, cy:cy + nt]                                   # [nb, n_traits]
        if self.dom:
            A = torch.cat([A, self._group(skd_list, "Z").to(f64)], dim=0)
            if C is not None:
                C = torch.cat([C, self._group(skd_list, "UZ").to(f64)], dim=0)
            r = torch.cat([r, nmd_list[py][:, cy:cy + nt]], dim=0)
        if self.n_gxe:
            e = self.env_pad
            # elem_mul(X, env) then float32 GEMVs in the reference (streaming_genie.py:39-41): X_e X_e^T z = e.(X X^T (e.z))
            gz = self._group(sk_list, "eZ")
            Ae = (gz * e.to(gz.dtype)).to(f64)
            A = torch.cat([A, Ae], dim=0)
            if C is not None:
                gu = self._group(sk_list, "eUZ")
                Ce = (gu * e.to(gu.dtype)).to(f64)
                C = torch.cat([C, Ce], dim=0)
            pe, ce = self.loc["ey"]
            r = torch.cat([r, nm_list[pe][:, ce:ce + nt]], dim=0)
        return A, C, r

    def _nxe_sketch(self):
        """Closed form of the N x E sketch the reference builds from an N x N diagonal (base_streaming.py:45-58,
        SURVEY Q10): XXz = fp32(e . fp32(e . z)), yXXy = ||e . y~||^2, XXUz likewise with UZ."""
        f32, f64 = torch.float32, torch.float64
        e32 = self.env_pad.to(f32)
        A = (e32 * (e32 * self.Z64.to(f32))).to(f64)[None]                 # [1, k, n_pad]
        ypad = self._pad_cols(torch.from_numpy(np.ascontiguousarray(self.y_res.T)))      # [n_traits, n_pad]
        v = (e32 * ypad.to(f32)).to(f64)
        r = (v * v).sum(dim=1).to(f32).to(f64).reshape(1, -1)                             # [1, n_traits]
        C = (e32 * (e32 * self.UZ64.to(f32))).to(f64)[None] if self.cov is not None else None
        return A, C, r

    # Exchange rounds of one sketch pass at most (several ranks): a rank's engine calls are grouped into that many all-reduces.
    MAX_EXCHANGE_ROUNDS = 4

    def _exchange_plan(self):
        """(engine calls per exchange round, rounds) -- the same on every rank: the largest shard's calls, grouped so that a pass
        has at most MAX_EXCHANGE_ROUNDS rounds (wire bytes and buffers must not grow with the calls per rank: 2 GPUs with 32 bins
        at N = 500k make 25 calls of 2 blocks each)."""
        largest = max(b_ - a_ for a_, b_ in shard_ranges(self.plan.n_jack, self.world))
        n_calls = -(-largest // max(1, self._blocks_per_call()))
        stride = max(1, -(-n_calls // self.MAX_EXCHANGE_ROUNDS))
        return stride, -(-n_calls // stride)

    def _pass1_exchange_rounds(self, q4, w, z_part):
        """Several ranks: sketch the local blocks and all-reduce float64 partial totals while the sketch runs.  The engine calls
        of a rank (a call = a batch of blocks) are grouped into at most MAX_EXCHANGE_ROUNDS rounds (_exchange_plan); behind the
        last call of a round the partial totals of the round's blocks are summed (pyrhe_sum_blocks over their slice of the
        store) and all-reduced asynchronously -- RCCL moves them over xGMI on its own stream while the next calls' kernels run,
        only the last round's exchange is exposed (8 GPUs, 8 bins + covariates: two calls of 8 + 5 blocks per rank = two rounds
        of 320 MB).  At most TWO rounds are in flight: before a third is issued the oldest is waited for and folded into ONE
        accumulator per bucket, so the memory of the exchange is three times the totals whatever the number of calls
        (advisor, round 3: it used to keep every round's buffers until the end).  The whole-data totals are the sums of the
        reduced partials in round order (float64: any grouping agrees to 1e-16).  Every rank takes part in the same rounds in
        the same order; a rank with fewer calls (or no block) contributes zeros.
        Returns (tot_sk, tot_skd, w_sk, w_skd): per-pass totals, their dominance twins, the (jn - j)-weighted XXz of quirk Q4."""
        dev, f64 = self.device, torch.float64
        names = ("Z", "eZ") if self.n_gxe else ("Z",)
        stride, n_rounds = self._exchange_plan()
        zero = lambda b: torch.zeros(b.shape[1:], dtype=f64, device=dev)
        acc, inflight = {}, []
        st = {"calls": 0, "start": 0, "end": 0, "rounds": 0}

        def fold_oldest():
            hs, bufs = inflight.pop(0)
            for h in hs:
                h.wait()
            for key, t in bufs:
                if key in acc:
                    acc[key] += t
                else:
                    acc[key] = t

        def issue(jl0, jl1):
            """One round: the partial totals of local blocks jl0 .. jl1 - 1 (none: zeros)."""
            while len(inflight) >= 2:
                fold_oldest()
            n = jl1 - jl0
            hs, bufs = [], []

            def send(key, t):
                h = self._allreduce(t, async_op=True)
                if h is not None:
                    hs.append(h)
                bufs.append((key, t))
            wj = w[jl0:jl1].contiguous() if (q4 and n) else None
            weighted = {"w": [None] * len(self.blocks_sk), "wd": [None] * len(self.blocks_sk)}
            for p_, b in enumerate(self.blocks_sk):
                if not b.shape[2]:
                    continue                                   # a pass without sketch columns (cross columns alone)
                for key, wkey, store in (("sk", "w", self.blocks_sk), ("skd", "wd", self.blocks_skd if self.dom else None)):
                    if store is None:
                        continue
                    bs = store[p_]
                    if not n:
                        plain, wt_ = zero(bs), (zero(bs) if q4 else None)
                    elif q4:                                   # plain and (jn - j)-weighted partial totals from one read of the slice
                        plain, wt_ = self.eng.sum_blocks_both(bs[jl0:jl1], wj)
                    else:
                        plain, wt_ = self.eng.sum_blocks(bs[jl0:jl1]), None
                    send((key, p_), plain)
                    weighted[wkey][p_] = wt_
            if q4:
                for key in ("w", "wd") if self.dom else ("w",):
                    for n_, chunks in z_part(weighted[key]).items():
                        for i_, t in enumerate(chunks):
                            send((key, n_, i_), t)
            inflight.append((hs, bufs))
            st["rounds"] += 1
            self.timings["exchange_rounds"] = st["rounds"]

        def after_call(jl0, n):
            st["calls"] += 1
            st["end"] = jl0 + n
            if st["calls"] % stride == 0:
                issue(st["start"], st["end"])
                st["start"] = st["end"]
        self.sketch_local_blocks(self.blocks_sk, self.blocks_nm, pass_index=0, after_call=after_call)
        if st["end"] > st["start"]:
            issue(st["start"], st["end"])
        if st["rounds"] > n_rounds:
            raise RuntimeError(f"{st['rounds']} exchange rounds on this rank, {n_rounds} planned: the ranks would not meet")
        while st["rounds"] < n_rounds:
            issue(0, 0)
        while inflight:
            fold_oldest()
        tot_sk = [acc.get(("sk", p_), None) if b.shape[2] else None for p_, b in enumerate(self.blocks_sk)]
        tot_sk = [t if t is not None else zero(b) for t, b in zip(tot_sk, self.blocks_sk)]
        tot_skd = None
        if self.dom:
            tot_skd = [acc[("skd", p_)] if ("skd", p_) in acc else zero(b) for p_, b in enumerate(self.blocks_skd)]
        w_sk = w_skd = None
        if q4:
            like = lambda n_, i_: self.blocks_sk[self.loc[n_][i_][0]][:, :, :self.loc[n_][i_][3]]
            get = lambda key, n_, i_: acc[(key, n_, i_)] if (key, n_, i_) in acc else zero(like(n_, i_))
            w_sk = {n_: [get("w", n_, i_) for i_ in range(len(self.loc[n_]))] for n_ in names}
            w_skd = {n_: [get("wd", n_, i_) for i_ in range(len(self.loc[n_]))] for n_ in names} if self.dom else None
        return tot_sk, tot_skd, w_sk, w_skd

    def pass1(self):
        """One streaming sketch pass over the local blocks (base_streaming.py:85-104) + aggregate
        (base_streaming.py:21-58 / base.py:465-500: float64 sums in block order) + the cross-GPU all-reduce.

        Multi-GPU (world > 1): the float64 partial totals of every engine call (a batch of blocks) are all-reduced as soon as
        the call is enqueued, in the engines' own layout ``[bin][sketch column][n_pad]`` (every estimate-level quantity below is
        linear in them) -- see _pass1_exchange_rounds; the small norm sums and cross Grams travel in one packed buffer.  Float64
        on the wire: the single-GPU path adds the blocks in float64 (base.py:468-471 widens every float32 block result before
        adding), and a float32 hand-over would round every rank's partial sum to 6e-8 -- the 1 / 2 / 4 / 8-GPU results would then
        agree to 1e-7 instead of 1e-15.  Bytes per rank and pass (ring: x 2 (G-1)/G on the links): k = 10, N = 500k: 1 bin 40 MB;
        8 bins + covariates (cross columns) 320 MB (+ 320 MB for the (jn - j)-weighted XXz of quirk Q4); G + GxE 80 MB."""
        plan, k = self.plan, self.k
        jn, jl, dev = plan.n_jack, self.j1 - self.j0, self.device
        f64 = torch.float64
        use_cov = self.cov is not None
        q4 = use_cov and self.streaming and self.compat == "reference"
        w = torch.arange(jn - self.j0, jn - self.j0 - jl, -1, dtype=f64, device=dev) if jl else torch.zeros(0, dtype=f64, device=dev)
        handles = []
        w_reduced = False             # the (jn - j)-weighted totals of quirk Q4 already went through the collective (per call)

        def reduce_async(t):
            if self.collective:
                h = self._allreduce(t, async_op=True)
                if h is not None:
                    handles.append(h)
            return t

        def z_part(lst):
            """The XXz columns (groups Z and e.Z) of weighted totals as contiguous tensors, chunk by chunk: all that quirk Q4
            needs of them (the U Z columns are not sent)."""
            return {name: [lst[p_][:, c0:c0 + nc].contiguous() for p_, c0, _, nc in self.loc[name]]
                    for name in (("Z", "eZ") if self.n_gxe else ("Z",))}

        tot_skd = tot_nmd = w_sk = w_skd = tot_crd = None
        if not self.one_slot and self.collective:
            tot_sk, tot_skd, w_sk, w_skd = self._pass1_exchange_rounds(q4, w, z_part)
            tot_nm = [b[:jl].sum(dim=0) for b in self.blocks_nm]
            tot_cr = [b[:jl].sum(dim=0) for b in self.blocks_cr]
            if self.dom:
                tot_nmd = [b[:jl].sum(dim=0) for b in self.blocks_nmd]
                tot_crd = [b[:jl].sum(dim=0) for b in self.blocks_crd]
            w_reduced = True
        elif not self.one_slot:
            self.sketch_local_blocks(self.blocks_sk, self.blocks_nm, pass_index=0)
            tot_cr = [b[:jl].sum(dim=0) for b in self.blocks_cr]
            if self.dom:
                tot_crd = [b[:jl].sum(dim=0) for b in self.blocks_crd]
            # Block sums as ONE float64 reduction over the stored per-block sketches per bucket (the reference adds the
            # blocks one by one in float64, base.py:468-471; any order agrees to 1e-16).  (Round 4 measured the alternative --
            # the sums of groups of calls on a side stream beside the next calls' contractions: 8 bins + covariates 55.0 ->
            # 57.4 ms per pass, headline 45.6 -> 46.4: a third stream of reads slows GEMM 1 by more than the sums take.)  U is linear, so the "fixed" UXXz
            # total is U applied to the total (float32 rounding order differs from per-block application at 1e-7).
            zeros_like_tot = lambda b: torch.zeros(b.shape[1:], dtype=f64, device=dev)
            sum_blocks = lambda b: self.eng.sum_blocks(b) if b.shape[2] else zeros_like_tot(b)     # (a pass without sketch columns)
            # quirk Q4 needs the (jn - j)-weighted totals beside the plain ones: both from ONE read of the store
            # (pyrhe_sum_blocks_both; 16 GB at 8 bins: 2.9 ms per read)
            both = lambda b: self.eng.sum_blocks_both(b, w) if b.shape[2] else (zeros_like_tot(b), zeros_like_tot(b))
            if q4:
                pairs = [both(b) for b in self.blocks_sk]
                tot_sk = [reduce_async(p_[0]) for p_ in pairs]
                w_sk = z_part([p_[1] for p_ in pairs])
            else:
                tot_sk = [reduce_async(sum_blocks(b)) for b in self.blocks_sk]
            tot_nm = [b[:jl].sum(dim=0) for b in self.blocks_nm]
            if self.dom:
                if q4:
                    pairs = [both(b) for b in self.blocks_skd]
                    tot_skd = [reduce_async(p_[0]) for p_ in pairs]
                    w_skd = z_part([p_[1] for p_ in pairs])
                else:
                    tot_skd = [reduce_async(sum_blocks(b)) for b in self.blocks_skd]
                tot_nmd = [b[:jl].sum(dim=0) for b in self.blocks_nmd]
        else:
            # one-slot store: every block is added to the float64 totals (and the (jn - j)-weighted ones of quirk Q4)
            # right after its sketch, in block order
            zeros = lambda lst: [torch.zeros(b.shape[1:], dtype=f64, device=dev) for b in lst]
            tot_sk, tot_nm, tot_cr = zeros(self.blocks_sk), zeros(self.blocks_nm), zeros(self.blocks_cr)
            tot_skd, tot_nmd, tot_crd = (zeros(self.blocks_skd), zeros(self.blocks_nmd), zeros(self.blocks_crd)) if self.dom else (None, None, None)
            wf = wfd = None
            if q4:
                wf = zeros(self.blocks_sk)
                wfd = zeros(self.blocks_skd) if self.dom else None

            def add(jl_i, j):
                for p_ in range(self.n_pass):
                    tot_sk[p_] += self.blocks_sk[p_][0]
                    tot_nm[p_] += self.blocks_nm[p_][0]
                    tot_cr[p_] += self.blocks_cr[p_][0]
                    if self.dom:
                        tot_skd[p_] += self.blocks_skd[p_][0]
                        tot_nmd[p_] += self.blocks_nmd[p_][0]
                        tot_crd[p_] += self.blocks_crd[p_][0]
                    if q4:
                        wf[p_].add_(self.blocks_sk[p_][0], alpha=float(jn - j))
                        if self.dom:
                            wfd[p_].add_(self.blocks_skd[p_][0], alpha=float(jn - j))
            self.sketch_local_blocks(self.blocks_sk, self.blocks_nm, pass_index=0, after_block=add)
            for t in tot_sk + (tot_skd or []):
                reduce_async(t)
            if q4:
                w_sk, w_skd = z_part(wf), (z_part(wfd) if self.dom else None)
        if self.collective:
            if not w_reduced:
                for d_ in (w_sk, w_skd):
                    for chunks in (d_ or {}).values():
                        for t in chunks:
                            reduce_async(t)
            small = tot_nm + (tot_nmd or []) + tot_cr + (tot_crd or [])     # [nbin, n_norm] / [nbin, c, c] each: one packed buffer
            flat = torch.cat([t.reshape(-1) for t in small])
            self._allreduce(flat)
            off = 0
            for t in small:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
            for h in handles:
                h.wait()
        tot_A, tot_C, tot_r = self._block_estimates(tot_sk, tot_nm, tot_skd, tot_nmd, tot_cr, tot_crd)
        tot_A, tot_r = tot_A.contiguous(), tot_r.clone()
        tot_C = tot_C.contiguous() if use_cov else None
        wsum = None
        if q4:
            # quirk Q4 (streaming_rhe.py:18-21): the reference applies U to the RUNNING sum of XXz after every block;
            # U is linear, so sum_j U(sum_{i<=j} XXz_i) = U(sum_i (jn - i) XXz_i): one weighted block sum, one U
            # (float32 rounding order differs from the block-by-block application at the 1e-7 level)
            cat = lambda chunks: torch.cat(chunks, dim=1) if len(chunks) > 1 else chunks[0]
            wsum = cat(w_sk["Z"])                                  # the estimates in _block_estimates' order: G bins,
            if self.dom:
                wsum = torch.cat([wsum, cat(w_skd["Z"])], dim=0)   # then dominance bins
            if self.n_gxe:
                wsum = torch.cat([wsum, cat(w_sk["eZ"]) * self.env_pad], dim=0)      # or G x E bins: e . (X X^T (e . z))
            wsum = wsum.contiguous()
        D_tot = self._apply_U(wsum if q4 else tot_A, f32=True) if use_cov else None     # a10: pyrhe_engine_apply_projector
        self.tot_A, self.tot_C, self.tot_r, self.D_tot = tot_A, tot_C, tot_r, D_tot
        self.tot_sk = tot_A            # [n_est_g, k, n_pad] float64 (kept under the old name for callers)

    def pass2(self) -> Systems:
        """Leave-one-out normal equations where each block lives (base_streaming.py:110-144), gathered to every rank."""
        from .estimate import single_threaded_blas
        with single_threaded_blas():
            return self._pass2()

    def _pass2(self) -> Systems:
        self._tot_dots = {}              # whole-data inner products, valid for this pass only (see _dots)
        plan, k, ne = self.plan, self.k, self.n_est
        jn, nb = plan.n_jack, plan.n_bin
        use_cov = self.cov is not None
        # M (rhe.py:16, genie.py:36-44,77-82): SNPs of every estimate outside block j; the N x E term counts 1
        M = np.zeros((jn + 1, ne), np.int64)

        def row(c):
            parts = [c] + ([c] if (self.n_gxe or self.dom) else []) + ([np.ones(1, np.int64)] if self.n_nxe else [])
            return np.concatenate(parts)
        last = row(plan.len_bin)
        M[jn] = last
        for j in range(jn):
            blk = row(plan.block_counts(j))
            if self.n_nxe:
                blk[-1] = 0
            M[j] = last - blk
        yy = self.yy                                               # [n_traits]

        tot_A, tot_C, tot_r, D_tot = self.tot_A, self.tot_C, self.tot_r, self.D_tot
        nxe = self._nxe_sketch() if (self.n_nxe or self.quirk_last_total) else None
        if self.quirk_last_total:
            # whole-data slot of the LAST estimate := N x E sketch; its UXXz / XXUz only for the last probe vector
            # (base.py:472-481: the covariate lines sit outside the loop over b)
            tot_A, tot_r = tot_A.clone(), tot_r.clone()
            tot_A[-1], tot_r[-1] = nxe[0][0], nxe[2][0]
            if use_cov:
                tot_C, D_tot = tot_C.clone(), D_tot.clone()
                tot_C[-1], D_tot[-1] = 0, 0
                tot_C[-1, k - 1] = self._c_form(nxe[1][0, k - 1])
                D_tot[-1, k - 1] = self._apply_U(nxe[0][0, k - 1], f32=True)

        def with_nxe(A, C, D, r):
            if not self.n_nxe:
                return A, C, D, r
            A = torch.cat([A, nxe[0]], dim=0)
            r = torch.cat([r, nxe[2]], dim=0)
            if use_cov:
                Dn = torch.zeros_like(nxe[0])
                Cn = torch.zeros((1,) + tuple(C.shape[1:]), dtype=C.dtype, device=C.device)
                Cn[0, k - 1] = self._c_form(nxe[1][0, k - 1])
                Dn[0, k - 1] = self._apply_U(nxe[0][0, k - 1], f32=True)
                C = torch.cat([C, Cn], dim=0)
                D = torch.cat([D, Dn], dim=0)
            return A, C, D, r

        if self.pass2_impl == "torch":
            if self.n_traits != 1:
                raise ValueError("the materialised cross-check path takes one phenotype")
            nxe1 = None if nxe is None else (nxe[0], nxe[1], nxe[2][:, 0])
            if nxe is not None:
                nxe = nxe1
            return self._pass2_torch(M, float(yy[0]), tot_A, tot_C, D_tot, tot_r[:, 0], nxe, with_nxe)
        # ---- on-device assembly: every inner product of a slot comes from pyrhe_engine_loo_dots, whole batches of
        # blocks per call; only [batch, ne, ne]-sized results reach the host
        tots = (tot_A, tot_C, D_tot, tot_r)
        local = []
        js = list(range(self.j0, self.j1))
        if jn == 1 and js == [0]:                               # base.py:654-655: the only slot is the whole-data one
            js = []
        if js and not self.one_slot:
            store = (self.blocks_sk, self.blocks_nm, self.blocks_skd, self.blocks_nmd, self.blocks_cr, self.blocks_crd)
            Ts, qs = self._loo_systems(tots, nxe, store, len(js), M[js], yy)
            local = [(j, Ts[i], qs[i]) for i, j in enumerate(js)]
        else:
            for j in js:
                store = tuple(None if lst is None else [t[None] for t in lst] for lst in self._resketch(j))
                Ts, qs = self._loo_systems(tots, nxe, store, 1, M[j:j + 1], yy)
                local.append((j, Ts[0], qs[0]))
        Ts, qs = self._loo_systems(tots, nxe, None, 1, M[jn:jn + 1], yy)
        T_tot, q_tot = Ts[0], qs[0]
        if jn == 1 and self.j0 == 0 and self.j1 == 1:
            local = [(0, T_tot, q_tot)]
        if self.collective:
            local = self._gather_systems(local, T_tot.shape, q_tot.shape)
        local.sort(key=lambda t: t[0])
        Ts = np.stack([t[1] for t in local] + [T_tot])
        qs = np.stack([t[2] for t in local] + [q_tot])           # [jn+1, n_traits, ne+1]
        q_all = np.ascontiguousarray(np.transpose(qs, (1, 0, 2)))
        return Systems(Ts, q_all[0], M, q_all)

    def _c_form(self, row: torch.Tensor) -> torch.Tensor:
        """An explicit XXUz row ``[n_pad]`` in the form the C arrays are kept in: itself, or its projection W^T row ``[c]``."""
        return self.WT64 @ row if self.cross else row

    def _pass2_torch(self, M, yy, tot_A, tot_C, D_tot, tot_r, nxe, with_nxe):
        """The same systems from materialised leave-one-out arrays and torch reductions: slow (degenerate GEMM shapes),
        kept as an independent cross-check of the on-device assembly (tests set ``pass2_impl = "torch"``)."""
        jn, use_cov = self.plan.n_jack, self.cov is not None
        local = []
        for jl_i, j in enumerate(range(self.j0, self.j1)):
            stored = self._resketch(j) if self.one_slot else self._stored_block(jl_i)
            slot = jn if (jn == 1 and j == 0) else j          # base.py:654-655
            if slot == jn:
                T, q = self._system(*with_nxe(tot_A, tot_C, D_tot, tot_r), M[jn], yy)
            else:
                A, C, r = self._block_estimates(*stored)
                r = r[:, 0]
                D = None
                if use_cov:
                    D = D_tot - self._apply_U(A, f32=True)      # streaming_rhe.py:35-36
                    C = tot_C - C
                T, q = self._system(*with_nxe(tot_A - A, C, D, tot_r - r), M[j], yy)
            local.append((j, T, q))
        T_tot, q_tot = self._system(*with_nxe(tot_A, tot_C, D_tot, tot_r), M[jn], yy)
        if self.collective:
            local = self._gather_systems(local, T_tot.shape, q_tot.shape)
        local.sort(key=lambda t: t[0])
        return Systems(np.stack([t[1] for t in local] + [T_tot]), np.stack([t[2] for t in local] + [q_tot]), M)

    # ---- on-device normal equations ---------------------------------------------
    def _loo_rows(self, tot, store, group, egroup, nxe_row, s0, nc):
        """Row descriptors, restricted to probe columns ``s0 .. s0+nc-1``, of the ne estimates of one accumulator family
        (XXz: group "Z"/"eZ", XXUz: "UZ"/"eUZ"): whole-data slot minus the stored block sketch (``store`` None: the
        whole-data slot alone)."""
        from .engine import Row
        nb = self.plan.n_bin
        sk, skd = (store[0], store[2]) if store is not None else (None, None)

        def view(lst, name, b):
            p_, c0 = self._loc_at(name, s0)
            t = lst[p_]                                            # [batch, nb, n_sketch, n_pad] float32
            return t[0, b, c0:c0 + nc], t.stride(0)
        rows = []
        for a in range(tot.shape[0]):
            blk, stride, mul = None, 0, None
            if sk is not None:
                if a < nb:
                    blk, stride = view(sk, group, a)
                elif self.dom:
                    blk, stride = view(skd, group, a - nb)
                else:
                    blk, stride = view(sk, egroup, a - nb)
                    mul = self.env32
            rows.append(Row(tot[a, s0:s0 + nc], blk, mul, stride))
        if nxe_row is not None:
            rows.append(Row(nxe_row[s0:s0 + nc]))
        return rows

    def _sub_rows(self, rows, nc):
        """The nc sub-rows (one probe vector each, length n_pad) of every row, for projections onto the covariates."""
        from .engine import Row
        out = []
        for r in rows:
            for s_ in range(nc):
                out.append(None if r is None else
                           Row(None if r.tot is None else r.tot[s_], None if r.blk is None else r.blk[s_], r.mul, r.stride))
        return out

    def _lin_dots(self, x, y, n_sub, batch):
        """<x_a, y_b> for leave-one-out rows x against FIXED rows y, split by linearity into the whole-data part (read
        once, not once per block) and the block part (float32 rows only): [batch, len(x), len(y)], and the block part
        alone (the inner products of the stored block sketches, sign restored)."""
        from .engine import Row
        tot_rows = [None if r.tot is None else Row(r.tot) for r in x]
        blk_rows = [None if r.blk is None else Row(None, r.blk, r.mul, r.stride) for r in x]
        t = self._dots(tot_rows, y, n_sub, 1)
        b = self._dots(blk_rows, y, n_sub, batch)                  # rows evaluate to -blk
        return t + b, -b

    def _multi_products(self, multi, live, y_sub, y_flat, want_self, nc, batch):
        """``pyrhe_engine_loo_dots_multi`` beyond its 16 x 16 shapes.  More than 16 sub-strided fixed rows (GENIE with 8 bins: 16
        estimates + Z): further calls for the rest.  More than 16 block rows (over 16 estimates: 32 bins, RHE_DOM with 12): the
        rows go in OCTETS and every pair of octets shares one call, whose 16 x 16 self products then hold that pair's cross block
        -- all of ``<blk_a, blk_b>`` from the one-read kernel; the calls (0, 1), (2, 3), ... also deliver the products with the
        fixed rows.  (32 bins + covariates at N = 500k: pass 2 494 -> see DESIGN 3.4c.)"""
        f64 = dict(dtype=torch.float64, device=self.device)
        L, P = len(live), self.n_pad

        def with_fixed(rows):
            """sub / flat / self products of <= 16 block rows against all fixed rows"""
            o_sub, o_flat, o_self = multi(rows, y_sub[:16], y_flat, want_self, P, nc, batch)
            for c0 in range(16, len(y_sub), 16):
                o_sub = torch.cat([o_sub, multi(rows, y_sub[c0:c0 + 16], [], False, P, nc, batch)[0]], dim=2)
            return o_sub, o_flat, o_self

        if L <= 16:
            return with_fixed(live)
        octs = [list(range(i, min(i + 8, L))) for i in range(0, L, 8)]
        G = len(octs)
        o_sub = torch.zeros((batch, L, len(y_sub)), **f64)
        o_flat = torch.zeros((batch, L, nc, len(y_flat)), **f64) if y_flat else None
        o_self = torch.zeros((batch, L, L), **f64) if want_self else None
        first = [(i, i + 1) if i + 1 < G else (i - 1, i) for i in range(0, G, 2)]     # every octet once (the last may repeat one)
        rest = [(i, j) for i in range(G) for j in range(i + 1, G) if (i, j) not in first] if want_self else []
        done = set()
        for pi, (i, j) in enumerate(first + rest):
            rix = octs[i] + octs[j]
            rows = [live[t] for t in rix]
            ix = torch.as_tensor(rix, device=self.device)
            if pi < len(first):
                a_sub, a_flat, a_self = with_fixed(rows)
                new = [t_ for t_, r_ in enumerate(rix) if r_ not in done]
                sel = ix[new]
                o_sub[:, sel] = a_sub[:, new]
                if o_flat is not None:
                    o_flat[:, sel] = a_flat[:, new]
                done.update(rix)
            else:
                a_self = multi(rows, y_sub[:1], [], True, P, nc, batch)[2]
            if want_self:
                o_self[:, ix[:, None], ix[None, :]] = a_self
        return o_sub, o_flat, o_self

    def _blk_products(self, rows_blk, y_sub, y_flat, want_self, nc, batch):
        """All pass-2 products of the block-only rows ``rows_blk`` (None = an exact zero row; rows evaluate to ``-blk``):
        ``b_sub [batch, n, len(y_sub)]`` against the sub-strided fixed rows, ``b_flat [batch, n, nc, len(y_flat)]`` against
        the flat fixed rows per sub-row (None without flat rows), ``b_self [batch, n, n]`` against themselves (None unless
        asked).  Through ``pyrhe_engine_loo_dots_multi`` -- ONE pass over the store up to 16 x 16 rows, a few for more
        (``_multi_products``) -- wherever the rows have its form, otherwise the three ``loo_dots`` calls it stands for."""
        n = len(rows_blk)
        idx = [i for i, r in enumerate(rows_blk) if r is not None]
        f64 = dict(dtype=torch.float64, device=self.device)
        if not idx:
            return (torch.zeros((batch, n, len(y_sub)), **f64),
                    torch.zeros((batch, n, nc, len(y_flat)), **f64) if y_flat else None,
                    torch.zeros((batch, n, n), **f64) if want_self else None)
        live = [rows_blk[i] for i in idx]
        multi = getattr(self.eng, "loo_dots_multi", None)
        flat_ok = len(y_flat) <= 16
        if multi is not None and flat_ok and self.eng.multi_ok(live[:16], y_sub[:16], y_flat, self.n_pad) and \
                all(self.eng.multi_ok([r], y_sub[:1], [], self.n_pad) for r in live[16:]) and \
                all(self.eng.multi_ok(live[:1], [r], [], self.n_pad) for r in y_sub[16:]):
            o_sub, o_flat, o_self = self._multi_products(multi, live, y_sub, y_flat, want_self, nc, batch)
            if len(idx) == n:
                return o_sub, o_flat, o_self
            ix = torch.as_tensor(idx, device=self.device)
            b_sub = torch.zeros((batch, n, len(y_sub)), **f64)
            b_sub[:, ix] = o_sub
            b_flat = b_self = None
            if o_flat is not None:
                b_flat = torch.zeros((batch, n, nc, len(y_flat)), **f64)
                b_flat[:, ix] = o_flat
            if o_self is not None:
                b_self = torch.zeros((batch, n, n), **f64)
                b_self[:, ix[:, None], ix[None, :]] = o_self
            return b_sub, b_flat, b_self
        b_sub = self._dots(rows_blk, y_sub, nc, batch)
        b_flat = None
        if y_flat:
            b_flat = self._dots(self._sub_rows([r if r is not None else None for r in rows_blk], nc), y_flat, 1, batch)
            b_flat = b_flat.reshape(batch, n, nc, len(y_flat))
        b_self = self._dots(rows_blk, rows_blk, nc, batch) if want_self else None
        return b_sub, b_flat, b_self

    def _to_host(self, t: torch.Tensor) -> np.ndarray:
        """Device tensor -> NumPy through a cached pinned buffer (asynchronous copy, then an event poll)."""
        if t.device.type != "cuda":
            return t.numpy()
        key = (tuple(t.shape), t.dtype)
        if not hasattr(self, "_pinned"):
            self._pinned = {}
        buf = self._pinned.get(key)
        if buf is None:
            buf = self._pinned[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        while not ev.query():
            pass
        return buf.numpy().copy()

    def _dots(self, x, y, n_sub, batch):
        """loo_dots with None rows (exact zeros) filtered out; float64 DEVICE tensor [batch, len(x), len(y)] -- nothing
        here waits for the GPU."""
        xi = [i for i, r in enumerate(x) if r is not None]
        yi = [i for i, r in enumerate(y) if r is not None]
        if not xi or not yi:
            return torch.zeros((batch, len(x), len(y)), dtype=torch.float64, device=self.device)
        # products of whole-data rows only do not depend on the block: computed once per pass 2 (the batch of
        # leave-one-out slots, the whole-data slot and -- when blocks are re-sketched -- every block ask for the same)
        key = None
        if batch == 1 and all(r is None or r.blk is None for r in x) and all(r is None or r.blk is None for r in y):
            key = (tuple(None if r is None else r.tot.data_ptr() for r in x),
                   tuple(None if r is None else r.tot.data_ptr() for r in y), n_sub)
            hit = self._tot_dots.get(key)
            if hit is not None:
                return hit
            out = self._dots_uncached(x, y, xi, yi, n_sub, batch)
            self._tot_dots[key] = out
            return out
        return self._dots_uncached(x, y, xi, yi, n_sub, batch)

    def _dots_uncached(self, x, y, xi, yi, n_sub, batch):
        res = self.eng.loo_dots([x[i] for i in xi], [y[i] for i in yi], self.n_pad, n_sub, batch)
        if len(xi) == len(x) and len(yi) == len(y):
            return res
        out = torch.zeros((batch, len(x), len(y)), dtype=torch.float64, device=self.device)
        if not hasattr(self, "_idx_cache"):
            self._idx_cache = {}
        key = (tuple(xi), tuple(yi))
        if key not in self._idx_cache:      # uploaded once per pattern: no host->device copy on the steady path
            self._idx_cache[key] = (torch.as_tensor(xi, device=self.device)[:, None], torch.as_tensor(yi, device=self.device)[None, :])
        ix, iy = self._idx_cache[key]
        out[:, ix, iy] = res
        return out

    def _loo_systems(self, tots, nxe, store, batch, M_rows, yy):
        """Normal equations (base.py:568-628) of ``batch`` slots: whole-data accumulators minus the stored sketches of
        ``batch`` blocks (``store`` None: the whole-data slot).  Returns T ``[batch, ne+1, ne+1]``, q ``[batch, ne+1]``
        (NumPy); the inner products and the small algebra stay on the device, one copy brings T and q back."""
        from .engine import Row
        tot_A, tot_C, D_tot, tot_r = tots
        ne, k, nb = self.n_est, self.k, self.plan.n_bin
        use_cov = self.cov is not None
        dev, f64 = self.device, torch.float64
        nA = nC = nD = None
        if self.n_nxe:
            nA = nxe[0][0]
            if use_cov:      # only the last probe vector carries XXUz / UXXz (base.py:472-481, see pass2)
                nD = torch.zeros_like(nxe[0][0])
                nC = torch.zeros((k, self.cov.shape[1]) if self.cross else nD.shape, dtype=f64, device=dev)
                nC[k - 1] = self._c_form(nxe[1][0, k - 1])
                nD[k - 1] = self._apply_U(nxe[0][0, k - 1], f32=True)
        # Every inner product is a sum over probe columns: walk the column intervals on which each group sits in one
        # engine pass (a single interval unless num_random_vec exceeds what a pass holds) and add the pieces up.
        gram = tz = tk = gCD = None
        PA_l, PbA_l, PC_l, PD_l = [], [], [], []
        if use_cov:
            Wr = [Row(self.WT64[m]) for m in range(self.WT64.shape[0])]
            c = len(Wr)
        acc = lambda old, new: new if old is None else old + new
        for s0, nc in self.intervals:
            A = self._loo_rows(tot_A, store, "Z", "eZ", nA, s0, nc)
            # <XXz_a, XXz_b> of the slot, expanded around the whole-data rows: (t - b)(t - b)^T = tt - bt - (bt)^T + bb.
            # The whole-data rows are then read once per 64 block rows (matrix-core path) instead of once per block.
            A_tot = [Row(r.tot) for r in A]
            A_blk = [None if r.blk is None else Row(None, r.blk, r.mul, r.stride) for r in A]
            fixed = list(A_tot)
            if self.genie_model is not None:
                fixed.append(Row(self.Z64[s0:s0 + nc]))
            # <XXz_a, U Z>: U Z = W G lies in span(W), so it is also sum_{s, m} <A_a[s], W_m> G[m, s] -- the projections computed
            # below anyway.  Taken that way when the U Z row would be the 17th fixed row (RHE_DOM with 8 bins + covariates: 16
            # estimates): the one-read kernel takes 16 fixed rows per pass over the stored sketches.
            tk_proj = use_cov and (self.force_tk_proj or len(fixed) == 16)
            if use_cov and not tk_proj:
                fixed.append(Row(self.UZ64[s0:s0 + nc]))
            # one pass over the stored XXz rows: -<b_a, fixed>, the projections -<b_a[s], W_m> and <b_a, b_b>
            bpart, bA_flat, bA_self = self._blk_products(A_blk, fixed, Wr if use_cov else [], True, nc, batch)
            g = self._dots(A_tot, fixed, nc, 1) + bpart
            gram = acc(gram, g[:, :, :ne] + bpart[:, :, :ne].transpose(1, 2) + bA_self)
            col = ne
            if self.genie_model is not None:
                tz = acc(tz, g[:, :, col]); col += 1
            if use_cov:
                Dr = [Row(D_tot[a, s0:s0 + nc]) for a in range(D_tot.shape[0])] + ([Row(nD[s0:s0 + nc])] if nD is not None else [])
                tA = self._dots(self._sub_rows(A_tot, nc), Wr, 1, 1).reshape(1, ne, nc, c)      # whole-data parts: once per pass
                PA_l.append(tA + bA_flat); PbA_l.append(-bA_flat)
                tk = acc(tk, torch.einsum("zasm,ms->za", PA_l[-1], self.G64[:, s0:s0 + nc]) if tk_proj else g[:, :, col])
                if self.cross:
                    # XXUz is kept as W^T XXUz = S (Q W^T Z) only (see __init__); UXXz lies in span(W), so its products with
                    # XXUz go through W^T UXXz (whole-data rows, once per pass 2)
                    PD_l.append(self._dots(self._sub_rows(Dr, nc), Wr, 1, 1).reshape(1, ne, nc, c))
                    continue
                Cr = self._loo_rows(tot_C, store, "UZ", "eUZ", nC, s0, nc)
                C_tot = [Row(r.tot) for r in Cr]
                C_blk = [None if r.blk is None else Row(None, r.blk, r.mul, r.stride) for r in Cr]
                # one pass over the stored XXUz rows: -<c_a, UXXz_b> and the projections -<c_a[s], W_m>
                bC_sub, bC_flat, _ = self._blk_products(C_blk, Dr, Wr, False, nc, batch)
                gCD = acc(gCD, self._dots(C_tot, Dr, nc, 1) + bC_sub)     # <XXUz_a (LOO), UXXz_b (whole data)>
                # projections onto the covariates, P[z, a, s, m] = <row_a[s], W_m>; U = W Q W^T is applied through them:
                # <U A_a, A_b> = sum_s PA[a,s] Q PA[b,s],  <C_a, U blkA_b> = sum_s PC[a,s] Q PblkA[b,s]  (base.py:583-593;
                # UXXz_b = whole-data UXXz_b - U(block XXz_b), streaming_rhe.py:35-36)
                tC = self._dots(self._sub_rows(C_tot, nc), Wr, 1, 1).reshape(1, ne, nc, c)
                PC_l.append(tC + bC_flat)
        if use_cov:
            cat2 = lambda l: torch.cat(l, dim=2) if len(l) > 1 else l[0]
            PA, PbA = cat2(PA_l), cat2(PbA_l)
            Q = self.Q64
            flat = lambda P: P.reshape(P.shape[0], ne, k * c)
            uaa = flat(PA @ Q) @ flat(PA).transpose(1, 2)
            if self.cross:
                PC = tot_C if nC is None else torch.cat([tot_C, nC[None]], dim=0)          # [ne, k, c] whole data
                PC = PC[None]
                if store is not None:
                    blk = self._block_pc(store, batch)                                      # [batch, ne_g, k, c]
                    if self.n_nxe:
                        blk = torch.cat([blk, torch.zeros_like(blk[:, :1])], dim=1)
                    PC = PC - blk
                PD = cat2(PD_l)
                cd = flat(PC @ Q) @ (flat(PD) - flat(PbA)).transpose(1, 2)      # <XXUz_a, UXXz_b - U blk XXz_b>
            else:
                PC = cat2(PC_l)
                cd = gCD - flat(PC @ Q) @ flat(PbA).transpose(1, 2)
            gram = gram + cd - 2.0 * uaa
        # r = yXXy: whole-data minus block
        nt = self.n_traits
        r = tot_r[None].expand(batch, -1, -1)                     # [batch, ne_g, n_traits]
        if store is not None:
            py, cy = self.loc["y"]
            parts = [store[1][py][:batch, :, cy:cy + nt]]
            if self.dom:
                parts.append(store[3][py][:batch, :, cy:cy + nt])
            if self.n_gxe:
                pe, ce = self.loc["ey"]
                parts.append(store[1][pe][:batch, :, ce:ce + nt])
            r = r - torch.cat(parts, dim=1)
        if self.n_nxe:
            r = torch.cat([r, nxe[2][None].expand(batch, 1, nt)], dim=1)
        # one copy back: gram | tz | tk | r  -- no host->device traffic and no intermediate synchronisation in here; the
        # division by the SNP counts is host arithmetic on [batch, ne, ne]
        cols = [gram.reshape(batch, -1), r.reshape(batch, -1)]
        if tz is not None:
            cols.append(tz.expand(batch, -1) if tz.shape[0] != batch else tz)
        if tk is not None:
            cols.append(tk.expand(batch, -1) if tk.shape[0] != batch else tk)
        host = self._to_host(torch.cat(cols, dim=1))
        gram_h = host[:, :ne * ne].reshape(batch, ne, ne)
        r_h = host[:, ne * ne:ne * ne + ne * nt].reshape(batch, ne, nt)
        off = ne * ne + ne * nt
        tz_h = tk_h = None
        if tz is not None:
            tz_h = host[:, off:off + ne]; off += ne
        if tk is not None:
            tk_h = host[:, off:off + ne]
        Mf = M_rows.astype(np.float64)                            # [batch, ne]
        T = np.zeros((batch, ne + 1, ne + 1))
        q = np.zeros((batch, nt, ne + 1))                        # one right-hand side per phenotype, T is shared
        mm = Mf[:, :, None] * Mf[:, None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            T[:, :ne, :ne] = np.where(mm != 0, gram_h / k / mm, 0.0)
            tr = np.full((batch, ne), float(self.N))              # rhe.py:24-26
            if self.genie_model is not None and ne > nb:
                tr[:, nb:] = tz_h[:, nb:] / (k * Mf[:, nb:])       # stochastic trace of a GxE / NxE kernel (genie.py:85-89)
            if use_cov:
                tr = tr - tk_h / (k * Mf)                          # base.py:614-615
            q[:, :, :ne] = np.transpose(np.where(Mf[:, :, None] != 0, r_h / Mf[:, :, None], 0.0), (0, 2, 1))
        T[:, :ne, ne] = tr
        T[:, ne, :ne] = tr
        T[:, ne, ne] = self.N - (self.cov.shape[1] if use_cov else 0)
        q[:, :, ne] = np.asarray(yy)[None, :]
        return T, q

    def _block_pc(self, store, batch):
        """W^T XXUz of ``batch`` stored blocks ``[batch, ne_g, k, c]`` from their cross Gram matrices."""
        cr = [t[:batch] for t in store[4]]
        crd = None if store[5] is None else [t[:batch] for t in store[5]]
        return self._pc_of(self._cross_grams(cr, crd))

    def _stored_block(self, jl_i: int):
        sk, nm, cr = [b[jl_i] for b in self.blocks_sk], [b[jl_i] for b in self.blocks_nm], [b[jl_i] for b in self.blocks_cr]
        if self.dom:
            return sk, nm, [b[jl_i] for b in self.blocks_skd], [b[jl_i] for b in self.blocks_nmd], cr, [b[jl_i] for b in self.blocks_crd]
        return sk, nm, None, None, cr, None

    def _resketch(self, j: int):
        """Sketch block j again (the reference's pass 2 re-reads and re-imputes, base_streaming.py:110-118).  Blocks are
        visited in ascending order; missing calls get the SAME fill values as in pass 1: the reference's pass-2 worker
        is forked from the same parent state as its pass-1 worker (mp_handler.py:28-37), so it replays the same draws."""
        fill = None
        if self._check_missing and self._ingest_stats()[1].any():
            fill = self._block_fills(0)[j - self.j0]
        nb, dev = self.plan.n_bin, self.device
        rows = self._block_rows(j)
        pieces = self._pieces(j) or [(0, rows.shape[0]) + self._entries(j)]
        sk = [torch.empty((nb, e_.n_sketch, self.n_pad), dtype=torch.float32, device=dev) for e_ in self.engs[:self.n_pass]]
        nm = [torch.empty((nb, max(e_.n_norm, 1)), dtype=torch.float64, device=dev) for e_ in self.engs[:self.n_pass]]
        skd = [torch.empty_like(t) for t in sk] if self.dom else None
        nmd = [torch.empty_like(t) for t in nm] if self.dom else None
        cr = [torch.zeros((nb, nx, nx), dtype=torch.float64, device=dev) for nx in self.pass_cross]
        crd = [torch.zeros_like(t) for t in cr] if self.dom else None
        for p in range(self.n_pass):
            has_cr = self.pass_cross[p] > 0
            for i_, (a, b, ent, offs) in enumerate(pieces):
                self._sketch_call(p, rows[a:b], ent, offs, None if fill is None else fill[a:b].contiguous(), sk[p], nm[p],
                                  skd[p] if self.dom else None, nmd[p] if self.dom else None, accumulate=i_ > 0,
                                  ocr=cr[p] if has_cr else None, ocrd=crd[p] if (has_cr and self.dom) else None)
        return sk, nm, skd, nmd, cr, crd

    def _allreduce(self, t: torch.Tensor, async_op: bool = False):
        """Sum over the ranks, in place (RCCL over xGMI under the "nccl" backend); ``async_op``: returns the work handle."""
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _gather_systems(self, local, t_shape, q_shape):
        """Every rank's ``(j, T_j, q_j)`` to every rank as ONE tensor all-gather (the reference's result Queue,
        base_streaming.py:141-172): rows ``[j | T_j | q_j]`` float64, padded with j = -1 to the largest shard."""
        per = max(1, max(b - a for a, b in shard_ranges(self.plan.n_jack, self.world)))
        nt_, nq = int(np.prod(t_shape)), int(np.prod(q_shape))
        buf = np.full((per, 1 + nt_ + nq), -1.0)
        for i, (j, T, q) in enumerate(local):
            buf[i, 0] = j
            buf[i, 1:1 + nt_] = np.asarray(T).reshape(-1)
            buf[i, 1 + nt_:] = np.asarray(q).reshape(-1)
        mine = torch.from_numpy(buf).to(self.device)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        out = []
        for part in parts:
            rows = part.cpu().numpy()
            for r in rows[rows[:, 0] >= 0]:
                out.append((int(r[0]), r[1:1 + nt_].reshape(t_shape).copy(), r[1 + nt_:].reshape(q_shape).copy()))
        return out

    # ---- normal equations (base.py:568-628) ----------------------------------------------------
    def _system(self, A, C, D, r, M_row, yy: float):
        """A = XXz, C = XXUz, D = UXXz ``[ne, k, n_pad]`` float64, r = yXXy ``[ne]`` of one slot -> (T, q)."""
        ne, k = self.n_est, self.k
        A2 = A.reshape(ne, -1)
        gram = (A2 @ A2.T).cpu().numpy()
        use_cov = self.cov is not None
        if use_cov:
            UA = self._apply_U(A, f32=False).reshape(ne, -1)
            t1 = (UA @ A2.T).cpu().numpy()                       # <U A_a, A_b>
            if self.cross:      # C = W^T XXUz [ne, k, c]; UXXz lies in span(W): <XXUz_a, UXXz_b> = sum_s (C_a[s] Q) . (W^T UXXz_b[s])
                PD = D.reshape(ne, k, -1) @ self.W64
                t2 = ((C @ self.Q64).reshape(ne, -1) @ PD.reshape(ne, -1).T).cpu().numpy()
            else:
                t2 = (C.reshape(ne, -1) @ D.reshape(ne, -1).T).cpu().numpy()   # <XXUz_a, UXXz_b>
            tk_all = (A2 @ self.UZ64.reshape(-1)).cpu().numpy()
        if self.genie_model is not None:
            tz_all = (A2 @ self.Z64.reshape(-1)).cpu().numpy()
        r = r.cpu().numpy()
        T = np.zeros((ne + 1, ne + 1))
        q = np.zeros(ne + 1)
        for a in range(ne):
            for b in range(ne):
                t = gram[a, b]
                if use_cov:
                    t += (t2[a, b] - 2 * t1[a, b])
                t /= k
                mm = M_row[a] * M_row[b]
                T[a, b] = t / mm if mm != 0 else 0
        c = self.cov.shape[1] if use_cov else 0
        for a in range(ne):
            Ma = M_row[a]
            if self.genie_model is not None and a >= self.plan.n_bin:
                tr = tz_all[a] / (k * Ma)                        # stochastic trace of a GxE / NxE kernel (genie.py:85-89)
            else:
                tr = self.N                                      # rhe.py:24-26
            if use_cov:
                tk = tk_all[a]
                tk = 1 / (k * Ma) * tk                           # base.py:614-615
                tr = tr - tk
            T[a, ne] = tr
            T[ne, a] = tr
            q[a] = r[a] / Ma if Ma != 0 else 0
        T[ne, ne] = self.N - c
        q[ne] = yy
        return T, q
