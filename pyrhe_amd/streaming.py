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
  * the Gram matrices of the normal equations are float64 GEMMs on the GPU (torch, plumbing).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

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
        _, offs = self.entries(j)
        return np.diff(offs)


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
    T: np.ndarray        # [jn+1, ne+1, ne+1]
    q: np.ndarray        # [jn+1, ne+1]
    M: np.ndarray        # [jn+1, ne]


class SketchRunner:
    """Pass 1 (sketch every local block, accumulate, all-reduce) and pass 2 (leave-one-out systems).

    ``engine`` is a :class:`pyrhe_amd.engine.SketchEngine` (or anything with its ``set_rhs/sketch_block/
    block_stats/keep_vector`` interface: the multi-process CPU tests inject a stand-in, the product never does).
    ``geno`` holds the packed rows of THIS rank's blocks ``[j0, j1)`` on the device, first row = ``range(j0)[0]``.
    """

    def __init__(self, engine, geno: torch.Tensor, plan: BlockPlan, local_blocks: Tuple[int, int], *,
                 n_orig: int, missing_indv: Sequence[int], Z: np.ndarray, cov: Optional[np.ndarray] = None,
                 env: Optional[np.ndarray] = None, genie_model: Optional[str] = None,
                 impute: str = "binary", seed: int = 0, streaming: bool = True, compat: str = "reference",
                 recompute_pass2: bool = False, rng_prefix_draws: int = 0, group=None):
        self.eng = engine
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
        self.genie_model = genie_model
        self.impute = impute
        self.seed = int(seed)
        self.streaming = bool(streaming)
        if compat not in ("reference", "fixed"):
            raise ValueError("compat must be 'reference' or 'fixed'")
        self.compat = compat
        self.recompute_pass2 = bool(recompute_pass2)
        self.rng_prefix_draws = int(rng_prefix_draws)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.device = geno.device
        self.n_pad = engine.n_pad
        self.row0 = plan.range(self.j0)[0] if self.j1 > self.j0 else 0
        self._keep_t = torch.as_tensor(self.keep_idx, device=self.device)
        self.keep = engine.keep_vector(self.missing)
        self.n_gxe = plan.n_bin if genie_model in ("G+GxE", "G+GxE+NxE") else 0
        self.n_est = plan.n_bin + self.n_gxe
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
        self.Z64 = self._pad_cols(torch.from_numpy(np.ascontiguousarray(self.Z.T)))        # [k, n_pad] f64
        self._entries_cache = {}
        self._stats = None
        self.timings = {}

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
        """Columns handed to the engine.  Sketch groups (k columns each): Z, [UZ], [e.Z]; norm: y~, [e.y~]."""
        k = self.k
        groups = [self.Z64]
        self.g_Z, self.g_UZ, self.g_eZ = 0, None, None
        if self.cov is not None:
            self.g_UZ = len(groups)
            groups.append(self.UZ64)
        y64 = np.asarray(y, np.float64).reshape(-1)
        if self.cov is not None:          # regress_pheno (base.py:396-401), float64 on the host
            W = self.cov
            y64 = y64 - W @ (np.linalg.pinv(W.T @ W) @ (W.T @ y64))
        self.y_res = y64
        norms = [self._pad_cols(torch.from_numpy(y64[None, :]))]
        if self.n_gxe:
            e64 = self._pad_cols(torch.from_numpy(self.env[None, :]))
            self.env_pad = e64[0]
            self.g_eZ = len(groups)
            groups.append(self.Z64 * e64)
            if self.cov is not None:
                raise NotImplementedError("GENIE G+GxE with covariates needs > 30 right-hand columns")
            norms.append(norms[0] * e64)
        sk = torch.cat(groups, dim=0).to(torch.float32)
        nm = torch.cat(norms, dim=0).to(torch.float32)
        self.eng.set_rhs(sk, nm, self.keep)
        self.n_sketch = sk.shape[0]
        self.n_norm = nm.shape[0]

    # ---- imputation fill values ---------------------------------------------------------------
    def _block_rows(self, j: int) -> torch.Tensor:
        s, e = self.plan.range(j)
        return self.geno[s - self.row0: e - self.row0]

    def _ingest_stats(self):
        """Per-SNP observed sums / missing counts of the local rows (one extra read, once per cohort)."""
        if self._stats is None:
            s, nm = self.eng.block_stats(self.geno)
            self._stats = (s.cpu().numpy().astype(np.int64), nm.cpu().numpy().astype(np.int64))
        return self._stats

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
        base_streaming.py:85-104,110-118): non-streaming reseeds per block, streaming keeps drawing."""
        if self.impute != "binary":
            return lambda j, m: None
        if not self.streaming:
            return lambda j, m: np.random.RandomState(self.seed).random_sample(m)
        rs = np.random.RandomState(self.seed)
        if self.rng_prefix_draws:
            rs.randint(0, 2, self.rng_prefix_draws)
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

    def sketch_local_blocks(self, out_sk: Optional[torch.Tensor], out_norm: Optional[torch.Tensor],
                            totals_sk: Optional[torch.Tensor] = None, pass_index: int = 0):
        """Enqueue the sketch of every local block.  ``out_sk [jl, nbin, n_sketch, n_pad]`` receives the per-block
        sketches (or, if None, ``totals_sk [nbin, n_sketch, n_pad]`` float32 is accumulated in place)."""
        any_missing = bool(self._check_missing and self._ingest_stats()[1].any())
        draw = self._uniform_stream(pass_index) if any_missing else None
        for jl, j in enumerate(range(self.j0, self.j1)):
            ent, offs = self._entries(j)
            s, e = self.plan.range(j)
            fill = None
            if any_missing:
                u = draw(j, e - s)                    # one uniform per SNP, missing or not (base.py:266,285)
                fill = self._fill_for_block(j, u)
            if out_sk is not None:
                self.eng.sketch_block(self._block_rows(j), ent, offs, fill=fill, out_sketch=out_sk[jl],
                                      out_norm=out_norm[jl], accumulate=False)
            else:
                self.eng.sketch_block(self._block_rows(j), ent, offs, fill=fill, out_sketch=totals_sk,
                                      out_norm=out_norm[jl], accumulate=True)

    def _apply_U(self, v: torch.Tensor, f32: bool) -> torch.Tensor:
        """``W (Q (W^T v))`` for rows of v ``[..., n_pad]``; float32 like _compute_UXXz (base.py:407-408) or float64
        like setup_lhs_rhs_jackknife (base.py:584-586).  Result float64."""
        W, Q = (self.W32, self.Q32) if f32 else (self.W64, self.Q64)
        x = v.to(W.dtype)
        shp = x.shape
        x = x.reshape(-1, shp[-1])
        out = ((x @ W) @ Q.T) @ W.T
        return out.reshape(shp).to(torch.float64)

    def run_trait(self, y: np.ndarray, check_missing: bool = True) -> Systems:
        """Pass 1 + pass 2 for one centred phenotype ``y [N]``; returns the jn+1 normal-equation systems (every rank)."""
        self.begin_trait(y, check_missing)
        self.pass1()
        return self.pass2()

    def begin_trait(self, y: np.ndarray, check_missing: bool = True):
        """Install the right-hand columns and allocate the per-block sketch store."""
        nb, jl = self.plan.n_bin, self.j1 - self.j0
        self._check_missing = check_missing
        self._install_rhs(y)
        dev = self.device
        self.blocks_sk = torch.empty((jl, nb, self.n_sketch, self.n_pad), dtype=torch.float32, device=dev)
        self.blocks_nm = torch.zeros((max(jl, 1), nb, self.n_norm), dtype=torch.float64, device=dev)

    def pass1(self):
        """One streaming sketch pass over the local blocks (base_streaming.py:85-104) + aggregate
        (base_streaming.py:21-58 / base.py:465-500: float64 sums in block order) + the cross-GPU all-reduce."""
        plan, k, nb = self.plan, self.k, self.plan.n_bin
        jn, jl, dev = plan.n_jack, self.j1 - self.j0, self.device
        blocks_sk, blocks_nm = self.blocks_sk, self.blocks_nm
        self.sketch_local_blocks(blocks_sk, blocks_nm, pass_index=0)
        tot_sk = torch.zeros((nb, self.n_sketch, self.n_pad), dtype=torch.float64, device=dev)
        tot_nm = torch.zeros((nb, self.n_norm), dtype=torch.float64, device=dev)
        use_cov = self.cov is not None
        D_tot = torch.zeros((nb, k, self.n_pad), dtype=torch.float64, device=dev) if use_cov else None
        wsum = torch.zeros_like(D_tot) if (use_cov and self.world > 1) else None
        q4 = use_cov and self.streaming and self.compat == "reference"
        for jl_i in range(jl):
            blk = blocks_sk[jl_i]
            tot_sk += blk
            tot_nm += blocks_nm[jl_i]
            if use_cov:
                if q4 and self.world == 1:
                    # quirk Q4 (streaming_rhe.py:18-21): U applied to the RUNNING sum of XXz, float32
                    D_tot += self._apply_U(tot_sk[:, :k], f32=True)
                elif q4:
                    wsum += blk[:, :k].to(torch.float64) * float(jn - (self.j0 + jl_i))
                else:
                    D_tot += self._apply_U(blk[:, :k], f32=True)
        if self.world > 1:
            self._allreduce(tot_sk)
            self._allreduce(tot_nm)
            if use_cov:
                if q4:   # U is linear: sum_j U(running_j) = U(sum_j (jn - j) XXz_j)  (float32 rounding order differs)
                    self._allreduce(wsum)
                    D_tot = self._apply_U(wsum, f32=True)
                else:
                    self._allreduce(D_tot)
        self.tot_sk, self.tot_nm, self.D_tot = tot_sk, tot_nm, D_tot

    def pass2(self) -> Systems:
        """Leave-one-out normal equations where each block lives (base_streaming.py:110-144), gathered to every rank."""
        plan, k, ne = self.plan, self.k, self.n_est
        jn = plan.n_jack
        use_cov = self.cov is not None
        tot_sk, tot_nm, D_tot = self.tot_sk, self.tot_nm, self.D_tot
        # M (rhe.py:16): SNPs of every estimate outside block j
        M = np.zeros((jn + 1, ne), np.int64)
        last = np.concatenate([plan.len_bin, plan.len_bin]) if self.n_gxe else plan.len_bin
        M[jn] = last
        for j in range(jn):
            c = plan.block_counts(j)
            M[j] = last - (np.concatenate([c, c]) if self.n_gxe else c)
        yy = float(self.y_res @ self.y_res)
        local = []
        for jl_i, j in enumerate(range(self.j0, self.j1)):
            if self.recompute_pass2:
                sk, nm, _ = self._resketch(j)
            else:
                sk, nm = self.blocks_sk[jl_i], self.blocks_nm[jl_i]
            slot = jn if (jn == 1 and j == 0) else j          # base.py:654-655
            if slot == jn:
                T, q = self._system(tot_sk, tot_nm, D_tot, M[jn], yy)
            else:
                blk = sk.to(torch.float64)
                D = None
                if use_cov:
                    D = D_tot - self._apply_U(blk[:, :k], f32=True)      # streaming_rhe.py:35-36
                T, q = self._system(tot_sk - blk, tot_nm - nm, D, M[j], yy)
            local.append((j, T, q))
        T_tot, q_tot = self._system(tot_sk, tot_nm, D_tot, M[jn], yy)
        if self.world > 1:
            gathered = [None] * self.world
            dist.all_gather_object(gathered, local, group=self.group)
            local = [item for part in gathered for item in part]
        local.sort(key=lambda t: t[0])
        Ts = np.stack([t[1] for t in local] + [T_tot])
        qs = np.stack([t[2] for t in local] + [q_tot])
        return Systems(Ts, qs, M)

    def _resketch(self, j: int):
        ent, offs = self._entries(j)
        return self.eng.sketch_block(self._block_rows(j), ent, offs)

    def _allreduce(self, t: torch.Tensor):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    # ---- normal equations (base.py:568-628) ----------------------------------------------------
    def _estimate_tensors(self, sk: torch.Tensor, nm: torch.Tensor, D: Optional[torch.Tensor]):
        """Per-estimate A = XXz, C = XXUz, D = UXXz ``[ne, k, n_pad]`` float64 and r = yXXy ``[ne]``."""
        k = self.k
        A = sk[:, 0:k]
        C = sk[:, self.g_UZ * k:(self.g_UZ + 1) * k] if self.g_UZ is not None else None
        r = nm[:, 0]
        if self.n_gxe:
            e = self.env_pad
            Ae = sk[:, self.g_eZ * k:(self.g_eZ + 1) * k] * e       # X_e X_e^T z = e . (X X^T (e . z))
            A = torch.cat([A, Ae], dim=0)
            r = torch.cat([r, nm[:, 1]], dim=0)
        return A, C, D, r

    def _system(self, sk, nm, D, M_row, yy: float):
        ne, k = self.n_est, self.k
        A, C, D, r = self._estimate_tensors(sk, nm, D)
        A2 = A.reshape(ne, -1)
        gram = (A2 @ A2.T).cpu().numpy()
        use_cov = self.cov is not None
        if use_cov:
            UA = self._apply_U(A, f32=False).reshape(ne, -1)
            t1 = (UA @ A2.T).cpu().numpy()                       # <U A_a, A_b>
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
                tr = tz_all[a] / (k * Ma)                        # stochastic trace of a GxE kernel (genie.py:85-89)
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
