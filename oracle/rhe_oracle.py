"""CPU restatement of PyRHE's (streaming) randomized-trace sketch + estimator.

TEST INFRASTRUCTURE -- the product (``pyrhe_amd``) never imports this module.  It is the
checker for the HIP path and the timed ``cpu_baseline`` ("port") in ``bench.py``.

Pinned by: ``tests/golden/*.npz`` -- outputs of the reference itself, run in the build
container by ``oracle/make_goldens.py`` (``tests/test_oracle_golden.py`` checks every array).

All ``file:line`` citations are into ``/root/reference/``.  The arithmetic follows the
reference's order and dtypes exactly (float32 decode/standardize, float32 torch-CPU GEMV
pairs, float64 accumulation), so per-block sketches agree with the goldens bit-for-bit on
the same torch build.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import scipy.linalg
import torch

from . import plink


# --------------------------------------------------------------------------- block indexing
def block_range(j: int, n_snp: int, n_jack: int) -> tuple:
    """SNP range of jackknife block ``j``  (pyrhe/src/base/base.py:362-371).

    ``step = M // jn``; block j starts at ``j*step``; the LAST block also takes ``M % jn``.
    """
    step = n_snp // n_jack
    start = j * step
    length = step if j < n_jack - 1 else step + n_snp % n_jack
    return start, start + length


def distribute_work(n_jobs: int, n_workers: int) -> List[tuple]:
    """Contiguous ``ceil(jobs/workers)`` ranges (base.py:530-533) -- also the per-GPU shard rule."""
    per = int(np.ceil(n_jobs / n_workers))
    return [(i * per, min((i + 1) * per, n_jobs)) for i in range(n_workers)]


# --------------------------------------------------------------------------- genotype ingest
def read_geno(packed: np.ndarray, n_indv_orig: int, start: int, end: int,
              missing_indv: Sequence[int]) -> np.ndarray:
    """decode -> drop missing individuals -> 0<->2 flip  (base.py:338-356).

    Returns float32 [N, m] in C order (``subsample.copy()`` in the reference yields C order),
    values = count of allele 2, NaN = missing call.
    """
    sub = plink.decode_count_a1(packed[start:end], n_indv_orig)        # bed_reader.read, F order
    if len(missing_indv):
        sub = np.delete(sub, list(missing_indv), axis=0)               # base.py:343-345
    flipped = sub.copy()                                               # base.py:352
    flipped[sub == 2] = 0
    flipped[sub == 0] = 2
    return flipped


def hwe_draw(p: float) -> int:
    """One genotype from HWE(p) using ONE ``np.random.random()`` (base.py:265-274)."""
    r = np.random.random()
    p0 = (1 - p) * (1 - p)
    p1 = 2 * p * (1 - p)
    if r < p0:
        return 0
    if r < p0 + p1:
        return 1
    return 2


def impute_geno(X: np.ndarray, method: str = "binary") -> np.ndarray:
    """Per-SNP imputation, in place (base.py:277-289).

    binary: one RNG draw per SNP *whether or not anything is missing*; all missing entries of
    the SNP get the same drawn genotype.  mean: missing -> 0 (before standardisation).
    """
    for m in range(X.shape[1]):
        col = X[:, m]
        miss = np.isnan(col)
        if method == "binary":
            obs_mean = np.nanmean(col)
            col[miss] = hwe_draw(obs_mean * 0.5)
        else:
            col[miss] = 0
    return X


def partition_bins(X: np.ndarray, sub_annot: np.ndarray) -> List[np.ndarray]:
    """Column subsets per annotation bin; bins may overlap or miss SNPs (base.py:315-336)."""
    return [X[:, np.nonzero(sub_annot[:, b])[0].tolist()] for b in range(sub_annot.shape[1])]


def standardize_geno(G: np.ndarray) -> np.ndarray:
    """``(g - mean) / sqrt(mean*(1-mean/2))`` in float32, no zero-variance guard (base.py:291-296)."""
    means = np.mean(G, axis=0)
    inv_sd = 1 / np.sqrt(means * (1 - 0.5 * means))
    return (G - means) * inv_sd


def encode_geno_dom(G: np.ndarray):
    """Dominance encoding (rhe_dom.py:23-41): 0 -> 0, 1 -> 2p, 2 -> 4p - 2 with p = mean/2 (float32), stored float64."""
    maf = np.mean(G, axis=0) / 2
    enc = np.zeros_like(G, dtype=np.float64)
    enc += (G == 1) * (2 * maf[np.newaxis, :])
    enc += (G == 2) * (4 * maf[np.newaxis, :] - 2)
    return enc, maf


def standardize_geno_dom(enc: np.ndarray, maf: np.ndarray) -> np.ndarray:
    """(enc - mean) / (2 p (1 - p))  (rhe_dom.py:15-21)."""
    means = np.mean(enc, axis=0)
    inv = 1 / (2 * maf * (1 - maf))
    return (enc - means) * inv


# --------------------------------------------------------------------------- the four sketches
def _t32(x: np.ndarray) -> torch.Tensor:
    """``to_tensor``: every operand becomes a float32 CPU tensor (pyrhe/src/util/mat_mul.py:4-15)."""
    return torch.from_numpy(x).float()


# Accumulation order of the float32 products.  "reference": torch's float32 ``@`` on this build, what the reference itself
# runs (mat_mul.py:27; the order inside the BLAS kernel is not contractual).  "exact": the same float32 operands,
# every product summed in float64 and rounded to float32 ONCE -- the correctly rounded value of each ``@``.  The distance
# between the two is the reference's own float32 accumulation noise: tests/ use it to tell a result that differs from the
# reference by its accumulation order from one that is wrong (tests/helpers.py: accumulation_noise_floor).
ACCUM = "reference"


def _mm(*mats: np.ndarray) -> np.ndarray:
    """Chained ``@`` left to right on float32 tensors, back to NumPy (mat_mul.py:17-31)."""
    acc = _t32(mats[0])
    for m in mats[1:]:
        if ACCUM == "exact":
            acc = (acc.double() @ _t32(m).double()).float()
        else:
            acc = acc @ _t32(m)
    return acc.numpy()


def sketch_XXz(X: np.ndarray, z: np.ndarray) -> np.ndarray:
    """``X (X^T z)`` for ONE probe vector, float32, flattened (base.py:403-405,410-412)."""
    return _mm(X, _mm(X.T, z.reshape(-1, 1))).flatten()


def project_U(v: np.ndarray, W: np.ndarray, Q: np.ndarray) -> np.ndarray:
    """``W (Q (W^T v))`` in float32 (base.py:407-408)."""
    return _mm(W, _mm(Q, _mm(W.T, v))).flatten()


def regress_pheno(W: np.ndarray, y: np.ndarray) -> np.ndarray:
    """``y - W pinv(W^T W) W^T y`` in float64 (base.py:396-401)."""
    Q = np.linalg.pinv(W.T @ W)
    return y - W @ (Q @ (W.T @ y))


def sketch_yXXy(X: np.ndarray, y: np.ndarray, W: Optional[np.ndarray]) -> float:
    """``||X^T y~||^2`` in float32 (base.py:414-417)."""
    yy = y if W is None else regress_pheno(W, y)
    v = _mm(X.T, yy)
    return float(_mm(v.T, v)[0][0])


# --------------------------------------------------------------------------- problem + state
@dataclass
class Problem:
    packed: np.ndarray                 # [M, ceil(N_orig/4)] uint8
    n_indv_orig: int
    annot: np.ndarray                  # [M, nbin] 0/1
    pheno: np.ndarray                  # [N, 1] float64, centred, missing individuals removed
    n_jack: int
    k: int
    seed: int
    missing_indv: List[int] = field(default_factory=list)
    cov: Optional[np.ndarray] = None   # [N, c]
    env: Optional[np.ndarray] = None   # [N, 1]  (GENIE, one environment)
    genie_model: Optional[str] = None  # None (RHE) | "G" | "G+GxE"
    impute: str = "binary"
    Z: Optional[np.ndarray] = None     # [N, k] float64 probe vectors (all_zb)
    dom: bool = False                  # RHE_DOM: a dominance estimate per bin after the additive ones
    # "reference": every quirk of the reference kept.  "fixed": the model as its equations state it -- the LAST estimate is a
    # heteroscedastic-noise (N x E) slot only when the model has one ("G+GxE+NxE"); for "G+GxE" every estimate's total is the
    # sum over its blocks (genie.py:46-94 without base.py:467,472-481's overwrite).  This is the definition the product gives
    # StreamingGENIE("G+GxE") (BASELINE config 5), which the reference cannot run (base_streaming.py:45-46 raises).
    compat: str = "reference"

    def __post_init__(self):
        if self.cov is not None:
            # ``read_cov`` returns ``df.values`` (file_processing.py:185), which pandas hands out
            # Fortran-ordered; torch's float32 GEMV picks its kernel by layout, so keep it.
            self.cov = np.asfortranarray(self.cov, dtype=np.float64)

    @property
    def n_snp(self) -> int:
        return self.packed.shape[0]

    @property
    def n_indv(self) -> int:
        return self.n_indv_orig - len(self.missing_indv)

    @property
    def nbin(self) -> int:
        return self.annot.shape[1]

    @property
    def has_gxe(self) -> bool:
        return self.genie_model in ("G+GxE", "G+GxE+NxE")

    @property
    def has_nxe(self) -> bool:
        return self.genie_model == "G+GxE+NxE"

    @property
    def n_est(self) -> int:
        """genie.py:26-34 (one environment); rhe_dom.py:7-9."""
        if self.dom:
            return 2 * self.nbin
        return self.nbin * (2 if self.has_gxe else 1) + (1 if self.has_nxe else 0)

    def len_bin(self) -> np.ndarray:
        return (self.annot == 1).sum(axis=0)

    def m_last_row(self) -> np.ndarray:
        """genie.py:36-44."""
        lb = self.len_bin()
        if self.dom:                                  # rhe_dom.py:11-13
            return np.concatenate([lb, lb])
        parts = [lb] + ([lb] if self.has_gxe else []) + ([np.ones(1, lb.dtype)] if self.has_nxe else [])
        return np.concatenate(parts)


def make_probe_vectors(seed: int, n_indv: int, k: int) -> np.ndarray:
    """``np.random.seed(seed); randn(N, k)`` -- NumPy legacy stream (base.py:72-73,176)."""
    np.random.seed(seed)
    return np.random.randn(n_indv, k)


@dataclass
class BlockSketch:
    XXz: np.ndarray      # [n_est, k, N] float64 holding float32 values
    yXXy: np.ndarray     # [n_est]
    m: np.ndarray        # [n_est] SNPs of this block per estimate
    XXUz: Optional[np.ndarray] = None
    UXXz: Optional[np.ndarray] = None   # U applied to THIS block's XXz ("fixed" semantics)


def sketch_block(p: Problem, j: int, reseed: bool) -> BlockSketch:
    """Everything the reference does for one jackknife block in pass 1.

    Non-streaming: base.py:503-521 (reseeds NumPy per block, :510) + rhe.py:13-22 / genie.py:46-75.
    Streaming:     base_streaming.py:85-104 (no reseed) + streaming_rhe.py:13-28.
    """
    if reseed:
        np.random.seed(p.seed)
    start, end = block_range(j, p.n_snp, p.n_jack)
    X = read_geno(p.packed, p.n_indv_orig, start, end, p.missing_indv)
    X = impute_geno(X, p.impute)
    assert X.shape[0] == p.n_indv
    bins = partition_bins(X, p.annot[start:end])
    ne, k, N = p.n_est, p.k, p.n_indv
    out = BlockSketch(np.zeros((ne, k, N)), np.zeros(ne), np.zeros(ne, np.int64))
    use_cov = p.cov is not None
    if use_cov:
        Q = np.linalg.pinv(p.cov.T @ p.cov)
        UZ = p.cov @ Q @ (p.cov.T @ p.Z)
        out.XXUz = np.zeros((ne, k, N))
        out.UXXz = np.zeros((ne, k, N))

    def one(est: int, Xs: np.ndarray):
        out.m[est] = Xs.shape[1]
        for b in range(k):
            out.XXz[est, b] = sketch_XXz(Xs, p.Z[:, b])
            if use_cov:
                out.UXXz[est, b] = project_U(out.XXz[est, b], p.cov, Q)
                out.XXUz[est, b] = sketch_XXz(Xs, UZ[:, b])
        out.yXXy[est] = sketch_yXXy(Xs, p.pheno, p.cov)

    for b_idx, Gb in enumerate(bins):
        one(b_idx, standardize_geno(Gb))
    if p.dom:                                           # rhe_dom.py:56-68
        for b_idx, Gb in enumerate(bins):
            one(b_idx + p.nbin, standardize_geno_dom(*encode_geno_dom(Gb)))
    if p.has_gxe:                                       # genie.py:59-75 ; k_gxe = (e+1)*k + nbin, e = 0
        for b_idx, Gb in enumerate(bins):
            Xs = standardize_geno(Gb)
            Xe = (_t32(Xs) * _t32(p.env.reshape(-1, 1))).numpy()   # elem_mul, mat_mul.py:34-47
            one(b_idx + p.nbin, Xe)
    return out


@dataclass
class State:
    """What ``setup_lhs_rhs_jackknife`` reads: slot j<jn = leave-block-j-out, slot jn = total."""
    XXz: np.ndarray      # [n_est, jn+1, k, N]
    yXXy: np.ndarray     # [n_est, jn+1]
    M: np.ndarray        # [jn+1, n_est]
    XXUz: Optional[np.ndarray] = None
    UXXz: Optional[np.ndarray] = None
    blocks: Optional[List[BlockSketch]] = None


def precompute(p: Problem, streaming_compat: bool = False) -> State:
    """Pass 1 + aggregate.

    ``streaming_compat=False``: non-streaming semantics (base.py:465-500, 535-554).
    ``streaming_compat=True`` : one-worker streaming semantics incl. quirk Q4 -- ``UXXz`` accumulates
    ``U`` applied to the RUNNING sum of ``XXz`` (streaming_rhe.py:18-21), and pass 2 subtracts ``U`` applied
    to the block's own ``XXz`` (streaming_rhe.py:30-43).  No per-block reseed (base_streaming.py:85-104).
    """
    if p.Z is None:
        p.Z = make_probe_vectors(p.seed, p.n_indv, p.k)
    if streaming_compat:
        # The one-worker multiprocessing mode forks its pass-1 worker from the parent right after construction, so the
        # worker's legacy stream stands where Base.__init__ left it: seed (base.py:73), then randn(N, k) (base.py:176).
        # The pass-2 worker is forked from the SAME parent state (mp_handler.py:28-37), so it re-imputes every block
        # with the same draws: both passes see identical genotypes, which is what reusing `blocks` below expresses.
        np.random.seed(p.seed)
        np.random.randn(p.n_indv, p.k)
    jn, ne, k, N = p.n_jack, p.n_est, p.k, p.n_indv
    blocks = [sketch_block(p, j, reseed=not streaming_compat) for j in range(jn)]
    st = State(np.zeros((ne, jn + 1, k, N)), np.zeros((ne, jn + 1)), np.zeros((jn + 1, ne), np.int64), blocks=blocks)
    st.M[jn] = p.m_last_row()
    use_cov = p.cov is not None
    if use_cov:
        st.XXUz = np.zeros_like(st.XXz)
        st.UXXz = np.zeros_like(st.XXz)
        Q = np.linalg.pinv(p.cov.T @ p.cov)
    for j, blk in enumerate(blocks):
        st.M[j] = st.M[jn] - blk.m                      # rhe.py:16 / streaming_rhe.py:16
    n_env = 1 if p.genie_model is not None else 0       # GENIE sets num_env = 1 for every genie_model (genie.py:18)
    if p.compat == "fixed" and not p.has_nxe:
        n_env = 0
    for e in range(ne):
        # streaming: aggregate skips the last num_env estimates and then fills the N x E slots from the same closed form
        # (base_streaming.py:24-58); only "G+GxE+NxE" reaches that code without an IndexError
        if e >= ne - n_env and (not streaming_compat or p.genie_model == "G+GxE+NxE"):
            # Reference quirk (base.py:467,472-481): the LAST estimate of a GENIE model is always treated
            # as "hetero noise", also for "G"/"G+GxE": its total slot is the N x E sketch diag(env)^2 z
            # (closed form of elem_mul(eye(N), env)), not the sum over blocks.
            env32 = p.env.reshape(-1).astype(np.float32)
            for b in range(k):
                st.XXz[e, jn, b] = env32 * (env32 * p.Z[:, b].astype(np.float32))
            yy = p.pheno if not use_cov else regress_pheno(p.cov, p.pheno)
            v = env32 * yy.reshape(-1).astype(np.float32)
            st.yXXy[e, jn] = float(_mm(v.reshape(1, -1), v.reshape(-1, 1))[0][0])   # v.T @ v as the reference shapes it
            if use_cov:                                  # only the last b is set (base.py:479-481)
                UZ = p.cov @ Q @ (p.cov.T @ p.Z)
                b = k - 1
                st.UXXz[e, jn, b] = project_U(st.XXz[e, jn, b], p.cov, Q)
                st.XXUz[e, jn, b] = env32 * (env32 * UZ[:, b].astype(np.float32))
        elif not streaming_compat:
            for j, blk in enumerate(blocks):            # same order as base.py:468-471
                st.XXz[e, jn] += blk.XXz[e]
                st.yXXy[e, jn] += blk.yXXy[e]
                if use_cov:
                    st.XXUz[e, jn] += blk.XXUz[e]
                    st.UXXz[e, jn] += blk.UXXz[e]
        else:
            for j, blk in enumerate(blocks):
                for b in range(k):
                    st.XXz[e, jn, b] += blk.XXz[e, b]
                    if use_cov:                         # quirk Q4: U(running sum)
                        st.UXXz[e, jn, b] += project_U(st.XXz[e, jn, b], p.cov, Q)
                        st.XXUz[e, jn, b] += blk.XXUz[e, b]
                st.yXXy[e, jn] += blk.yXXy[e]
        for j, blk in enumerate(blocks):                # leave-one-out slots (base.py:483-500 / streaming pass 2)
            st.XXz[e, j] = st.XXz[e, jn] - blk.XXz[e]
            st.yXXy[e, j] = st.yXXy[e, jn] - blk.yXXy[e]
            if use_cov:
                st.XXUz[e, j] = st.XXUz[e, jn] - blk.XXUz[e]
                st.UXXz[e, j] = st.UXXz[e, jn] - blk.UXXz[e]
    return st


# --------------------------------------------------------------------------- normal equations
def b_trace(p: Problem, st: State, e: int, j: int) -> float:
    """``tr(K_e)``: N for genetic bins (rhe.py:24-26); stochastic for GxE bins (genie.py:84-94)."""
    if p.genie_model is not None and e >= p.nbin:
        return float(np.sum(st.XXz[e, j] * p.Z.T) / (p.k * st.M[j, e]))
    return float(p.n_indv)


def normal_equations(p: Problem, st: State, j: int) -> tuple:
    """``T sigma = q`` for slot j  (base.py:568-628)."""
    ne, k, N = p.n_est, p.k, p.n_indv
    T = np.zeros((ne + 1, ne + 1))
    q = np.zeros((ne + 1, 1))
    use_cov = p.cov is not None
    if use_cov:
        W = p.cov
        Q = np.linalg.pinv(W.T @ W)
        UZ = W @ Q @ (W.T @ p.Z)
    for a in range(ne):
        for b in range(ne):
            Ma, Mb = st.M[j, a], st.M[j, b]
            A, B = st.XXz[a, j], st.XXz[b, j]
            t = np.sum(A * B)
            if use_cov:
                UA = W @ (Q @ (W.T @ A.T))
                t1 = np.sum(UA.T * B)
                t2 = np.sum(st.XXUz[a, j] * st.UXXz[b, j])
                t += t2 - 2 * t1
            t /= k
            T[a, b] = t / (Ma * Mb) if (Ma * Mb) != 0 else 0
    for a in range(ne):
        Ma = st.M[j, a]
        tr = b_trace(p, st, a, j)
        if use_cov:
            tk = np.sum(st.XXz[a, j] * UZ.T)
            tk = 1 / (k * Ma) * tk                     # base.py:614-615 (this order)
            tr = tr - tk
        T[a, ne] = tr
        T[ne, a] = tr
        q[a] = st.yXXy[a, j] / Ma if Ma != 0 else 0
    T[ne, ne] = N if not use_cov else N - p.cov.shape[1]
    yy = p.pheno if not use_cov else regress_pheno(p.cov, p.pheno)
    q[ne] = yy.T @ yy
    return T, q


def solve(T: np.ndarray, q: np.ndarray, method: str = "QR") -> np.ndarray:
    """QR (base.py:306-312) or lstsq (base.py:298-303)."""
    if method == "QR":
        Qm, R = scipy.linalg.qr(T)
        return np.ravel(scipy.linalg.solve_triangular(R, np.dot(Qm.T, q)))
    if method == "lstsq":
        return np.ravel(np.linalg.lstsq(T, q, rcond=None)[0])
    raise ValueError("Unsupported method for solving linear equation")


def estimate(p: Problem, st: State, method: str = "QR"):
    """All jn+1 solves (base.py:630-678; genie.py:97-144 adds the trace-adjusted copy)."""
    sig, sig_adj, Ts, qs = [], [], [], []
    for j in range(p.n_jack + 1):
        jj = 1 if (p.n_jack == 1 and j == 0) else j       # base.py:654-655
        T, q = normal_equations(p, st, jj)
        s = solve(T, q, method)
        sig.append(s)
        sig_adj.append(s * T[:, p.n_est])                 # genie.py:127-130
        Ts.append(T); qs.append(q)
    sig = np.array(sig); sig_adj = np.array(sig_adj)
    return sig[:-1], sig[-1], sig_adj[:-1], sig_adj[-1], np.array(Ts), np.array(qs)


def jackknife_se(ests: np.ndarray, n_jack: int) -> np.ndarray:
    """``sqrt((jn-1)/jn * sum (theta_j - mean)^2)``  (base.py:680-703)."""
    out = []
    for c in range(ests.shape[1]):
        v = ests[:, c]
        mean = np.mean(v)
        sq = 0
        for x in v:
            sq += (x - mean) ** 2
        out.append(np.sqrt((n_jack - 1) * sq / n_jack))
    return np.array(out)


def h2_nonoverlapping(sig_all: np.ndarray, ne: int) -> np.ndarray:
    """RHE h2 rows: [h2_0..h2_{ne-1}, h2_SNP]  (base.py:705-742)."""
    rows = []
    for s in sig_all:
        total = 0
        for e in range(ne):
            total += s[e]
        rows.append([x / (total + s[-1]) for x in s[:-1]] + [total / (total + s[-1])])
    return np.array(rows)


def h2_overlapping(p: Problem, st: State, sig_all: np.ndarray) -> np.ndarray:
    """Overlapping-annotation h2 (base.py:744-785); per-SNP sum of sigma_bin / M[j][bin]."""
    rows = []
    ne = p.n_est
    for j, s in enumerate(sig_all):
        total = 0
        for e in range(ne):
            total += s[e]
        # base.py:382-393: rows outside block j.  NB for j == jn the reference's arithmetic yields
        # start = jn*step, chunk = step + rem  -> mask[start:end] is empty when rem == 0.
        step = p.n_snp // p.n_jack
        rem = p.n_snp % p.n_jack
        chunk = step if j < p.n_jack - 1 else step + rem
        s0 = j * step
        mask = np.ones(p.n_snp, bool)
        mask[s0:s0 + chunk] = False
        sub = p.annot[mask]
        per_snp_bin = np.array([s[b] / st.M[j, b] if st.M[j, b] != 0 else 0 for b in range(ne)])
        h = []
        for e in range(ne):
            idx = np.where(sub[:, e] == 1)[0]
            tot_e = 0
            for si in idx:
                inner = 0
                for b in np.where(sub[si, :] == 1)[0].tolist():
                    inner += per_snp_bin[b]
                tot_e += inner
            h.append(tot_e / (total + s[-1]))
        h.append(total / (total + s[-1]))
        rows.append(h)
    return np.array(rows)


def enrichment(h2_all: np.ndarray, M: np.ndarray, ne: int) -> np.ndarray:
    """``(h2_e/h2_SNP)/(M_e/sum M)``  (base.py:788-825)."""
    rows = []
    for j, h in enumerate(h2_all):
        Mt = sum(M[j])
        rows.append([(h[e] / h[-1]) / (M[j][e] / Mt) if (Mt != 0 and M[j][e] != 0) else 0 for e in range(ne)])
    return np.array(rows)


def rhe_results(p: Problem, st: State, method: str = "QR") -> dict:
    """The result dict of ``RHE.run`` (pyrhe/src/models/rhe/rhe.py:28-101)."""
    sj, stot, _, _, Ts, qs = estimate(p, st, method)
    ne = p.n_est
    sig_all = np.vstack([sj, stot[None]])
    h2 = h2_nonoverlapping(sig_all, ne)
    enr = enrichment(h2, st.M, ne)
    h2o = h2_overlapping(p, st, sig_all)
    enro = enrichment(h2o, st.M, ne)
    jn = p.n_jack
    return {
        "sigma_ests_total": stot, "sig_errs": jackknife_se(sj, jn),
        "h2_total": h2[-1], "h2_errs": jackknife_se(h2[:-1], jn),
        "enrichment_total": enr[-1], "enrichment_errs": jackknife_se(enr[:-1], jn),
        "h2_total_overlap": h2o[-1], "h2_errs_overlap": jackknife_se(h2o[:-1], jn),
        "enrichment_total_overlap": enro[-1], "enrichment_errs_overlap": jackknife_se(enro[:-1], jn),
        "sigma_jack": sj, "T": Ts, "q": qs,
    }


def rhe_dom_results(p: Problem, st: State, method: str = "QR") -> dict:
    """The result dict of ``RHE_DOM.run`` (rhe_dom.py:76-117): 2 nbin estimates, non-overlapping h2 + enrichment only."""
    sj, stot, _, _, Ts, qs = estimate(p, st, method)
    ne, jn = p.n_est, p.n_jack
    h2 = h2_nonoverlapping(np.vstack([sj, stot[None]]), ne)
    enr = enrichment(h2, st.M, ne)
    return {"sigma_ests_total": stot, "sig_errs": jackknife_se(sj, jn), "h2_total": h2[-1],
            "h2_errs": jackknife_se(h2[:-1], jn), "enrichment_total": enr[-1],
            "enrichment_errs": jackknife_se(enr[:-1], jn), "sigma_jack": sj, "T": Ts, "q": qs}


# --------------------------------------------------------------------------- GENIE result shaping
def genie_h2(sig_adj_all: np.ndarray, nbin: int, n_gxe: int, n_nxe: int = 0) -> np.ndarray:
    """GENIE h2 rows (genie.py:146-189): per-bin g, per-bin gxe, nxe, total, total_g, [total_gxe]."""
    rows = []
    ne = nbin + n_gxe + n_nxe
    for s in sig_adj_all:
        tv = 0
        for e in range(ne):
            tv += s[e]
        den = tv + s[-1]
        g = [s[i] / den for i in range(nbin)]
        x = [s[nbin + i] / den for i in range(n_gxe)]
        nx = [s[nbin + n_gxe + i] / den for i in range(n_nxe)]
        gt = 0
        for v in g:
            gt += v
        xt = 0
        for v in x:
            xt += v
        nt = 0
        for v in nx:
            nt += v
        row = g + x + nx + [gt + xt + nt, gt]
        if n_gxe:
            row.append(xt)
        rows.append(row)
    return np.array(rows)


def genie_enrichment(h2_all: np.ndarray, M_last: np.ndarray, nbin: int) -> np.ndarray:
    """GENIE enrichment over genetic bins only, always with ``M[-1]`` (genie.py:191-219)."""
    rows = []
    for h in h2_all:
        th = 0
        ts = 0
        for i in range(nbin):
            th += h[i]
            ts += M_last[i]
        den = th / ts
        rows.append([(h[i] / M_last[i]) / den for i in range(nbin)])
    return np.array(rows)


def genie_results(p: Problem, st: State, method: str = "QR") -> dict:
    """The result dict of ``GENIE.run`` (genie.py:221-301)."""
    sj, stot, sja, stota, Ts, qs = estimate(p, st, method)
    jn, nbin = p.n_jack, p.nbin
    n_gxe = nbin if p.has_gxe else 0
    h2 = genie_h2(np.vstack([sja, stota[None]]), nbin, n_gxe, 1 if p.has_nxe else 0)
    enr = genie_enrichment(h2, st.M[-1], nbin)
    return {
        "sigma_ests_total": stot, "sig_errs": jackknife_se(sj, jn),
        "h2_total": h2[-1], "h2_errs": jackknife_se(h2[:-1], jn),
        "enrichment_total": enr[-1], "enrichment_errs": jackknife_se(enr[:-1], jn),
        "sigma_jack": sj, "T": Ts, "q": qs,
    }
