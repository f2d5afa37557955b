"""Host-side estimator: consumes the tiny ``(T, q)`` systems the GPU assembles and reproduces the
reference's solve / jackknife / heritability arithmetic (``pyrhe/src/base/base.py:298-312,630-825``,
``pyrhe/src/models/genie/genie.py:146-219``).  O(ne^3) per jackknife sample -- not on the hot path."""
from __future__ import annotations

import numpy as np
import scipy.linalg


def solve_normal_equations(T: np.ndarray, q: np.ndarray, method: str = "QR") -> np.ndarray:
    """``T sigma = q`` by full QR + back-substitution (default, base.py:306-312) or lstsq (base.py:298-303)."""
    if method == "QR":
        Qm, Rm = scipy.linalg.qr(T)
        return np.ravel(scipy.linalg.solve_triangular(Rm, np.dot(Qm.T, q)))
    if method == "lstsq":
        return np.ravel(np.linalg.lstsq(T, q, rcond=None)[0])
    raise ValueError("Unsupported method for solving linear equation")


_controller = None


def single_threaded_blas():
    """Context manager pinning the host BLAS/LAPACK/OpenMP pools to one thread.  The host arithmetic of this package is
    tiny ((ne+1)^2 solves, [blocks, ne, ne] scalings, one dot product of length N); on a 256-core host every multi-threaded
    BLAS call that finds its pool asleep pays 50-100 ms to wake it (measured: pass 2 at 31 ms vs 100 ms)."""
    global _controller
    try:
        from threadpoolctl import ThreadpoolController
    except ImportError:
        import contextlib
        return contextlib.nullcontext()
    if _controller is None:
        _controller = ThreadpoolController()
    return _controller.limit(limits=1)


def solve_all(T: np.ndarray, q: np.ndarray, method: str = "QR") -> np.ndarray:
    """The jn+1 solves of one trait, ``T [jn+1, ne+1, ne+1]``, ``q [jn+1, ne+1]`` -> ``[jn+1, ne+1]``.  QR (the reference's
    default, base.py:306-312): ONE stacked Householder factorisation (LAPACK geqrf/orgqr per system, like scipy's) and a
    back-substitution vectorised over the systems -- 0.2 ms for 101 systems instead of 1.6 ms of Python-level calls, which
    is a quarter of the end-to-end time of a 50k-sample cohort.  A singular or non-finite system falls back to the
    per-system path, so errors surface exactly as the reference's would.  The BLAS/LAPACK thread pool is pinned to one
    thread (on a 256-core host waking the pool costs tens of milliseconds, the solves themselves microseconds)."""
    T = np.asarray(T, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64).reshape(T.shape[0], -1)
    with single_threaded_blas():
        if method == "QR" and T.ndim == 3 and T.shape[0] > 1 and np.all(np.isfinite(T)) and np.all(np.isfinite(q)):
            try:
                Qm, Rm = np.linalg.qr(T)
                diag = np.diagonal(Rm, axis1=1, axis2=2)
                if np.all(diag != 0.0):
                    rhs = np.einsum("bji,bj->bi", Qm, q)
                    n = T.shape[1]
                    x = np.zeros_like(rhs)
                    for i in range(n - 1, -1, -1):
                        x[:, i] = (rhs[:, i] - np.einsum("bj,bj->b", Rm[:, i, i + 1:], x[:, i + 1:])) / diag[:, i]
                    if np.all(np.isfinite(x)):
                        return x
            except np.linalg.LinAlgError:
                pass
        return np.array([solve_normal_equations(T[j], q[j].reshape(-1, 1), method) for j in range(T.shape[0])])


def jackknife_se(ests: np.ndarray, n_jack: int) -> list:
    """Delete-one-block jackknife SE per column: sqrt((jn-1)/jn * sum_j (theta_j - mean)^2)  (base.py:680-703)."""
    ests = np.asarray(ests)
    out = []
    for col in range(ests.shape[1]):
        vals = ests[:, col]
        centre = np.mean(vals)
        ss = 0
        for v in vals:
            ss += (v - centre) ** 2
        out.append(np.sqrt((n_jack - 1) * ss / n_jack))
    return out


def _stack(jack: np.ndarray, total: np.ndarray) -> np.ndarray:
    return np.vstack([jack, total[np.newaxis, :]])


def h2_nonoverlapping(sigma_jack: np.ndarray, sigma_total: np.ndarray, n_est: int):
    """h2_e = sigma_e / (sum_g sigma + sigma_err); last column = SNP heritability (base.py:705-742)."""
    rows = []
    for s in _stack(sigma_jack, sigma_total):
        g = 0
        for e in range(n_est):
            g += s[e]
        denom = g + s[-1]
        rows.append([v / denom for v in s[:-1]] + [g / denom])
    rows = np.array(rows)
    return rows[:-1], rows[-1]


def h2_overlapping(sigma_jack: np.ndarray, sigma_total: np.ndarray, n_est: int, annot: np.ndarray,
                   M: np.ndarray, n_snp: int, n_jack: int):
    """Overlapping-annotation definition (base.py:744-785): every SNP of bin e contributes
    sum_{bins b containing it} sigma_b / M[j][b].  Row j uses the SNPs outside block j computed with the
    reference's own index arithmetic (base.py:382-393), which for the total row j = jn masks
    [jn*step, jn*step + step + M%jn) -- i.e. the trailing M%jn SNPs -- and nothing when M%jn == 0."""
    step, rem = n_snp // n_jack, n_snp % n_jack
    # The reference walks every SNP of every bin in Python (M x bins x (jn+1) iterations).  Same sums through the
    # co-membership counts C[e, b] = #{SNPs in bins e and b}: row e of the definition is sum_b C_j[e, b] sigma_b / M[j][b]
    # with C_j counted outside the masked range of row j -- exact integers, O(M nb^2) once plus O(step nb^2) per row.
    member = (np.asarray(annot)[:, :n_est] == 1).astype(np.float64)      # counts < 2^53: exact in float64, and BLAS-fast
    C_all = member.T @ member
    rows = []
    for j, s in enumerate(_stack(sigma_jack, sigma_total)):
        g = 0
        for e in range(n_est):
            g += s[e]
        denom = g + s[-1]
        chunk = step if j < n_jack - 1 else step + rem
        cut = member[j * step: j * step + chunk]                  # NumPy slicing clips at n_snp like the reference's mask
        C = C_all - cut.T @ cut
        per_bin = np.array([s[b] / M[j][b] if M[j][b] != 0 else 0 for b in range(n_est)], dtype=np.float64)
        row = list((C @ per_bin) / denom)
        row.append(g / denom)
        rows.append(row)
    rows = np.array(rows)
    return rows[:-1], rows[-1]


def enrichment(h2_jack: np.ndarray, h2_total: np.ndarray, M: np.ndarray, n_est: int):
    """(h2_e / h2_SNP) / (M_e / sum M)  per jackknife row (base.py:788-825)."""
    rows = []
    for j, h in enumerate(_stack(h2_jack, h2_total)):
        Mt = sum(M[j])
        rows.append([(h[e] / h[-1]) / (M[j][e] / Mt) if (Mt != 0 and M[j][e] != 0) else 0 for e in range(n_est)])
    rows = np.array(rows)
    return rows[:-1], rows[-1]


# ----------------------------------------------------------------------------- GENIE shaping
def genie_h2(sig_adj_jack: np.ndarray, sig_adj_total: np.ndarray, n_bin: int, n_gxe: int, n_nxe: int = 0):
    """GENIE heritabilities from trace-adjusted variance components (genie.py:146-189):
    per-bin g, per-bin gxe, (nxe), total, total_g, [total_gxe]."""
    ne = n_bin + n_gxe + n_nxe
    rows = []
    for s in _stack(sig_adj_jack, sig_adj_total):
        tv = 0
        for e in range(ne):
            tv += s[e]
        denom = tv + s[-1]
        row, g_tot, x_tot, n_tot = [], 0, 0, 0
        for i in range(n_bin):
            v = s[i] / denom
            row.append(v)
            g_tot += v
        for i in range(n_gxe):
            v = s[n_bin + i] / denom
            row.append(v)
            x_tot += v
        for i in range(n_nxe):
            v = s[n_bin + n_gxe + i] / denom
            row.append(v)
            n_tot += v
        row.append(g_tot + x_tot + n_tot)
        row.append(g_tot)
        if n_gxe:
            row.append(x_tot)
        rows.append(row)
    rows = np.array(rows)
    return rows[:-1], rows[-1]


def genie_enrichment(h2_jack: np.ndarray, h2_total: np.ndarray, M_last: np.ndarray, n_bin: int):
    """Enrichment over the genetic bins with the whole-data SNP counts (genie.py:191-219)."""
    rows = []
    for h in _stack(h2_jack, h2_total):
        h_sum, m_sum = 0, 0
        for i in range(n_bin):
            h_sum += h[i]
            m_sum += M_last[i]
        per_snp = h_sum / m_sum
        rows.append([(h[i] / M_last[i]) / per_snp for i in range(n_bin)])
    rows = np.array(rows)
    return rows[:-1], rows[-1]


def liability_scale(h2, se, pop_prev: float, samp_prev: float):
    """Observed-scale -> liability-scale h2 for case/control traits (base.py:857-868)."""
    from scipy.stats import chi2, norm
    K, P = pop_prev, samp_prev
    zv = norm.pdf(norm.ppf(K))
    factor = K ** 2 * (1 - K) ** 2 / P / (1 - P) / zv ** 2
    h2_l = h2 * factor
    var_l = (se * factor) ** 2
    return h2_l, var_l ** 0.5, chi2.sf(h2_l ** 2 / var_l, 1)
