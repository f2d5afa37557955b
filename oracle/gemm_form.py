"""NumPy emulation of the DEVICE algorithm (TEST INFRASTRUCTURE, oracle side).

The HIP path does not materialise the standardized matrix X.  With G the raw genotype counts
(0/1/2, exact in any float format), mu_s the SNP mean and d_s = 1/sqrt(mu_s (1 - mu_s/2)):

    X^T R  = D (G^T R - mu (1^T R))                 "GEMM 1" + rank-1 correction
    X  V   = G (D V) - 1 (mu^T D V)                 "GEMM 2" + rank-1 correction

This module mirrors that formulation including the operand rounding of each precision mode
(``f16``: R and W=DV rounded once to fp16; ``f16x2``: hi+lo fp16 split, ~22 significant bits;
``f32``: no rounding) with float32 accumulation, so tests can bound the parity each mode can reach
against the reference-order oracle *before* the GPU is involved, and so a failing GPU test can be
bisected into "algorithm" vs "kernel".  It restates the maths of
``/root/reference/pyrhe/src/base/base.py:291-296,403-417`` in GEMM form.
"""
from __future__ import annotations

import numpy as np


def pow2_scale(max_abs: np.ndarray, target: float = 16384.0) -> np.ndarray:
    """Per-column power-of-two scale s with max|col| * s <= target (fp16 max is 65504)."""
    m = np.where(max_abs > 0, max_abs, 1.0)
    return 2.0 ** np.floor(np.log2(target / m))


def split_f16(x: np.ndarray, mode: str):
    """Return the list of fp16 operand parts (as float32 arrays of fp16-representable values)."""
    x = np.asarray(x, np.float32)
    if mode == "f32":
        return [x]
    hi = x.astype(np.float16).astype(np.float32)
    if mode == "f16":
        return [hi]
    if mode == "f16x2":
        lo = (x - hi).astype(np.float16).astype(np.float32)
        return [hi, lo]
    raise ValueError(mode)


def snp_stats(G: np.ndarray, n_indv: int):
    """mu, d in float32 with the reference's operation order (base.py:292-293)."""
    colsum = G.sum(axis=1).astype(np.float32)                 # exact integer < 2^24
    mu = colsum / np.float32(n_indv)
    d = np.float32(1) / np.sqrt(mu * (np.float32(1) - np.float32(0.5) * mu))
    return mu, d


def sketch_gemm_form(G: np.ndarray, R: np.ndarray, n_out: int, mode: str = "f16", mode2: str = None):
    """G [m, N] ints (missing already filled), R [N, B] float64 right-hand columns.

    Returns (V [m, B] float32 = X^T R,  A [N, n_out] float32 = X V[:, :n_out]).
    """
    m, N = G.shape
    G32 = G.astype(np.float32)
    mu, d = snp_stats(G, N)
    # ---- GEMM 1: P = G^T R with R rounded per mode, scaled per column into fp16 range
    sR = pow2_scale(np.abs(R).max(axis=0), 8192.0)
    parts = split_f16(R * sR, mode)
    P = np.zeros((m, R.shape[1]), np.float32)
    csum = np.zeros(R.shape[1], np.float64)
    for part in parts:
        P += G32 @ part                                       # float32 accumulate
        csum += part.astype(np.float64).sum(axis=0)
    invs = (1.0 / sR).astype(np.float32)
    V = (d[:, None] * (P - mu[:, None] * csum.astype(np.float32)[None, :])) * invs[None, :]
    # ---- GEMM 2: A = G W - 1 (mu^T W), W = D V
    W = d[:, None] * V[:, :n_out]
    sW = pow2_scale(np.abs(W).max(axis=0), 8192.0)
    wparts = split_f16(W * sW.astype(np.float32), mode2 or mode)
    A = np.zeros((N, n_out), np.float32)
    corr = np.zeros(n_out, np.float64)
    for part in wparts:
        A += G32.T @ part
        corr += (mu[:, None].astype(np.float64) * part.astype(np.float64)).sum(axis=0)
    A = (A - corr.astype(np.float32)[None, :]) * (1.0 / sW).astype(np.float32)[None, :]
    return V, A
