"""Shared helpers for the parity tests (load goldens, build oracle problems)."""
import os

import numpy as np

from oracle import fixtures
from oracle import rhe_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

RHE_CASES = ["rhe_1bin_nocov", "rhe_3bin_nocov_rem", "rhe_3bin_cov", "rhe_1bin_missing",
             "rhe_1bin_missing_mean", "rhe_1bin_n1024", "rhe_2bin_jn1", "rhe_1bin_2traits", "rhe_2bin_cov_k35"]
STREAMING_CASES = ["srhe_3bin_cov", "srhe_2bin_missing_cov"]
DOM_CASES = ["rhe_dom_2bin", "rhe_dom_cov"]
GENIE_CASES = ["genie_gxe_2bin", "genie_gxe_cov", "genie_gxe_nxe_2bin", "genie_gxe_nxe_cov", "genie_gxe_k33"]
RESULT_KEYS_RHE = ["sigma_ests_total", "sig_errs", "h2_total", "h2_errs", "enrichment_total", "enrichment_errs",
                   "h2_total_overlap", "h2_errs_overlap", "enrichment_total_overlap", "enrichment_errs_overlap"]
RESULT_KEYS_GENIE = RESULT_KEYS_RHE[:6]


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def oracle_problem(z, trait=0):
    """Build the oracle's Problem from a golden file (inputs only + the reference's parsed covariates)."""
    c = fixtures.cohort_from_npz(z)
    miss = [int(i) for i in z["ref_missing_indv"]]
    ph = np.delete(c.pheno.copy(), miss, axis=0)
    ph = ph - np.mean(ph, axis=0)                      # base.py:153-154
    cov = z["ref_cov_matrix"] if "ref_cov_matrix" in z else None
    p = O.Problem(packed=c.packed, n_indv_orig=c.n_indv, annot=c.annot, pheno=ph[:, trait].reshape(-1, 1),
                  n_jack=int(z["param_num_jack"]), k=int(z["param_num_random_vec"]), seed=int(z["param_seed"]),
                  missing_indv=miss, cov=cov,
                  impute=str(z["param_geno_impute_method"]) if "param_geno_impute_method" in z else "binary")
    if str(z["param_model"]) in ("RHE_DOM", "StreamingRHE_DOM"):
        p.dom = True
    if "param_genie_model" in z:
        p.genie_model = str(z["param_genie_model"])
        p.env = c.env.reshape(-1, 1)
    p.Z = O.make_probe_vectors(p.seed, p.n_indv, p.k)
    return p, c


def oracle_results(p, streaming=False):
    """The oracle's result dict for a golden's problem, by model family."""
    st = O.precompute(p, streaming_compat=streaming)
    fn = O.genie_results if p.genie_model else (O.rhe_dom_results if p.dom else O.rhe_results)
    return fn(p, st)


def accumulation_noise_floor(z, keys, streaming=False):
    """How far the REFERENCE's own results move when only the accumulation order of its float32 products changes: the oracle
    in its reference order (torch float32 ``@``: bit-equal to the goldens) against the same float32 operands summed in
    float64 and rounded once (``O.ACCUM = "exact"``).  Returned per result key, relative to the key's largest entry.  A
    product result may differ from the golden by a small multiple of this and still be the same computation; GENIE's
    normal equations amplify it 100x more than RHE's (measured: sigma^2 4e-6 vs 2e-8 of the largest component)."""
    p, _ = oracle_problem(z)
    out = {}
    try:
        ref = oracle_results(p, streaming)
        O.ACCUM = "exact"
        alt = oracle_results(p, streaming)
    finally:
        O.ACCUM = "reference"
    for k in keys:
        a, b = np.asarray(ref[k], float), np.asarray(alt[k], float)
        out[k] = float(np.abs(a - b).max() / max(np.abs(a).max(), 1e-3))
    return out
