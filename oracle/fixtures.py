"""Tiny synthetic PLINK cohorts used as parity fixtures (oracle side, TEST INFRASTRUCTURE).

Nothing here comes from the reference tree: the cohorts are generated from seeds and
written in the public file formats the reference parses
(``/root/reference/pyrhe/src/util/file_processing.py:6-227``):
``.bed/.bim/.fam`` (PLINK1), phenotype ``FID IID name...``, annotation = M rows of 0/1,
covariates ``FID IID c...``, environment ``FID IID env``.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import plink


@dataclass
class Cohort:
    name: str
    n_indv: int
    n_snp: int
    geno: np.ndarray                      # [M, N] int8 count of A2, -1 = missing call
    pheno: np.ndarray                     # [N, n_traits] float64; NaN = "NA" in file
    annot: np.ndarray                     # [M, nbin] 0/1
    cov: Optional[np.ndarray] = None      # [N, c] float64
    env: Optional[np.ndarray] = None      # [N] float64
    packed: np.ndarray = field(default=None)  # [M, ceil(N/4)] uint8

    def __post_init__(self):
        if self.packed is None:
            self.packed = plink.encode_bed_rows(self.geno)


def make_cohort(name: str, n_indv: int, n_snp: int, *, seed: int, nbin: int = 1,
                overlap: bool = False, with_cov: int = 0, with_env: bool = False,
                missing_geno: float = 0.0, missing_pheno: int = 0, h2: float = 0.3,
                n_traits: int = 1) -> Cohort:
    rng = np.random.default_rng(seed)
    maf = 0.05 + 0.45 * rng.random(n_snp)
    geno = np.empty((n_snp, n_indv), np.int8)
    for s in range(n_snp):
        while True:
            g = (rng.random(n_indv) < maf[s]).astype(np.int8) + (rng.random(n_indv) < maf[s]).astype(np.int8)
            tot = int(g.sum())
            if 0 < tot < 2 * n_indv:      # no monomorphic SNPs (reference has no guard)
                break
        geno[s] = g
    # phenotype from the complete genotypes
    mu = geno.mean(axis=1, keepdims=True)
    sd = np.sqrt(mu * (1 - mu / 2))
    X = ((geno - mu) / sd).T              # [N, M]
    pheno = np.empty((n_indv, n_traits))
    for t in range(n_traits):
        beta = rng.normal(0, np.sqrt(h2 / n_snp), n_snp)
        pheno[:, t] = X @ beta + rng.normal(0, np.sqrt(1 - h2), n_indv)
    # annotation
    annot = np.zeros((n_snp, nbin), np.int64)
    annot[np.arange(n_snp), rng.integers(0, nbin, n_snp)] = 1
    if overlap and nbin > 1:
        annot[3, :] = 0                   # one SNP in no bin
        annot[5, 0] = 1; annot[5, 1] = 1  # one SNP in two bins
        annot[n_snp - 2, nbin - 1] = 1; annot[n_snp - 2, 0] = 1
    cov = None
    if with_cov:
        cov = np.concatenate([np.ones((n_indv, 1)), rng.normal(size=(n_indv, with_cov - 1))], axis=1)
        pheno = pheno + (cov[:, 1:] @ rng.normal(0, 0.3, (with_cov - 1, 1)))
    env = None
    if with_env:
        env = (rng.random(n_indv) < 0.5).astype(np.float64)
    if missing_geno > 0:
        mask = rng.random(geno.shape) < missing_geno
        geno = geno.copy()
        geno[mask] = -1
    if missing_pheno:
        idx = rng.choice(n_indv, missing_pheno, replace=False)
        pheno = pheno.copy()
        pheno[idx, 0] = np.nan
    return Cohort(name, n_indv, n_snp, geno, pheno, annot, cov, env)


def write_cohort_files(c: Cohort, out_dir: str) -> dict:
    """Write the cohort in the reference's input formats. Returns a dict of paths."""
    os.makedirs(out_dir, exist_ok=True)
    prefix = os.path.join(out_dir, c.name)
    plink.write_bed(prefix + ".bed", c.packed)
    with open(prefix + ".fam", "w") as f:
        for i in range(c.n_indv):
            f.write(f"{i} {i} 0 0 0 -9\n")
    with open(prefix + ".bim", "w") as f:
        for s in range(c.n_snp):
            f.write(f"1\trs{s}\t0\t{s}\tA\tG\n")
    paths = {"geno": prefix}
    pheno_path = prefix + ".pheno"
    with open(pheno_path, "w") as f:
        names = " ".join(f"pheno{t}" for t in range(c.pheno.shape[1]))
        f.write(f"FID IID {names}\n")
        for i in range(c.n_indv):
            vals = " ".join("NA" if np.isnan(v) else repr(float(v)) for v in c.pheno[i])
            f.write(f"{i} {i} {vals}\n")
    paths["pheno"] = pheno_path
    annot_path = prefix + ".annot"
    with open(annot_path, "w") as f:
        for s in range(c.n_snp):
            f.write(" ".join(str(int(v)) for v in c.annot[s]) + "\n")
    paths["annot"] = annot_path
    if c.cov is not None:
        cov_path = prefix + ".cov"
        with open(cov_path, "w") as f:
            f.write("FID IID " + " ".join(f"cov{j}" for j in range(c.cov.shape[1])) + "\n")
            for i in range(c.n_indv):
                f.write(f"{i} {i} " + " ".join(repr(float(v)) for v in c.cov[i]) + "\n")
        paths["cov"] = cov_path
    if c.env is not None:
        env_path = prefix + ".env"
        with open(env_path, "w") as f:
            f.write("FID IID env\n")
            for i in range(c.n_indv):
                f.write(f"{i} {i} {repr(float(c.env[i]))}\n")
        paths["env"] = env_path
    return paths


def cohort_to_npz_dict(c: Cohort) -> dict:
    d = {"in_name": np.array(c.name), "in_n_indv": np.array(c.n_indv), "in_n_snp": np.array(c.n_snp),
         "in_packed": c.packed, "in_pheno": c.pheno, "in_annot": c.annot.astype(np.int8)}
    if c.cov is not None:
        d["in_cov"] = c.cov
    if c.env is not None:
        d["in_env"] = c.env
    return d


def cohort_from_npz(z) -> Cohort:
    n, m = int(z["in_n_indv"]), int(z["in_n_snp"])
    packed = z["in_packed"]
    codes = plink.decode_codes(packed, n)
    geno = np.array([0, -1, 1, 2], np.int8)[codes]
    return Cohort(str(z["in_name"]), n, m, geno, z["in_pheno"], z["in_annot"].astype(np.int64),
                  z["in_cov"] if "in_cov" in z else None, z["in_env"] if "in_env" in z else None,
                  packed)
