"""Host-side input parsers with the reference's semantics (``pyrhe/src/util/file_processing.py``).

Small text files only; the genotype ``.bed`` body is never decoded on the host -- its packed rows go
to the GPU as they are (``read_bed_packed``).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import pandas as pd

BED_MAGIC = bytes([0x6C, 0x1B, 0x01])


def count_bim(path: str) -> int:
    """Number of SNPs = number of lines of the .bim (file_processing.py:6-23)."""
    n = 0
    with open(path, "r") as f:
        for _ in f:
            n += 1
    return n


def read_fam(path: str) -> Tuple[int, pd.DataFrame]:
    """Individuals = rows of the whitespace table (file_processing.py:25-35)."""
    df = pd.read_csv(path, sep=r"\s+", header=None)
    return df.shape[0], df


def read_bed_packed(path: str, n_indv: int, n_snp: int) -> np.ndarray:
    """Packed SNP-major rows ``[n_snp, ceil(n_indv/4)]`` uint8 (memory-mapped; no decode)."""
    rb = (n_indv + 3) // 4
    with open(path, "rb") as f:
        magic = f.read(3)
    if magic != BED_MAGIC:
        raise ValueError(f"{path}: not a SNP-major PLINK1 .bed file (magic {magic.hex()})")
    mm = np.memmap(path, dtype=np.uint8, mode="r", offset=3)
    if mm.size < rb * n_snp:
        raise ValueError(f"{path}: truncated: {mm.size} bytes for {n_snp} SNPs x {rb} bytes")
    return mm[: rb * n_snp].reshape(n_snp, rb)


def read_annot(path: str) -> Tuple[int, np.ndarray, np.ndarray]:
    """M rows of space-separated 0/1, one column per bin -> (nbin, matrix [M, nbin], SNPs per bin)
    (file_processing.py:37-69; '#' lines skipped, counts are of entries equal to 1).  A file in the plain grammar -- one
    integer per bin, single spaces -- is parsed by pandas' C reader (1 M SNPs x 8 bins: 0.6 s instead of 2.2 s); anything it
    does not take as such goes through the reference's own line loop below, so errors surface as the reference's would."""
    try:
        df = pd.read_csv(path, sep=" ", header=None, comment="#", dtype=np.int64, skip_blank_lines=False)
        mat = np.ascontiguousarray(df.values)
        if mat.ndim == 2 and mat.shape[1] >= 1:
            return mat.shape[1], mat, (mat == 1).sum(axis=0).astype(int)
    except Exception:
        pass
    return _read_annot_lines(path)


def _read_annot_lines(path: str) -> Tuple[int, np.ndarray, np.ndarray]:
    rows = []
    with open(path, "r") as f:
        for line in f:
            if line.startswith("#"):
                continue
            rows.append([int(tok) for tok in line.strip().split(" ")])
    mat = np.array(rows)
    nbin = mat.shape[1]
    len_bin = (mat == 1).sum(axis=0).astype(int)
    return nbin, mat, len_bin


def read_pheno(path: str) -> Tuple[np.ndarray, List[int], bool]:
    """Header ``FID IID name...``; a row with ``NA`` or ``-9`` in any trait marks the individual missing
    (file_processing.py:72-100).  Returns (y [N, traits], missing row indices, all values in {0,1,2})."""
    with open(path, "r") as f:
        lines = f.readlines()
    n_traits = len(lines[0].strip().split()) - 2
    y, missing = [], []
    binary = True
    for i, line in enumerate(lines[1:]):
        cols = line.strip().split()[2:]
        vals = [-9.0 if c == "NA" else float(c) for c in cols]
        if "NA" in cols or -9 in vals:
            y.append([-9] * n_traits)
            missing.append(i)
        else:
            y.append(vals)
            if not all(v in (0, 1, 2) for v in vals):
                binary = False
    return np.array(y), missing, binary


def write_random_annot(path: str, n_snp: int, n_bin: int) -> None:
    """One random bin per SNP from NumPy's legacy global stream (file_processing.py:102-111)."""
    with open(path, "w") as f:
        for _ in range(n_snp):
            row = [0] * n_bin
            row[np.random.randint(0, n_bin)] = 1
            f.write(" ".join(str(v) for v in row) + "\n")


def read_cov(path: str, missing_indvs: Optional[List[int]] = None, impute: str = "ignore") -> Tuple[np.ndarray, List[int]]:
    """Covariates ``FID IID c...`` (file_processing.py:121-195): rows of phenotype-missing individuals are
    dropped first; rows containing NA/-9 are dropped ("ignore") or mean-imputed ("mean").  Parsed with
    pandas exactly as the reference does, so the float values (and the Fortran layout of ``.values``) match."""
    df = pd.read_csv(path, sep=r"\s+")
    missing_indvs = list(missing_indvs or [])
    if missing_indvs:
        df = df.drop(index=missing_indvs, errors="ignore")
    for col in ("FID", "IID"):
        if col in df.columns:
            df.drop(col, axis=1, inplace=True)
    is_missing = df.replace("NA", np.nan).isin([np.nan, -9]).any(axis=1)
    newly = df.index[is_missing].tolist()
    if impute == "ignore":
        df = df[~is_missing]
    else:
        df = df.replace({"NA": np.nan, "-9": np.nan})
        for col in df.columns:
            df[col] = df[col].fillna(df[col].mean())
    return df.values, missing_indvs + newly


def read_env(path: str) -> Tuple[int, np.ndarray]:
    """Environment file: the column literally named ``env``; count = columns - 2 (file_processing.py:212-220)."""
    df = pd.read_csv(path, sep=r"\s+")
    return len(df.columns) - 2, df["env"].to_numpy()
