"""PLINK1 SNP-major ``.bed`` encode/decode in NumPy (oracle side).

TEST INFRASTRUCTURE -- not imported by the product (``pyrhe_amd``).

What it restates
----------------
The reference decodes genotypes with the third-party ``bed_reader==1.0.2``
(``/root/reference/pyrhe/setup.py:13``; call sites ``pyrhe/src/base/base.py:100,341,343``)
and then swaps 0<->2 (``base.py:347-355``).  ``bed_reader`` is not in the reference
tree, so this file restates its *published* behaviour for the call the reference
makes (``open_bed(path).read(index=np.s_[::1, a:b])`` with defaults):

* file = 3 magic bytes ``6c 1b 01`` (SNP-major), then per SNP ``ceil(N/4)`` bytes;
* sample ``i`` of a SNP sits in byte ``i // 4`` at bits ``2*(i%4) .. 2*(i%4)+1`` (LSB first);
* 2-bit code -> count of allele 1 (``count_A1=True``): ``0b00 -> 2``, ``0b10 -> 1``,
  ``0b11 -> 0``, ``0b01 -> NaN``;  dtype float32, Fortran order, shape ``[N, m]``.

After the reference's flip the value the maths sees is the count of allele 2:
``code 0 -> 0, code 2 -> 1, code 3 -> 2, code 1 -> missing``  (SURVEY.md Q7).
"""
from __future__ import annotations

import numpy as np

BED_MAGIC = bytes([0x6C, 0x1B, 0x01])

# value seen by the maths (after bed_reader count_A1 + the reference's 0<->2 flip)
CODE_TO_A2 = np.array([0.0, np.nan, 1.0, 2.0], dtype=np.float32)
# what bed_reader itself returns (count_A1=True)
CODE_TO_A1 = np.array([2.0, np.nan, 1.0, 0.0], dtype=np.float32)
# genotype value (count of A2) -> PLINK code
A2_TO_CODE = np.array([0, 2, 3], dtype=np.uint8)
MISSING_CODE = 1


def row_bytes(n_indv: int) -> int:
    return (n_indv + 3) // 4


def encode_bed_rows(geno_a2: np.ndarray) -> np.ndarray:
    """``geno_a2`` int array [m, N] with values 0/1/2 or -1 (missing) -> packed [m, ceil(N/4)] u8."""
    g = np.asarray(geno_a2)
    m, n = g.shape
    codes = np.where(g < 0, MISSING_CODE, A2_TO_CODE[np.clip(g, 0, 2)]).astype(np.uint8)
    pad = (-n) % 4
    if pad:
        codes = np.concatenate([codes, np.zeros((m, pad), np.uint8)], axis=1)
    c = codes.reshape(m, -1, 4)
    return (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).astype(np.uint8)


def write_bed(path: str, packed: np.ndarray) -> None:
    with open(path, "wb") as f:
        f.write(BED_MAGIC)
        f.write(np.ascontiguousarray(packed, dtype=np.uint8).tobytes())


def read_bed_packed(path: str, n_indv: int, n_snp: int) -> np.ndarray:
    rb = row_bytes(n_indv)
    with open(path, "rb") as f:
        magic = f.read(3)
        if magic != BED_MAGIC:
            raise ValueError(f"{path}: not a SNP-major PLINK1 .bed (magic {magic.hex()})")
        buf = np.frombuffer(f.read(rb * n_snp), dtype=np.uint8)
    if buf.size != rb * n_snp:
        raise ValueError(f"{path}: truncated ({buf.size} bytes, expected {rb * n_snp})")
    return buf.reshape(n_snp, rb)


def decode_codes(packed_rows: np.ndarray, n_indv: int) -> np.ndarray:
    """packed [m, rb] u8 -> 2-bit codes [m, N] u8."""
    p = np.asarray(packed_rows, dtype=np.uint8)
    m = p.shape[0]
    codes = np.empty((m, p.shape[1], 4), np.uint8)
    for k in range(4):
        codes[:, :, k] = (p >> (2 * k)) & 3
    return codes.reshape(m, -1)[:, :n_indv]


def decode_count_a1(packed_rows: np.ndarray, n_indv: int) -> np.ndarray:
    """What ``open_bed.read`` returns for these SNPs: [N, m] float32 Fortran order, NaN = missing."""
    codes = decode_codes(packed_rows, n_indv)
    return np.asfortranarray(CODE_TO_A1[codes].T)


def decode_count_a2(packed_rows: np.ndarray, n_indv: int) -> np.ndarray:
    """decode + the reference's 0<->2 flip (base.py:347-355): [N, m] float32 F-order."""
    codes = decode_codes(packed_rows, n_indv)
    return np.asfortranarray(CODE_TO_A2[codes].T)
