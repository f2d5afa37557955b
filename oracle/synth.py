"""CPU twin of the device synthetic-genotype generator (``synth_kernel`` in
``pyrhe_amd/csrc/sketch_engine.hip``).  TEST INFRASTRUCTURE.

SNP ``row`` has allele frequency ``p = thr/256``, ``thr = 13 + mix64(seed*K0 + row*K1) % 116``;
sample ``i`` of that row (dword ``w = i//16``, slot ``k = (i%16)//4``, ``t = i%4``) takes two bytes of
``mix64(seed*K0 + row*K1 + (w+1)*K2 + (k+1)*K0)`` and counts how many are ``< thr``.
"""
import numpy as np

K0 = np.uint64(0x9E3779B97F4A7C15)
K1 = np.uint64(0xD1B54A32D192ED03)
K2 = np.uint64(0x8CB92BA72F3D8DD7)


def mix64(x):
    x = np.asarray(x, np.uint64)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def synth_rows(seed: int, row0: int, n_rows: int, n_samples: int) -> np.ndarray:
    """Genotypes [n_rows, n_samples] int8 (count of allele 2), identical to the device generator."""
    out = np.empty((n_rows, n_samples), np.int8)
    n_words = (n_samples + 15) // 16
    w = np.arange(n_words, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for r in range(n_rows):
            row = np.uint64(row0 + r)
            rb = np.uint64(seed) * K0 + row * K1
            thr = np.uint64(13) + mix64(rb) % np.uint64(116)
            base = rb + (w + np.uint64(1)) * K2
            g = np.empty((n_words, 16), np.int8)
            for k in range(4):
                h = mix64(base + np.uint64(k + 1) * K0)
                for t in range(4):
                    a = (h >> np.uint64(16 * t)) & np.uint64(0xFF)
                    b = (h >> np.uint64(16 * t + 8)) & np.uint64(0xFF)
                    g[:, k * 4 + t] = (a < thr).astype(np.int8) + (b < thr).astype(np.int8)
            out[r] = g.reshape(-1)[:n_samples]
    return out
