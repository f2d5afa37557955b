"""A CPU stand-in with the SketchEngine interface, built from ORACLE math, for the multi-process (gloo) tests of
the streaming driver's sharding / all-reduce / gather logic.  Test infrastructure only: the product never
constructs it (``pyrhe_amd`` has no CPU path)."""
import numpy as np
import torch

from oracle import plink
from oracle import rhe_oracle as O


class CpuStubEngine:
    def __init__(self, n_orig, n_keep):
        self.n_samples_orig = n_orig
        self.n_keep = n_keep
        self.pitch = (plink.row_bytes(n_orig) + 255) // 256 * 256
        self.n_pad = 4 * self.pitch
        self.sub_snps = 1 << 20

    def keep_vector(self, missing=()):
        k = torch.zeros(self.n_pad, dtype=torch.float32)
        k[: self.n_samples_orig] = 1
        if len(missing):
            k[list(missing)] = 0
        return k

    def set_rhs(self, sketch_cols, norm_cols, keep, cross_cols=None, cross_digits=3):
        self.sk = None if sketch_cols is None else sketch_cols.double().numpy()
        self.nm = None if norm_cols is None else norm_cols.double().numpy()
        self.cr = None if cross_cols is None else cross_cols.double().numpy()
        if self.cr is not None:
            # what the int8 first contraction sees of a cross column: `nd` balanced radix-256 digits of w * 126 / max|w|
            # (the engine's prep_rimg_i8_kernel) = w rounded to the nearest multiple of max|w| / (126 * 256^(nd-1)); three
            # digits are the float32 value itself
            nds = list(cross_digits) if isinstance(cross_digits, (list, tuple, np.ndarray)) else [cross_digits] * len(self.cr)
            assert len(nds) == len(self.cr) and all(a >= b for a, b in zip(nds, nds[1:])), "digit counts must not grow with the column index"
            self.cross_digits = [int(d) for d in nds]
            for j, nd in enumerate(self.cross_digits):
                mx = np.abs(self.cr[j]).max()
                if nd < 3 and mx > 0:
                    step = mx / (126.0 * 256.0 ** (nd - 1))
                    self.cr[j] = np.rint(self.cr[j] / step) * step
        self.n_cross = 0 if self.cr is None else self.cr.shape[0]
        self.keep = keep.numpy().astype(bool)
        self.n_sketch = 0 if self.sk is None else self.sk.shape[0]
        self.n_norm = 0 if self.nm is None else self.nm.shape[0]

    def block_stats(self, packed):
        codes = plink.decode_codes(packed.numpy(), self.n_pad)[:, self.keep]
        g = np.array([0, 0, 1, 2])[codes]
        return torch.from_numpy(g.sum(1).astype(np.int32)), torch.from_numpy((codes == 1).sum(1).astype(np.int32))

    def sketch_block(self, packed, entries, offsets, *, fill=None, out_sketch=None, out_norm=None, accumulate=False,
                     want_colsum=False, out_cross=None):
        nbin = len(offsets) - 1
        if out_sketch is None:
            out_sketch = torch.zeros((nbin, self.n_sketch, self.n_pad), dtype=torch.float32)
        if out_norm is None:
            out_norm = torch.zeros((nbin, max(self.n_norm, 1)), dtype=torch.float64)
        G = plink.decode_count_a2(packed.numpy(), self.n_pad).T.astype(np.float64)       # [m, n_pad]
        if fill is not None:
            f = fill.numpy()
            for r in range(G.shape[0]):
                G[r, np.isnan(G[r])] = f[r]
        ent = entries.numpy()
        for b in range(nbin):
            rows = ent[offsets[b]:offsets[b + 1]]
            sk = np.zeros((self.n_sketch, self.n_pad))
            nm = np.zeros(max(self.n_norm, 1))
            if len(rows):
                Gk = G[rows][:, self.keep].astype(np.float32)
                X = O.standardize_geno(np.asfortranarray(Gk.T)).astype(np.float64)          # [N, m] reference arithmetic
                if self.n_sketch:
                    sk[:, self.keep] = (X @ (X.T @ self.sk[:, self.keep].T)).T
                if self.n_norm:
                    nm[: self.n_norm] = ((X.T @ self.nm[:, self.keep].T) ** 2).sum(axis=0)
                if self.n_cross:
                    V = X.T @ self.cr[:, self.keep].T                                   # [m, c]
                    S = torch.from_numpy(V.T @ V)
                    out_cross[b] = out_cross[b] + S if accumulate else S
            elif self.n_cross and not accumulate:
                out_cross[b] = 0
            if accumulate:
                out_sketch[b] += torch.from_numpy(sk).float()
                out_norm[b] += torch.from_numpy(nm)
            else:
                out_sketch[b] = torch.from_numpy(sk).float()
                out_norm[b] = torch.from_numpy(nm)
        return out_sketch, out_norm, None

    def sketch_block_dom(self, packed, entries, offsets, *, fill=None, out_sketch=None, out_norm=None, out_sketch_dom=None,
                         out_norm_dom=None, accumulate=False, out_cross=None, out_cross_dom=None):
        """Additive + dominance sketches (RHE_DOM, rhe_dom.py:15-41) from oracle math: the additive half as sketch_block, the
        dominance half with the oracle's encoding and scaling."""
        self.sketch_block(packed, entries, offsets, fill=fill, out_sketch=out_sketch, out_norm=out_norm, accumulate=accumulate,
                          out_cross=out_cross)
        nbin = len(offsets) - 1
        G = plink.decode_count_a2(packed.numpy(), self.n_pad).T.astype(np.float64)
        if fill is not None:
            f = fill.numpy()
            for r in range(G.shape[0]):
                G[r, np.isnan(G[r])] = f[r]
        ent = entries.numpy()
        for b in range(nbin):
            rows = ent[offsets[b]:offsets[b + 1]]
            sk = np.zeros((self.n_sketch, self.n_pad))
            nm = np.zeros(max(self.n_norm, 1))
            S = None
            if len(rows):
                Gk = np.asfortranarray(G[rows][:, self.keep].astype(np.float32).T)          # [N, m]
                X = O.standardize_geno_dom(*O.encode_geno_dom(Gk)).astype(np.float64)
                if self.n_sketch:
                    sk[:, self.keep] = (X @ (X.T @ self.sk[:, self.keep].T)).T
                if self.n_norm:
                    nm[: self.n_norm] = ((X.T @ self.nm[:, self.keep].T) ** 2).sum(axis=0)
                if self.n_cross:
                    V = X.T @ self.cr[:, self.keep].T
                    S = torch.from_numpy(V.T @ V)
            if self.n_cross:
                if S is None:
                    S = torch.zeros((self.n_cross, self.n_cross), dtype=torch.float64)
                out_cross_dom[b] = out_cross_dom[b] + S if accumulate else S
            if accumulate:
                out_sketch_dom[b] += torch.from_numpy(sk).float()
                out_norm_dom[b] += torch.from_numpy(nm)
            else:
                out_sketch_dom[b] = torch.from_numpy(sk).float()
                out_norm_dom[b] = torch.from_numpy(nm)
        return out_sketch, out_norm, out_sketch_dom, out_norm_dom

    def loo_dots(self, x, y, period, n_sub, n_batch=1):
        """Oracle-side twin of pyrhe_engine_loo_dots (float64 torch on the CPU)."""
        L = period * n_sub

        def mat(rows):
            out = torch.zeros((n_batch, len(rows), L), dtype=torch.float64)
            for a, r in enumerate(rows):
                if r.tot is not None:
                    out[:, a] += r.tot.reshape(-1)[:L]
                if r.blk is not None:
                    b = r.blk.as_strided((n_batch, L), (r.stride, 1), r.blk.storage_offset())
                    if r.mul is not None:
                        b = (b.reshape(n_batch, n_sub, period) * r.mul).reshape(n_batch, L)
                    out[:, a] -= b.double()
            return out
        return torch.einsum("zal,zbl->zab", mat(x), mat(y))

    def sum_blocks_both(self, blocks, weights):
        return self.sum_blocks(blocks), self.sum_blocks(blocks, weights)

    def sum_blocks(self, blocks, weights=None):
        """Oracle-side twin of pyrhe_sum_blocks[_weighted]."""
        if not blocks.shape[0]:
            return torch.zeros(blocks.shape[1:], dtype=torch.float64)
        b64 = blocks.to(torch.float64)
        if weights is not None:
            b64 = b64 * weights.reshape((-1,) + (1,) * (blocks.dim() - 1))
        return b64.sum(dim=0)
