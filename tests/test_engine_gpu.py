"""GPU parity of the HIP sketch engine (through the C ABI) against the oracle and the goldens.

Tolerances (written here, as the task requires):
  * decode, genotype sums, block indexing: bit-exact.
  * per-block sketches X(X^T z), ||X^T y||^2 vs the reference's float32 GEMV results:
      f16x2 mode: 5e-6 of the block's max |value|   (fp32 accumulation-order noise only)
      i8x3  mode: 5e-6 (int8-limb GEMM 1 with exact integer accumulation + hi/lo fp16 GEMM 2)
      f16   mode: 2e-3 of the block's max |value|   (one fp16 rounding of Z and of W = D V)
"""
import numpy as np
import pytest
import torch

from oracle import plink, synth
from oracle import rhe_oracle as O
from tests.helpers import load_golden, oracle_problem

pytestmark = pytest.mark.gpu

TOL = {"f16x2": 5e-6, "f16": 2e-3, "i8x3": 5e-6}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda", 0)


def test_decode_bit_exact(dev):
    from pyrhe_amd import engine
    rng = np.random.default_rng(0)
    n = 1003
    rows = rng.integers(0, 256, size=(37, plink.row_bytes(n)), dtype=np.uint8)
    d = engine.decode_rows(engine.upload_packed(rows, dev)).cpu().numpy()[:, :n]
    ref = plink.decode_count_a2(rows, n).T
    assert np.array_equal(np.isnan(d), np.isnan(ref))
    assert np.array_equal(np.nan_to_num(d, nan=-1), np.nan_to_num(ref, nan=-1))


def test_synth_matches_cpu_twin(dev):
    from pyrhe_amd import engine
    n, m = 1000, 9
    packed = engine.synth_genotypes(m, n, seed=20251212, row0=5, device=dev).cpu().numpy()
    codes = plink.decode_codes(packed, n)
    g = np.array([0, -1, 1, 2], np.int8)[codes]
    assert np.array_equal(g, synth.synth_rows(20251212, 5, m, n))
    assert not packed[:, plink.row_bytes(n):].any()          # padding bytes are zero


def _engine_block_sketches(p, c, dev, precision, fill_by_block=None, coding="plink"):
    """Run the engine block by block; returns XXz [ne, jn, k, N], yXXy [ne, jn], colsums list."""
    from pyrhe_amd import engine
    N0 = c.n_indv
    keep_idx = np.setdiff1d(np.arange(N0), np.array(p.missing_indv, dtype=np.int64))
    step = p.n_snp // p.n_jack
    eng = engine.SketchEngine(N0, p.n_indv, max_entries=step + p.n_snp % p.n_jack, precision=precision, device=dev)
    keep = eng.keep_vector(p.missing_indv)
    Zp = torch.zeros((p.k, eng.n_pad), dtype=torch.float32, device=dev)
    Zp[:, torch.as_tensor(keep_idx, device=dev)] = torch.from_numpy(p.Z.T.astype(np.float32)).to(dev)
    yp = torch.zeros((1, eng.n_pad), dtype=torch.float32, device=dev)
    yp[0, torch.as_tensor(keep_idx, device=dev)] = torch.from_numpy(p.pheno[:, 0].astype(np.float32)).to(dev)
    eng.set_rhs(Zp, yp, keep)
    packed = engine.upload_packed(c.packed, dev)
    if coding == "value":
        engine.recode_rows(packed, "value")
        eng.set_coding("value")
    XXz = np.zeros((p.nbin, p.n_jack, p.k, p.n_indv))
    yXXy = np.zeros((p.nbin, p.n_jack))
    colsums = []
    for j in range(p.n_jack):
        s, e = O.block_range(j, p.n_snp, p.n_jack)
        ann = p.annot[s:e]
        ent = [np.nonzero(ann[:, b])[0].astype(np.int32) for b in range(p.nbin)]
        offs = np.concatenate([[0], np.cumsum([len(x) for x in ent])])
        entries = torch.from_numpy(np.concatenate(ent).astype(np.int32)).to(dev)
        fill = None
        if fill_by_block is not None:
            fill = torch.from_numpy(fill_by_block[j].astype(np.float32)).to(dev)
        sk, nm, cs = eng.sketch_block(packed[s:e], entries, offs, fill=fill, want_colsum=True)
        torch.cuda.synchronize()
        XXz[:, j] = sk.cpu().numpy()[:, :, keep_idx]
        yXXy[:, j] = nm.cpu().numpy()[:, 0]
        colsums.append((cs.cpu().numpy(), ent, s))
        # dropped / padding samples are written as exact zeros
        dropped = np.setdiff1d(np.arange(eng.n_pad), keep_idx)
        assert not sk.cpu().numpy()[:, :, dropped].any()
    eng.close()
    return XXz, yXXy, colsums


@pytest.mark.parametrize("precision", ["f16x2", "f16", "i8x3"])
@pytest.mark.parametrize("name", ["rhe_1bin_nocov", "rhe_3bin_nocov_rem", "rhe_1bin_n1024"])
def test_block_sketch_vs_reference_golden(dev, name, precision):
    z = load_golden(name)
    p, c = oracle_problem(z)
    XXz, yXXy, colsums = _engine_block_sketches(p, c, dev, precision)
    ref = z["ref_t0_XXz_blocks"].astype(np.float64)          # [ne, jn, k, N] float32 values of the reference
    refy = z["ref_t0_yXXy_blocks"]
    tol = TOL[precision]
    for j in range(p.n_jack):
        for b in range(p.nbin):
            scale = np.abs(ref[b, j]).max()
            if scale == 0:
                assert not XXz[b, j].any()
                continue
            assert np.abs(XXz[b, j] - ref[b, j]).max() <= tol * scale, (j, b)
            assert abs(yXXy[b, j] - refy[b, j]) <= 2 * tol * abs(refy[b, j]), (j, b)
    # exact integer genotype sums per entry
    for cs, ent, s in colsums:
        exp = np.concatenate([c.geno[s + e_].astype(np.int64).sum(axis=1) for e_ in ent])
        assert np.array_equal(cs.astype(np.int64), exp)


@pytest.mark.parametrize("coding", ["plink", "value"])
@pytest.mark.parametrize("name", ["rhe_1bin_missing", "rhe_1bin_missing_mean"])
def test_block_sketch_with_missing_calls(dev, name, coding):
    """Missing genotype calls + dropped individuals; fill values replayed from NumPy's stream (base.py:265-289)."""
    z = load_golden(name)
    p, c = oracle_problem(z)
    fills = []
    for j in range(p.n_jack):
        s, e = O.block_range(j, p.n_snp, p.n_jack)
        X = O.read_geno(p.packed, p.n_indv_orig, s, e, p.missing_indv)
        miss = np.isnan(X)
        np.random.seed(p.seed)                               # non-streaming reseeds per block (base.py:510)
        Xi = O.impute_geno(X.copy(), p.impute)
        f = np.zeros(e - s)
        for m in range(e - s):
            if miss[:, m].any():
                f[m] = Xi[miss[:, m], m][0]
        fills.append(f)
    XXz, yXXy, _ = _engine_block_sketches(p, c, dev, "f16x2" if coding == "plink" else "i8x3", fill_by_block=fills, coding=coding)
    ref = z["ref_t0_XXz_blocks"].astype(np.float64)
    for j in range(p.n_jack):
        scale = np.abs(ref[0, j]).max()
        assert np.abs(XXz[0, j] - ref[0, j]).max() <= TOL["f16x2"] * scale, j
        assert abs(yXXy[0, j] - z["ref_t0_yXXy_blocks"][0, j]) <= 1e-5 * abs(z["ref_t0_yXXy_blocks"][0, j])


def test_block_stats_exact(dev):
    from pyrhe_amd import engine
    z = load_golden("rhe_1bin_missing")
    p, c = oracle_problem(z)
    eng = engine.SketchEngine(c.n_indv, p.n_indv, max_entries=64, precision="f16", device=dev)
    keep = eng.keep_vector(p.missing_indv)
    eng.set_rhs(torch.zeros((1, eng.n_pad), device=dev), None, keep)
    s, nm = eng.block_stats(engine.upload_packed(c.packed, dev))
    g = np.delete(c.geno.astype(np.int64), p.missing_indv, axis=1)
    assert np.array_equal(s.cpu().numpy(), np.where(g < 0, 0, g).sum(axis=1))
    assert np.array_equal(nm.cpu().numpy(), (g < 0).sum(axis=1))
    eng.close()


@pytest.mark.parametrize("precision,k,coding", [("i8x3", 10, "plink"), ("i8x3", 10, "value"), ("i8x3", 20, "plink"),
                                                ("i8x3", 20, "value"), ("i8x3", 7, "value"), ("i8x3", 13, "plink"),
                                                ("f16x2", 10, "plink"), ("f16", 10, "plink")])
@pytest.mark.parametrize("with_fill", [False, True])
def test_multibin_call_equals_one_call_per_bin(dev, precision, k, coding, with_fill):
    """Several bins in one call take the segmented flow (one GEMM 1 over the concatenated lists, per-bin scales,
    corrections, norms and GEMM 2 ranges by segment); one call per bin takes the single-list flow.  Same results,
    including: overlapping bins, an empty bin, bins shorter than a K block, lists cut by sub_snps chunks, accumulate.
    Under i8x3 the segmented flow runs the limb-packed int8 GEMM 2 (two A tiles up to 10 sketch columns, four up to 20,
    both genotype codings, with and without fill values) and the single-list flow the fp16 hi+lo GEMM 2: two independent
    kernels, one answer."""
    from pyrhe_amd import engine
    n, m = 5000, 900
    rng = np.random.default_rng(5)
    packed = engine.synth_genotypes(m, n, 77, 0, dev)
    fill = None
    if with_fill:
        # plant missing calls (code 01) in the first byte of some rows and give those SNPs a fill value
        rows = torch.as_tensor(rng.choice(m, 200, replace=False), device=dev)
        packed[rows, 3] = (packed[rows, 3] & 0xF0) | 0x05
        fill = torch.zeros(m, dtype=torch.float32, device=dev)
        fill[rows] = torch.as_tensor(rng.integers(0, 3, 200), dtype=torch.float32, device=dev)
    lists = [np.sort(rng.choice(m, sz, replace=False)).astype(np.int32) for sz in (311, 0, 17, 450, 64, 129)]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in lists])])
    entries = torch.from_numpy(np.concatenate(lists)).to(dev)
    if coding == "value":
        packed = engine.recode_rows(packed, "value")
    eng = engine.SketchEngine(n, n, max_entries=m, precision=precision, device=dev, sub_snps=256)
    eng.set_coding(coding)
    keep = eng.keep_vector([])
    Z = torch.zeros((k, eng.n_pad), device=dev)
    Z[:, :n] = torch.from_numpy(rng.standard_normal((k, n)).astype(np.float32)).to(dev)
    y = torch.zeros((2, eng.n_pad), device=dev)
    y[:, :n] = torch.from_numpy(rng.standard_normal((2, n)).astype(np.float32)).to(dev)
    eng.set_rhs(Z, y, keep)
    sk, nm, cs = eng.sketch_block(packed, entries, offs, fill=fill, want_colsum=True)
    sk2, nm2, _ = eng.sketch_block(packed, entries, offs, fill=fill, out_sketch=sk.clone(), out_norm=nm.clone(), accumulate=True)
    torch.cuda.synchronize()
    tol = {"i8x3": 2e-6, "f16x2": 2e-6, "f16": 2e-3}[precision]
    for b, lst in enumerate(lists):
        if len(lst) == 0:
            assert not sk[b].any() and not nm[b].any()
            continue
        e1 = torch.from_numpy(lst).to(dev)
        s1, n1, c1 = eng.sketch_block(packed, e1, [0, len(lst)], fill=fill, want_colsum=True)
        torch.cuda.synchronize()
        assert torch.equal(cs[offs[b]:offs[b + 1]], c1)
        scale = s1.abs().max().item()
        assert (sk[b] - s1[0]).abs().max().item() <= tol * scale, b
        assert torch.allclose(nm[b], n1[0], rtol=tol, atol=0), b
        assert (sk2[b] - 2 * s1[0]).abs().max().item() <= 2 * tol * scale, b
        assert torch.allclose(nm2[b], 2 * n1[0], rtol=tol, atol=0), b
    eng.close()


@pytest.mark.parametrize("nx,ny,n_sub,batch", [(3, 5, 2, 7), (8, 8, 3, 5), (20, 11, 1, 9), (1, 1, 4, 33), (16, 17, 2, 3)])
def test_loo_dots_paths_vs_float64_reference(dev, nx, ny, n_sub, batch):
    """pyrhe_engine_loo_dots: the register-tiled generic kernel (mixed tot/blk rows), the matrix-core path (block rows
    x fixed rows, fixed rows x fixed rows) and the self-Gram path (block rows x themselves) against a float64 torch evaluation of the row
    definition  tot - blk[z] * mul."""
    from pyrhe_amd import engine
    from pyrhe_amd.engine import Row
    period = 2048
    L = period * n_sub
    g = torch.Generator().manual_seed(nx * 100 + ny)
    eng = engine.SketchEngine(4 * 256, 4 * 256, max_entries=32, device=dev)
    store_c = torch.randn((batch, nx + 2, L), generator=g, dtype=torch.float32)       # block store: rows 1..nx used
    store = store_c.to(dev)
    mul_c = torch.randn(period, generator=g, dtype=torch.float32)
    mul = mul_c.to(dev)
    totx = torch.randn((nx, L), generator=g, dtype=torch.float64).to(dev)
    toty = torch.randn((ny, L), generator=g, dtype=torch.float64).to(dev)

    def blk_rows(with_tot):
        rows = []
        for a in range(nx):
            r = Row(totx[a] if with_tot else None, store[0, 1 + a], mul if a % 2 else None, store.stride(0))
            r_cpu_view = store_c[0, 1 + a]
            rows.append((r, r_cpu_view))
        return rows

    def run(xp, yp):
        x = [r for r, _ in xp]
        y = [r for r, _ in yp]
        got = eng.loo_dots(x, y, period, n_sub, batch).cpu()
        from tests.cpu_stub_engine import CpuStubEngine
        xc = [Row(None if r.tot is None else r.tot.cpu(), c, None if r.mul is None else mul_c, r.stride) for r, c in xp]
        yc = [Row(None if r.tot is None else r.tot.cpu(), c, None if r.mul is None else mul_c, r.stride) for r, c in yp]
        ref = CpuStubEngine.loo_dots(None, xc, yc, period, n_sub, batch)
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= 1e-12 * scale + 1e-9
    fixed = [(Row(toty[b]), None) for b in range(ny)]
    run(blk_rows(True), fixed)                       # generic: leave-one-out rows x fixed rows
    run(blk_rows(False), fixed)                      # matrix-core: block rows x fixed rows
    bx = blk_rows(False)
    run(bx, bx)                                      # self Gram (nx <= 16) or generic tiles
    lx = blk_rows(True)
    run(lx, lx)                                      # generic: leave-one-out Gram
    # fixed rows x fixed rows, one batch element: the float64-row matrix-core path
    fx = [(Row(totx[a]), None) for a in range(nx)]
    got = eng.loo_dots([r for r, _ in fx], [r for r, _ in fixed], period, n_sub, 1).cpu()
    ref = (totx @ toty.T).cpu()[None]
    assert (got - ref).abs().max().item() <= 1e-12 * ref.abs().max().item()
    eng.close()


def test_value_coding_recode_roundtrip_and_identical_results(dev):
    """pyrhe_recode_rows: PLINK -> value code -> PLINK is the identity (missing calls included); the value code is the
    allele count with 3 for missing; sketches, exact genotype sums and block statistics are IDENTICAL in both codings."""
    from pyrhe_amd import engine
    z = load_golden("rhe_1bin_missing")
    p, c = oracle_problem(z)
    raw = engine.upload_packed(c.packed, dev)
    val = engine.recode_rows(raw.clone(), "value")
    assert torch.equal(engine.recode_rows(val.clone(), "plink"), raw)
    codes = plink.decode_codes(c.packed, c.n_indv)                  # PLINK 2-bit codes per (snp, sample)
    want = np.array([0, 3, 1, 2], np.uint8)[codes]                   # 00->0, 01(missing)->3, 10->1, 11->2
    got = plink.decode_codes(val.cpu().numpy()[:, : c.packed.shape[1]], c.n_indv)
    assert np.array_equal(got, want)
    outs = []
    for coding, buf in (("plink", raw), ("value", val)):
        eng = engine.SketchEngine(c.n_indv, p.n_indv, max_entries=p.n_snp, precision="i8x3", device=dev)
        eng.set_coding(coding)
        keep = eng.keep_vector(p.missing_indv)
        gen = torch.Generator().manual_seed(3)
        Z = torch.zeros((p.k, eng.n_pad), device=dev)
        Z[:, : c.n_indv] = torch.randn((p.k, c.n_indv), generator=gen).to(dev)
        Z *= keep
        eng.set_rhs(Z, Z[:1].clone(), keep)
        s_, nmiss = eng.block_stats(buf)
        fill = (torch.arange(p.n_snp, device=dev) % 3).float()
        ent = torch.arange(p.n_snp, dtype=torch.int32, device=dev)
        sk, nm, cs = eng.sketch_block(buf, ent, [0, p.n_snp], fill=fill, want_colsum=True)
        torch.cuda.synchronize()
        outs.append((s_.clone(), nmiss.clone(), sk.clone(), nm.clone(), cs.clone()))
        eng.close()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_more_bins_than_one_segment_table(dev):
    """70 bins in one call: the by-value segment table holds 64, so the call is cut into two chunks; with RHS columns on
    two tiles.  Against one call per bin."""
    from pyrhe_amd import engine
    n, m, k = 3000, 1500, 12
    rng = np.random.default_rng(17)
    packed = engine.recode_rows(engine.synth_genotypes(m, n, 5, 0, dev), "value")
    bins = rng.integers(0, 70, m)
    lists = [np.nonzero(bins == b)[0].astype(np.int32) for b in range(70)]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in lists])])
    entries = torch.from_numpy(np.concatenate(lists)).to(dev)
    eng = engine.SketchEngine(n, n, max_entries=m, device=dev)
    eng.set_coding("value")
    keep = eng.keep_vector([])
    Z = torch.zeros((2 * k, eng.n_pad), device=dev)
    Z[:, :n] = torch.from_numpy(rng.standard_normal((2 * k, n)).astype(np.float32)).to(dev)
    y = torch.zeros((1, eng.n_pad), device=dev)
    y[:, :n] = torch.from_numpy(rng.standard_normal((1, n)).astype(np.float32)).to(dev)
    eng.set_rhs(Z, y, keep)
    sk, nm, _ = eng.sketch_block(packed, entries, offs)
    torch.cuda.synchronize()
    for b in (0, 1, 33, 63, 64, 65, 69):
        e1 = torch.from_numpy(lists[b]).to(dev)
        s1, n1, _ = eng.sketch_block(packed, e1, [0, len(lists[b])])
        torch.cuda.synchronize()
        assert (sk[b] - s1[0]).abs().max().item() <= 2e-6 * s1.abs().max().item(), b
        assert torch.allclose(nm[b], n1[0], rtol=1e-6), b
    eng.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_against_dense_float64(dev, seed):
    """Randomised small cohorts (sample counts that are not multiples of 4 / 16 / 256, dropped individuals, 1-90 SNPs,
    1-5 overlapping / empty bins, missing calls with random fills, both codings, both parity-safe precisions, with and
    without a second column tile): sketch, norms and genotype sums against a dense float64 evaluation of the definition."""
    from pyrhe_amd import engine
    rng = np.random.default_rng(1000 + seed)
    n0 = int(rng.integers(5, 900))
    m = int(rng.integers(1, 90))
    nbin = int(rng.integers(1, 6))
    k = int(rng.choice([1, 3, 10, 17]))
    precision = ["i8x3", "f16x2"][seed % 2]
    coding = ["plink", "value"][(seed // 2) % 2]
    drop = np.sort(rng.choice(n0, size=int(rng.integers(0, max(1, n0 // 10))), replace=False))
    keep_idx = np.setdiff1d(np.arange(n0), drop)
    n = len(keep_idx)
    p = rng.uniform(0.1, 0.9, m)
    geno = (rng.random((m, n0)) < p[:, None]).astype(np.int8) + (rng.random((m, n0)) < p[:, None]).astype(np.int8)
    miss = rng.random((m, n0)) < (0.03 if seed % 3 == 0 else 0.0)
    g_in = np.where(miss, -1, geno)
    packed = plink.encode_bed_rows(g_in)
    fill = rng.integers(0, 3, m).astype(np.float32)
    G = np.where(miss, fill[:, None], geno).astype(np.float64)[:, keep_idx]            # [m, n] after filling
    ok = (G.sum(1) > 0) & (G.sum(1) < 2 * n)                                         # the reference has no zero-variance guard
    lists = [np.sort(rng.choice(np.nonzero(ok)[0], size=int(rng.integers(0, ok.sum() + 1)), replace=False)).astype(np.int32)
             if ok.any() else np.zeros(0, np.int32) for _ in range(nbin)]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in lists])])
    eng = engine.SketchEngine(n0, n, max_entries=max(m, 1), precision=precision, device=dev)
    buf = engine.upload_packed(packed, dev)
    if coding == "value":
        engine.recode_rows(buf, "value")
        eng.set_coding("value")
    keep = eng.keep_vector(list(drop))
    Z = np.zeros((k, eng.n_pad), np.float32); Z[:, keep_idx] = rng.standard_normal((k, n)).astype(np.float32)
    Y = np.zeros((2, eng.n_pad), np.float32); Y[:, keep_idx] = rng.standard_normal((2, n)).astype(np.float32)
    eng.set_rhs(torch.from_numpy(Z).to(dev), torch.from_numpy(Y).to(dev), keep)
    ent = torch.from_numpy(np.concatenate(lists) if offs[-1] else np.zeros(0, np.int32)).to(dev)
    sk, nm, cs = eng.sketch_block(buf, ent, offs, fill=torch.from_numpy(fill).to(dev) if miss.any() else None, want_colsum=True)
    torch.cuda.synchronize()
    sk, nm, cs = sk.cpu().numpy(), nm.cpu().numpy(), cs.cpu().numpy()
    for b, lst in enumerate(lists):
        if len(lst) == 0:
            assert not sk[b].any() and not nm[b].any()
            continue
        Gb = G[lst]
        mu = (Gb.sum(1).astype(np.float32) / np.float32(n))
        d = (np.float32(1) / np.sqrt(mu * (np.float32(1) - np.float32(0.5) * mu))).astype(np.float64)
        X = ((Gb - mu[:, None].astype(np.float64)) * d[:, None]).T                    # [n, m_b]
        ref = (X @ (X.T @ Z[:, keep_idx].T.astype(np.float64))).T                     # [k, n]
        refn = ((X.T @ Y[:, keep_idx].T.astype(np.float64)) ** 2).sum(0)
        assert np.array_equal(cs[offs[b]:offs[b + 1]], Gb.sum(1).astype(np.int64))
        scale = np.abs(ref).max() + 1e-30
        assert np.abs(sk[b][:, keep_idx] - ref).max() <= 2e-5 * scale, (b, np.abs(sk[b][:, keep_idx] - ref).max() / scale)
        assert np.allclose(nm[b], refn, rtol=2e-5)
        assert not sk[b][:, np.setdiff1d(np.arange(eng.n_pad), keep_idx)].any()
    eng.close()


@pytest.mark.parametrize("precision", ["i8x3", "f16x2"])
def test_lists_longer_than_a_segment(dev, precision):
    """Bins with more entries than one segment holds (16384: GEMM 2 keeps a segment's row list in LDS) are cut into pieces
    that add into the same output slab.  The pieces must never share a launch (segments of a launch run concurrently):
    same results as one call per bin, and the same bits on every run."""
    from pyrhe_amd import engine
    n, m, k = 2000, 40000, 10
    rng = np.random.default_rng(11)
    packed = engine.synth_genotypes(m, n, 5, 0, dev)
    lists = [np.sort(rng.choice(m, sz, replace=False)).astype(np.int32) for sz in (20000, 300, 18000, 33000)]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in lists])])
    entries = torch.from_numpy(np.concatenate(lists)).to(dev)
    eng = engine.SketchEngine(n, n, max_entries=m, precision=precision, device=dev, sub_snps=65536)   # chunks of several segments
    keep = eng.keep_vector([])
    Z = torch.zeros((k, eng.n_pad), device=dev)
    Z[:, :n] = torch.from_numpy(rng.standard_normal((k, n)).astype(np.float32)).to(dev)
    y = torch.zeros((1, eng.n_pad), device=dev)
    y[:, :n] = torch.from_numpy(rng.standard_normal((1, n)).astype(np.float32)).to(dev)
    eng.set_rhs(Z, y, keep)
    sk, nm, _ = eng.sketch_block(packed, entries, offs)
    sk, nm = sk.clone(), nm.clone()
    sk_b, nm_b, _ = eng.sketch_block(packed, entries, offs)
    torch.cuda.synchronize()
    assert torch.equal(sk, sk_b) and torch.equal(nm, nm_b)
    for b, lst in enumerate(lists):
        s1, n1, _ = eng.sketch_block(packed, torch.from_numpy(lst).to(dev), [0, len(lst)])
        torch.cuda.synchronize()
        scale = s1.abs().max().item()
        assert (sk[b] - s1[0]).abs().max().item() <= 2e-6 * scale, b
        assert torch.allclose(nm[b], n1[0], rtol=2e-6, atol=0), b
    eng.close()


@pytest.mark.parametrize("nx,nys,nyf,n_sub,batch,want_self", [(8, 9, 10, 3, 7, True), (3, 5, 0, 2, 9, True), (5, 16, 16, 1, 33, False),
                                                              (16, 1, 4, 2, 4, True), (1, 1, 1, 4, 40, True), (7, 2, 3, 5, 3, False)])
@pytest.mark.parametrize("f32,tiles", [(False, 2), (True, 2), (False, 4), (True, 4), (True, 1)])
def test_loo_dots_multi_equals_separate_calls(dev, nx, nys, nyf, n_sub, batch, want_self, f32, tiles):
    """pyrhe_engine_loo_dots_multi (one pass over the block rows for the products with the sub-strided fixed rows, with the
    flat fixed rows per sub-row, and with themselves) against the three pyrhe_engine_loo_dots calls it replaces.  ``f32``:
    PYRHE_OPT_DOTS_F32, the products on the float32 matrix instruction with float32 partial sums over 128 elements -- within
    2e-6 of the float64 result's largest entry on random data (the real rows are positively correlated: 5e-9 on the SEs)."""
    from pyrhe_amd import engine
    from pyrhe_amd.engine import Row
    period = 1024
    L = period * n_sub
    g = torch.Generator().manual_seed(nx * 1000 + nys * 10 + nyf)
    eng = engine.SketchEngine(4 * 256, 4 * 256, max_entries=32, device=dev)
    store = torch.randn((batch, nx, L), generator=g, dtype=torch.float32).to(dev)
    mul = torch.randn(period, generator=g, dtype=torch.float32).to(dev)
    ys = torch.randn((nys, L), generator=g, dtype=torch.float64).to(dev)
    yf = torch.randn((max(nyf, 1), period), generator=g, dtype=torch.float64).to(dev)
    x = [Row(None, store[0, a], mul if a % 2 else None, store.stride(0)) for a in range(nx)]
    y_sub = [Row(ys[b]) for b in range(nys)]
    y_flat = [Row(yf[m]) for m in range(nyf)]
    assert eng.multi_ok(x, y_sub, y_flat, period)
    eng.set_option("dots_f32", 1 if f32 else 0)
    eng.set_option("dots_tiles", tiles)          # 16-row tiles per wave: 2 (default, two waves per SIMD) or 4
    o_sub, o_flat, o_self = eng.loo_dots_multi(x, y_sub, y_flat, want_self, period, n_sub, batch)
    r_sub = eng.loo_dots(x, y_sub, period, n_sub, batch)
    tol = lambda ref: (2e-6 if f32 else 1e-12) * ref.abs().max().item() + 1e-300
    assert (o_sub - r_sub).abs().max().item() <= tol(r_sub)
    if nyf:
        xs = [Row(None, store[0, a, s_ * period:(s_ + 1) * period], mul if a % 2 else None, store.stride(0))
              for a in range(nx) for s_ in range(n_sub)]
        r_flat = eng.loo_dots(xs, y_flat, period, 1, batch).reshape(batch, nx, n_sub, nyf)
        assert (o_flat - r_flat).abs().max().item() <= tol(r_flat)
    else:
        assert o_flat is None
    if want_self:
        r_self = eng.loo_dots(x, x, period, n_sub, batch)
        assert (o_self - r_self).abs().max().item() <= tol(r_self)
    else:
        assert o_self is None
    eng.close()


@pytest.mark.parametrize("coding,with_fill", [("value", False), ("plink", True), ("value", True)])
def test_large_cohort_paths_int8_vs_fp16(dev, coding, with_fill):
    """At >= 196608 samples the int8 engine takes paths the small fixtures never reach: a single list through the
    segmented flow (packed int8 GEMM 2) and RHE_DOM's second contraction on the packed kernel under both genotype views -- one
    fused launch for both estimates (default) or three launches (dom_fuse = 0) --, lists cut into 16384-entry pieces.  Same inputs through the fp16 hi+lo engine (other kernels in every stage):
    additive and dominance sketches and norms agree to 1e-5 of the largest entry."""
    from pyrhe_amd import engine
    n, m, k = 200_000, 18_000, 10
    rng = np.random.default_rng(21)
    packed = engine.synth_genotypes(m, n, 9, 0, dev)
    fill = None
    if with_fill:
        rows = torch.as_tensor(rng.choice(m, 400, replace=False), device=dev)
        packed[rows, 7] = (packed[rows, 7] & 0xF3) | 0x04          # missing call (01) in the second crumb of byte 7
        fill = torch.zeros(m, dtype=torch.float32, device=dev)
        fill[rows] = torch.as_tensor(rng.integers(0, 3, 400), dtype=torch.float32, device=dev)
    if coding == "value":
        packed = engine.recode_rows(packed, "value")
    lst = np.sort(rng.choice(m, 17_000, replace=False)).astype(np.int32)       # > 16384: two pieces, the second accumulates
    entries = torch.from_numpy(lst).to(dev)
    outs = {}
    for precision in ("i8x3", "i8x3-unfused", "f16x2"):
        eng = engine.SketchEngine(n, n, max_entries=m, precision=precision.split("-")[0], device=dev)
        eng.set_coding(coding)
        if precision == "i8x3-unfused":
            eng.set_option("dom_fuse", 0)
        keep = eng.keep_vector([])
        g = torch.Generator().manual_seed(4)
        Z = torch.zeros((k, eng.n_pad), device=dev)
        Z[:, :n] = torch.randn((k, n), generator=g).to(dev)
        y = torch.zeros((1, eng.n_pad), device=dev)
        y[:, :n] = torch.randn((1, n), generator=g).to(dev)
        eng.set_rhs(Z, y, keep)
        sk, nm, _ = eng.sketch_block(packed, entries, [0, len(lst)], fill=fill)
        a, na, d, nd = eng.sketch_block_dom(packed, entries, [0, len(lst)], fill=fill)
        torch.cuda.synchronize()
        if precision.startswith("i8x3"):       # one second contraction for both estimates (round 3), or three launches
            assert eng.last_path()["gemm2"] == (13 if precision == "i8x3" else 11)
        outs[precision] = [t.clone() for t in (sk, nm, a, na, d, nd)]
        eng.close()
    # 17 000 SNPs per sum: the fp16 engine accumulates them in float32 (a few 1e-6 of the largest entry), the int8 engine in
    # exact int32 -- the bound here is the parity bound, not the 2e-6 of the short lists above
    for x, r in zip(outs["i8x3"], outs["f16x2"]):
        assert (x - r).abs().max().item() <= 1e-5 * r.abs().max().item()
    # the additive sketch of the dominance call: the same digits, the same exact int32 sums and the same epilogue arithmetic in
    # the fused kernel (additive tiles beside the dominance tiles), in the three-launch form the same kernel path
    assert torch.equal(outs["i8x3"][0], outs["i8x3"][2])
    assert torch.equal(outs["i8x3-unfused"][0], outs["i8x3-unfused"][2])
    # fused against three launches: W1 and W2 share one digit scale per column in the fused form (24-bit digits either way)
    for x, r in zip(outs["i8x3"], outs["i8x3-unfused"]):
        assert (x - r).abs().max().item() <= 1e-6 * r.abs().max().item()


def test_pipelined_streams_equal_serial(dev):
    """pyrhe_engine_set_pipeline: GEMM 1 + mid kernels on one private stream, GEMM 2 on another, ordered by events (chunks of
    one call, cut lists that accumulate, several calls back to back) -- the same bits as the serial mode (int8 engine:
    integer atomics, fixed-order norm sums)."""
    from pyrhe_amd import engine
    n, m, k = 3000, 30000, 10
    rng = np.random.default_rng(31)
    packed = engine.synth_genotypes(m, n, 13, 0, dev)
    lists = [np.sort(rng.choice(m, sz, replace=False)).astype(np.int32) for sz in (18000, 700, 9000, 25, 17000)]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in lists])])
    entries = torch.from_numpy(np.concatenate(lists)).to(dev)
    res = []
    for piped in (False, True):
        eng = engine.SketchEngine(n, n, max_entries=m, precision="i8x3", device=dev, sub_snps=32768)
        eng.set_pipeline(piped)
        keep = eng.keep_vector([])
        g = torch.Generator().manual_seed(8)
        Z = torch.zeros((k, eng.n_pad), device=dev)
        Z[:, :n] = torch.randn((k, n), generator=g).to(dev)
        y = torch.zeros((2, eng.n_pad), device=dev)
        y[:, :n] = torch.randn((2, n), generator=g).to(dev)
        eng.set_rhs(Z, y, keep)
        sk, nm, cs = eng.sketch_block(packed, entries, offs, want_colsum=True)
        sk2, nm2, _ = eng.sketch_block(packed, entries, offs, out_sketch=sk.clone(), out_norm=nm.clone(), accumulate=True)
        s1, n1, _ = eng.sketch_block(packed, entries[:offs[1]], [0, offs[1]])
        torch.cuda.synchronize()
        res.append([t.clone() for t in (sk, nm, cs, sk2, nm2, s1, n1)])
        eng.close()
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("k,with_fill,coding", [(10, False, "value"), (20, False, "value"), (20, True, "plink"), (7, True, "value")])
def test_merged_segments_equal_one_segment_per_workgroup(dev, k, with_fill, coding):
    """PYRHE_OPT_G2_MERGE: the packed int8 GEMM 2 lets consecutive short lists (the bins of a block) share a workgroup -- one
    row list in LDS, one pass of the load ring, an epilogue per list inside the K loop, lists padded to whole ring turns.
    Same bits as one list per workgroup (exact int32 accumulation, same float32 epilogue), for overwrite and accumulate calls,
    with and without missing calls, 10- and 20-column kernels; info code 9 says whether lists really shared workgroups."""
    from pyrhe_amd import engine
    n, m = 5000, 12000
    rng = np.random.default_rng(100 + k)
    packed = engine.synth_genotypes(m, n, 17, 0, dev)
    fill = None
    if with_fill:
        rows = torch.as_tensor(rng.choice(m, 300, replace=False), device=dev)
        packed[rows, 11] = (packed[rows, 11] & 0xCF) | 0x10         # missing call (01) in the third crumb of byte 11
        fill = torch.zeros(m, dtype=torch.float32, device=dev)
        fill[rows] = torch.as_tensor(rng.integers(0, 3, 300), dtype=torch.float32, device=dev)
    if coding == "value":
        packed = engine.recode_rows(packed, "value")
    # odd, tiny, empty, ring-turn-sized lists -- and one list long enough to need the whole LDS budget for its rows alone
    # (it goes to the single-segment launch, the short ones share workgroups in the merged launch)
    sizes = (1250, 1250, 63, 1, 129, 640, 0, 2000, 1251, 64, 128, 700, 8800, 300, 200)
    lists = [np.sort(rng.choice(m, sz, replace=False)).astype(np.int32) for sz in sizes]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in lists])])
    entries = torch.from_numpy(np.concatenate(lists)).to(dev)
    res, merged = [], []
    for merge in (0, 2):                                            # 0: never; 2 workgroups per launch are enough to merge
        eng = engine.SketchEngine(n, n, max_entries=m, precision="i8x3", device=dev)
        eng.set_coding(coding)
        eng.set_option("g2_merge", merge)
        keep = eng.keep_vector([])
        g = torch.Generator().manual_seed(3)
        Z = torch.zeros((k, eng.n_pad), device=dev)
        Z[:, :n] = torch.randn((k, n), generator=g).to(dev)
        y = torch.zeros((1, eng.n_pad), device=dev)
        y[:, :n] = torch.randn((1, n), generator=g).to(dev)
        eng.set_rhs(Z, y, keep)
        sk, nm, _ = eng.sketch_block(packed, entries, offs, fill=fill)
        path = eng.last_path()
        assert path["gemm2"] == (11 if k <= 10 else 12) and path["segmented"]
        merged.append(path["g2_merged"])
        sk2, nm2, _ = eng.sketch_block(packed, entries, offs, fill=fill, out_sketch=sk.clone(), out_norm=nm.clone(), accumulate=True)
        torch.cuda.synchronize()
        res.append([t.clone() for t in (sk, nm, sk2, nm2)])
        eng.close()
    assert merged[0] == 1 and merged[1] > 1
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert torch.count_nonzero(res[0][0][6]).item() == 0             # the empty list's slab


def test_gemm1_launch_shape_options_do_not_change_a_bit(dev):
    """The int8 GEMM 1 accumulates in exact int32 (integer atomics across the sample splits), so every launch shape gives the
    same bits: workgroup target (G1_WORKGROUPS), round-filling choice of the split count (G1_FILL), 64- or 128-row kernel
    (G1_WIDE), grid order (G1_SPLIT_MAJOR), shared zero rows of the last operand tile (G1_ZERO_ROWS), the last tile's two live
    rows (10 probes + 1 phenotype = 34 digit rows) staged from the compact copy or like the other tiles (G1_COMPACT)."""
    from pyrhe_amd import engine
    n, m, k = 40_000, 24_000, 10
    rng = np.random.default_rng(77)
    packed = engine.recode_rows(engine.synth_genotypes(m, n, 23, 0, dev), "value")
    lists = [np.sort(rng.choice(m, sz, replace=False)).astype(np.int32) for sz in (9000, 5000, 7000)]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in lists])])
    entries = torch.from_numpy(np.concatenate(lists)).to(dev)
    res, paths = [], set()
    for opts in ({}, {"g1_fill": 0}, {"g1_workgroups": 300}, {"g1_workgroups": 5000}, {"g1_wide": 0}, {"g1_wide": 2},
                 {"g1_wide": 2, "g1_workgroups": 700, "g1_split_major": 1}, {"g1_zero_rows": 0},
                 {"g1_split_major": 2}, {"g1_wide": 0, "g1_split_major": 2, "g1_workgroups": 1000},
                 {"g1_compact": 0}, {"g1_compact": 0, "g1_wide": 2, "g1_workgroups": 900}):
        eng = engine.SketchEngine(n, n, max_entries=m, precision="i8x3", device=dev, sub_snps=32768)
        eng.set_coding("value")
        for name, v in opts.items():
            eng.set_option(name, v)
        keep = eng.keep_vector([])
        g = torch.Generator().manual_seed(12)
        Z = torch.zeros((k, eng.n_pad), device=dev)
        Z[:, :n] = torch.randn((k, n), generator=g).to(dev)
        y = torch.zeros((1, eng.n_pad), device=dev)
        y[:, :n] = torch.randn((1, n), generator=g).to(dev)
        eng.set_rhs(Z, y, keep)
        sk, nm, cs = eng.sketch_block(packed, entries, offs, want_colsum=True)
        torch.cuda.synchronize()
        paths.add(eng.last_path()["gemm1"])
        res.append([t.clone() for t in (sk, nm, cs)])
        eng.close()
    assert paths == {103, 203}                                  # both int8 GEMM 1 kernels took part
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize("k,dom", [(10, False), (20, False), (10, True)])
def test_many_short_lists_share_one_gemm1_launch(dev, k, dom):
    """A call of more lists than one segment table holds (kMaxSeg = 64): the chunks -- one table each -- share ONE GEMM 1 launch
    over all their rows (they fit the P buffer), the mid kernels and GEMM 2 go chunk by chunk over their slice of P.  150 lists
    (incl. one longer than a segment may be: cut, its pieces in different chunks) against one call per list; additive and
    dominance outputs."""
    from pyrhe_amd import engine
    n, m = 200_000 if dom else 30_000, 9_000
    rng = np.random.default_rng(31)
    packed = engine.recode_rows(engine.synth_genotypes(m, n, 3, 0, dev), "value")
    sizes = [int(x) for x in rng.integers(1, 60, 149)] + [2500]
    lists = [np.sort(rng.choice(m, sz, replace=False)).astype(np.int32) for sz in sizes]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in lists])])
    entries = torch.from_numpy(np.concatenate(lists)).to(dev)
    eng = engine.SketchEngine(n, n, max_entries=m, precision="i8x3", device=dev)
    eng.set_coding("value")
    keep = eng.keep_vector([])
    g = torch.Generator().manual_seed(9)
    Z = torch.zeros((k, eng.n_pad), device=dev)
    Z[:, :n] = torch.randn((k, n), generator=g).to(dev)
    y = torch.zeros((1, eng.n_pad), device=dev)
    y[:, :n] = torch.randn((1, n), generator=g).to(dev)
    eng.set_rhs(Z, y, keep)
    call = (lambda e_, o_: eng.sketch_block_dom(packed, e_, o_)) if dom else (lambda e_, o_: eng.sketch_block(packed, e_, o_)[:2])
    outs = [t.clone() for t in call(entries, offs)]
    torch.cuda.synchronize()
    assert eng.last_path()["segmented"]
    for b in (0, 1, 63, 64, 65, 100, 148, 149):
        e1 = torch.from_numpy(lists[b]).to(dev)
        ones = [t.clone() for t in call(e1, [0, len(lists[b])])]
        torch.cuda.synchronize()
        for a, r in zip(outs, ones):
            assert (a[b] - r[0]).abs().max().item() <= 2e-6 * max(r.abs().max().item(), 1e-30), b
    eng.close()


def test_sum_blocks_both_equals_the_two_single_sums(dev):
    """pyrhe_sum_blocks_both: the plain and the weighted float64 block sums from one read of the store -- the bits of
    pyrhe_sum_blocks and pyrhe_sum_blocks_weighted (same additions, same order), also on a slice of the store and on none."""
    from pyrhe_amd import engine
    g = torch.Generator().manual_seed(3)
    store = torch.randn((7, 3, 5, 1024), generator=g, dtype=torch.float32).to(dev)
    w = torch.arange(9, 2, -1, dtype=torch.float64, device=dev)
    for a, b in ((0, 7), (2, 5), (3, 3)):
        plain, weighted = engine.sum_blocks_both(store[a:b], w[a:b].contiguous())
        assert torch.equal(plain, engine.sum_blocks(store[a:b]))
        assert torch.equal(weighted, engine.sum_blocks(store[a:b], w[a:b].contiguous()))
    ref = (store.double() * w.reshape(-1, 1, 1, 1)).sum(dim=0)
    assert torch.allclose(engine.sum_blocks_both(store, w)[1], ref, rtol=1e-14, atol=0)


def test_stream_read_measures_a_bandwidth_and_rejects_bad_buffers(dev):
    """pyrhe_stream_read, bench.py's measuring stick (the box's measured stream bandwidth beside the nominal peak)."""
    from pyrhe_amd import engine
    buf = torch.zeros(1 << 28, dtype=torch.uint8, device=dev)
    gbs = engine.stream_read_gbs(buf)
    assert 100.0 < gbs < 20000.0                   # a real number: above PCIe rates, below any HBM peak
    with pytest.raises(engine.SketchError):
        engine.stream_read_gbs(buf[:8])            # fewer than 16 bytes


def test_gemm1_full_call_of_a_small_cohort_takes_no_sample_split(dev):
    """A call whose 512-row groups alone fill one round of the chip's resident workgroups (a small cohort's full call: 262 144
    entries) runs the 128-row GEMM 1 without sample splits -- plain stores instead of integer atomics.  Same bits as the split
    launch (`g1_fill = 0`), and as the 64-row kernel."""
    from pyrhe_amd import engine
    n, m, k = 20_000, 250_000, 10
    packed = engine.recode_rows(engine.synth_genotypes(m, n, 5, 0, dev), "value")
    entries = torch.arange(m, dtype=torch.int32, device=dev)
    res = []
    for opts in ({}, {"g1_fill": 0}, {"g1_wide": 0}):
        eng = engine.SketchEngine(n, n, max_entries=m, precision="i8x3", device=dev, sub_snps=262144)
        eng.set_coding("value")
        for name, v in opts.items():
            eng.set_option(name, v)
        keep = eng.keep_vector([])
        g = torch.Generator().manual_seed(8)
        Z = torch.zeros((k, eng.n_pad), device=dev)
        Z[:, :n] = torch.randn((k, n), generator=g).to(dev)
        y = torch.zeros((1, eng.n_pad), device=dev)
        y[:, :n] = torch.randn((1, n), generator=g).to(dev)
        eng.set_rhs(Z, y, keep)
        sk, nm, cs = eng.sketch_block(packed, entries, [0, 100_000, m], want_colsum=True)
        torch.cuda.synchronize()
        res.append([t.clone() for t in (sk, nm, cs)])
        eng.close()
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize("k,n_norm,tiles", [(16, 1, 4), (20, 2, 5), (26, 2, 6)])
@pytest.mark.parametrize("with_fill,coding", [(False, "value"), (True, "value"), (True, "plink")])
def test_gemm1_full_line_loads_do_not_change_a_bit(dev, k, n_norm, tiles, with_fill, coding):
    """The int8 GEMM 1 with more than three tiles steps over 256 samples = HALF a 128-byte line per row; G1_LINE (default)
    loads both halves of a line together into a wider register stage (one HBM fetch per line).  Exact int32 accumulation:
    the same bits as the half-line kernel, at several launch shapes, with fill values, in both genotype codings."""
    from pyrhe_amd import engine
    n, m = 30_000, 6000
    rng = np.random.default_rng(k)
    packed = engine.synth_genotypes(m, n, 41, 0, dev)
    fill = None
    if with_fill:
        rows = torch.as_tensor(rng.choice(m, 500, replace=False), device=dev)
        packed[rows, 1000] = (packed[rows, 1000] & 0xF0) | 0x05
        fill = torch.zeros(m, dtype=torch.float32, device=dev)
        fill[rows] = torch.as_tensor(rng.integers(0, 3, 500), dtype=torch.float32, device=dev)
    if coding == "value":
        packed = engine.recode_rows(packed, "value")
    lists = [np.sort(rng.choice(m, sz, replace=False)).astype(np.int32) for sz in (2500, 1700)]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in lists])])
    entries = torch.from_numpy(np.concatenate(lists)).to(dev)
    res = []
    for opts in ({"g1_line": 0}, {}, {"g1_workgroups": 200}, {"g1_workgroups": 4000, "g1_fill": 0}, {"g1_split_major": 2}):
        eng = engine.SketchEngine(n, n, max_entries=m, precision="i8x3", device=dev)
        eng.set_coding(coding)
        for name, v in opts.items():
            eng.set_option(name, v)
        keep = eng.keep_vector([])
        g = torch.Generator().manual_seed(3)
        Z = torch.zeros((k, eng.n_pad), device=dev)
        Z[:, :n] = torch.randn((k, n), generator=g).to(dev)
        y = torch.zeros((n_norm, eng.n_pad), device=dev)
        y[:, :n] = torch.randn((n_norm, n), generator=g).to(dev)
        eng.set_rhs(Z, y, keep)
        sk, nm, cs = eng.sketch_block(packed, entries, offs, fill=fill, want_colsum=True)
        torch.cuda.synchronize()
        assert eng.last_path()["gemm1"] == 100 + tiles
        res.append([t.clone() for t in (sk, nm, cs)])
        eng.close()
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize("precision,digits", [("i8x3", 3), ("i8x3", 2), ("i8x3", 1), ("f16x2", 3)])
@pytest.mark.parametrize("nbin,with_sketch", [(1, True), (3, True), (3, False)])
def test_cross_columns_gram_vs_float64(dev, precision, digits, nbin, with_sketch):
    """PYRHE_COL_CROSS through the C ABI (pyrhe_engine_sketch_block_cross): out_cross[b] = R_x^T X_b X_b^T R_x for the cross
    columns R_x, against float64 arithmetic on the decoded genotypes standardized the reference's way (base.py:291-296) --
    beside sketch and norm columns (whose outputs must not move), in both kernel flows (one list / several lists per call),
    with cross columns alone (no second contraction), and fed with 3, 2 or 1 int8 digits (PYRHE_COL_DIGITS: 24, 16, 8 bits of
    the column maximum; tolerance 2e-6 / 1e-4 / 2e-2 of the largest entry).  Accumulating calls add."""
    from pyrhe_amd import engine
    n, m, k, c = 1003, 330, 4, 5
    g = synth.synth_rows(20251212, 77, m, n)
    packed = engine.upload_packed(plink.encode_bed_rows(g), dev)
    rng = np.random.default_rng(8)
    Z = rng.standard_normal((k, n))
    y = rng.standard_normal((1, n))
    W = rng.standard_normal((c, n))
    W[0] = 1.0                                                   # an intercept: X^T 1 = 0 up to the float32 mean
    W[1] = rng.integers(0, 2, n)                                 # a binary covariate: exact in one digit
    eng = engine.SketchEngine(n, n, max_entries=m, precision=precision, device=dev)
    keep = eng.keep_vector([])
    pad = lambda a: torch.nn.functional.pad(torch.from_numpy(a.astype(np.float32)), (0, eng.n_pad - n)).to(dev)
    eng.set_rhs(pad(Z) if with_sketch else None, pad(y), keep, cross_cols=pad(W), cross_digits=digits)
    assert eng.n_cross == c and int(eng.lib.pyrhe_engine_info(eng._h, 10)) == c
    if precision == "i8x3":
        ncol = (k if with_sketch else 0) + 1 + c
        want_slots = ncol + 1 + (ncol - c if digits < 2 else ncol) + (ncol - c if digits < 3 else ncol)
        assert int(eng.lib.pyrhe_engine_info(eng._h, 11)) == want_slots
    lists = [np.sort(rng.choice(m, size=rng.integers(40, 200), replace=False)).astype(np.int32) for _ in range(nbin)]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in lists])])
    ent = torch.from_numpy(np.concatenate(lists)).to(dev)
    out_cross = torch.full((nbin, c, c), 7.0, dtype=torch.float64, device=dev)          # overwritten, not added to
    sk, nm, _ = eng.sketch_block(packed, ent, offs, out_cross=out_cross)
    sk2, nm2, _ = eng.sketch_block(packed, ent, offs, out_sketch=sk.clone(), out_norm=nm.clone(), out_cross=out_cross, accumulate=True)
    torch.cuda.synchronize()
    tol = {3: 2e-6, 2: 1e-4, 1: 2e-2}[digits] if precision == "i8x3" else 5e-6
    for b, rows in enumerate(lists):
        G = g[rows].astype(np.float32)
        mu = G.mean(axis=1, dtype=np.float32)
        d = (np.float32(1) / np.sqrt(mu * (np.float32(1) - np.float32(0.5) * mu))).astype(np.float64)
        X = (G.astype(np.float64) - mu[:, None].astype(np.float64)) * d[:, None]                    # [m_b, n]
        V = X @ W.astype(np.float32).astype(np.float64).T                                            # [m_b, c]
        S = V.T @ V
        got = out_cross[b].cpu().numpy() / 2                                                        # two accumulated calls
        assert np.abs(got - S).max() <= tol * np.abs(S).max(), (b, np.abs(got - S).max() / np.abs(S).max())
        assert np.array_equal(got, got.T)
        # exact covariates (the binary column, up to the intercept's zero) stay at float32 accuracy whatever the digit count
        assert abs(got[1, 1] - S[1, 1]) <= 5e-6 * S[1, 1]
        vy = X @ y[0].astype(np.float32).astype(np.float64)
        assert abs(float(nm[b, 0]) - vy @ vy) <= 1e-5 * (vy @ vy)
        if with_sketch:
            ref = (X.T @ (X @ Z.astype(np.float32).astype(np.float64).T)).T                          # [k, n]
            assert np.abs(sk[b, :, :n].double().cpu().numpy() - ref).max() <= (5e-6 if precision != "f16" else 2e-3) * np.abs(ref).max()
    eng.close()


def test_cross_columns_dominance_and_api_errors(dev):
    """The dominance encoding's Gram matrix (RHE_DOM: X_dom of rhe_dom.py:15-41) through the same entry point, and the
    argument checks of the cross API."""
    from pyrhe_amd import engine
    n, m, k, c = 900, 200, 3, 4
    g = synth.synth_rows(20251212, 5, m, n)
    packed = engine.upload_packed(plink.encode_bed_rows(g), dev)
    rng = np.random.default_rng(9)
    Z, y, W = rng.standard_normal((k, n)), rng.standard_normal((1, n)), rng.standard_normal((c, n))
    eng = engine.SketchEngine(n, n, max_entries=m, precision="i8x3", device=dev)
    keep = eng.keep_vector([])
    pad = lambda a: torch.nn.functional.pad(torch.from_numpy(a.astype(np.float32)), (0, eng.n_pad - n)).to(dev)
    eng.set_rhs(pad(Z), pad(y), keep, cross_cols=pad(W), cross_digits=3)
    rows = np.arange(m, dtype=np.int32)
    ent = torch.from_numpy(rows).to(dev)
    cr = torch.zeros((1, c, c), dtype=torch.float64, device=dev)
    crd = torch.zeros_like(cr)
    eng.sketch_block_dom(packed, ent, [0, m], out_cross=cr, out_cross_dom=crd)
    torch.cuda.synchronize()
    G = g.astype(np.float64)
    mu = G.mean(axis=1)
    X = (G - mu[:, None]) / np.sqrt(mu * (1 - mu / 2))[:, None]
    p_ = mu / 2
    enc = np.where(G == 1, (2 * p_)[:, None], np.where(G == 2, (4 * p_ - 2)[:, None], 0.0))
    Xd = (enc - enc.mean(axis=1, keepdims=True)) / (2 * p_ * (1 - p_))[:, None]
    W64 = W.astype(np.float32).astype(np.float64)
    for got, Xx in ((cr, X), (crd, Xd)):
        V = Xx @ W64.T
        S = V.T @ V
        assert np.abs(got[0].cpu().numpy() - S).max() <= 5e-6 * np.abs(S).max()
    # the plain entry points refuse an engine with cross columns; the cross entry point one without
    with pytest.raises(engine.SketchError, match="out_cross"):
        engine._check(eng.lib.pyrhe_engine_sketch_block(eng._h, packed.data_ptr(), m, ent.data_ptr(), (engine.C.c_int64 * 2)(0, m), 1, None,
                                                        torch.empty((1, k, eng.n_pad), device=dev).data_ptr(),
                                                        torch.empty((1, 1), dtype=torch.float64, device=dev).data_ptr(), None, 0, None), "sketch_block")
    eng.set_rhs(pad(Z), pad(y), keep)
    with pytest.raises(engine.SketchError, match="PYRHE_COL_CROSS"):
        engine._check(eng.lib.pyrhe_engine_sketch_block_cross(eng._h, packed.data_ptr(), m, ent.data_ptr(), (engine.C.c_int64 * 2)(0, m), 1, None,
                                                              None, None, cr.data_ptr(), None, None, None, 0, None), "sketch_block_cross")
    # digit counts must not grow with the column index; sketch columns keep all three
    roles = (engine.C.c_int32 * 3)(engine.COL_NORM | (2 << 8), engine.COL_CROSS | (3 << 8), engine.COL_CROSS | (1 << 8))
    R = pad(rng.standard_normal((3, n)))
    assert eng.lib.pyrhe_engine_set_rhs(eng._h, R.data_ptr(), roles, 3, keep.data_ptr(), None) == -1
    roles = (engine.C.c_int32 * 2)(engine.COL_SKETCH | (2 << 8), engine.COL_NORM)
    assert eng.lib.pyrhe_engine_set_rhs(eng._h, R.data_ptr(), roles, 2, keep.data_ptr(), None) == -1
    eng.close()


def test_upload_with_overlapped_recode_equals_upload_then_recode(dev):
    """models.py loads with ``upload_packed(recode="value")``: every staged piece is recoded on a second stream while the next
    one crosses PCIe.  Same bytes as the upload followed by pyrhe_recode_rows, over several pieces and a ragged last one."""
    from pyrhe_amd import engine
    rng = np.random.default_rng(3)
    n = 4001
    rows = rng.integers(0, 256, size=(523, plink.row_bytes(n)), dtype=np.uint8)
    want = engine.recode_rows(engine.upload_packed(rows, dev), "value")
    got = engine.upload_packed(rows, dev, chunk_bytes=64 * plink.row_bytes(n), recode="value")       # 9 pieces of 64 rows
    assert torch.equal(got, want)
    back = engine.recode_rows(got.clone(), "plink")
    assert torch.equal(back, engine.upload_packed(rows, dev))


@pytest.mark.parametrize("precision", ["i8x3", "i8f16"])
def test_wide_int8_pass_42_columns_six_tiles(dev, precision):
    """PYRHE_MAX_COLS = 46 for the int8 precisions: 20 sketch + 2 norm + 20 one-digit cross columns (GENIE G x E with 10
    covariates in ONE pass) = 87 packed digit rows = the six-tile GEMM 1; sketches, norms and the cross Gram against float64
    arithmetic on the decoded genotypes; the fp16 precisions refuse more than 30 columns.  ``i8f16`` (int8 first contraction,
    single-fp16 second): its operand buffer used to be sized for the fp16 image alone -- 64 bytes per sample against the 96 of
    six int8 tiles (advisor, round 3) --, so the same six-tile image is checked there, with the fp16 rounding of its second
    contraction in the sketch bound."""
    from pyrhe_amd import engine
    n, m, ks, c = 2003, 300, 20, 20
    g = synth.synth_rows(20251212, 123, m, n)
    packed = engine.upload_packed(plink.encode_bed_rows(g), dev)
    rng = np.random.default_rng(18)
    Z, y, W = rng.standard_normal((ks, n)), rng.standard_normal((2, n)), rng.standard_normal((c, n))
    W[:c // 2] = rng.integers(0, 3, (c // 2, n))                      # small-integer covariates: exact in one digit
    eng = engine.SketchEngine(n, n, max_entries=m, precision=precision, device=dev)
    keep = eng.keep_vector([])
    pad = lambda a: torch.nn.functional.pad(torch.from_numpy(a.astype(np.float32)), (0, eng.n_pad - n)).to(dev)
    eng.set_rhs(pad(Z), pad(y), keep, cross_cols=pad(W), cross_digits=1)
    assert int(eng.lib.pyrhe_engine_info(eng._h, 11)) == 3 * 22 + 20 + 1
    sk_tol = 5e-6 if precision == "i8x3" else 2e-3
    lists = [np.arange(0, 150, dtype=np.int32), np.arange(150, 300, dtype=np.int32)]
    ent = torch.from_numpy(np.concatenate(lists)).to(dev)
    out_cross = torch.zeros((2, c, c), dtype=torch.float64, device=dev)
    sk, nm, _ = eng.sketch_block(packed, ent, [0, 150, 300], out_cross=out_cross)
    torch.cuda.synchronize()
    assert eng.last_path()["gemm1"] == 106
    for b, rows in enumerate(lists):
        G = g[rows].astype(np.float32)
        mu = G.mean(axis=1, dtype=np.float32)
        d = (np.float32(1) / np.sqrt(mu * (np.float32(1) - np.float32(0.5) * mu))).astype(np.float64)
        X = (G.astype(np.float64) - mu[:, None].astype(np.float64)) * d[:, None]
        ref = (X.T @ (X @ Z.astype(np.float32).astype(np.float64).T)).T
        assert np.abs(sk[b, :, :n].double().cpu().numpy() - ref).max() <= sk_tol * np.abs(ref).max()
        for t in range(2):
            vy = X @ y[t].astype(np.float32).astype(np.float64)
            assert abs(float(nm[b, t]) - vy @ vy) <= 1e-5 * (vy @ vy)
        V = X @ W.astype(np.float32).astype(np.float64).T
        S = V.T @ V
        got = out_cross[b].cpu().numpy()
        assert np.abs(got - S).max() <= 2e-2 * np.abs(S).max()
        h = c // 2
        assert np.abs(got[:h, :h] - S[:h, :h]).max() <= 5e-6 * np.abs(S[:h, :h]).max()       # the exact (integer) covariates
    eng.close()
    e16 = engine.SketchEngine(n, n, max_entries=m, precision="f16x2", device=dev)
    with pytest.raises(engine.SketchError, match="n_cols"):
        e16.set_rhs(pad(Z), pad(y), e16.keep_vector([]), cross_cols=pad(W))
    e16.close()
