"""Covariates as CROSS columns (W through the first contraction, S = W^T X X^T W per (block, bin) back) against the
materialised U Z sketch columns, on the CPU stand-in engine: the two forms of the covariate terms of the normal equations
(base.py:583-593: XXUz only enters through sum(XXUz * UXXz), UXXz lies in span(W), W^T X X^T (U Z) = S (Q W^T Z)) must give the
same systems, for every model family the stand-in covers, in both UXXz semantics and with more probe vectors than a pass holds."""
import numpy as np
import pytest
import torch

from tests.helpers import load_golden, oracle_problem


def _runner(name, materialize, compat, streaming):
    from pyrhe_amd.streaming import BlockPlan, SketchRunner, n_engines_needed
    from tests.cpu_stub_engine import CpuStubEngine
    z = load_golden(name)
    p, c = oracle_problem(z)
    plan = BlockPlan(c.annot, p.n_jack)
    has_gxe = p.genie_model in ("G+GxE", "G+GxE+NxE")
    engs = [CpuStubEngine(c.n_indv, p.n_indv) for _ in range(n_engines_needed(p.k, True, has_gxe))]
    geno = torch.zeros((c.packed.shape[0], engs[0].pitch), dtype=torch.uint8)
    geno[:, : c.packed.shape[1]] = torch.from_numpy(c.packed.copy())
    env = None if p.genie_model is None else np.delete(np.asarray(p.env, np.float64).reshape(-1), p.missing_indv)
    r = SketchRunner(engs, geno, plan, (0, p.n_jack), n_orig=c.n_indv, missing_indv=p.missing_indv, Z=p.Z, cov=p.cov,
                     env=env, genie_model=p.genie_model, impute=p.impute, seed=p.seed, streaming=streaming, compat=compat,
                     materialize_xxuz=materialize)
    return r, p, z


@pytest.mark.parametrize("name,compat,streaming", [("rhe_3bin_cov", "fixed", False), ("srhe_3bin_cov", "reference", True),
                                                  ("rhe_2bin_cov_k35", "fixed", False), ("genie_gxe_cov", "reference", False),
                                                  ("genie_gxe_nxe_cov", "reference", False), ("genie_gxe_cov", "fixed", True)])
def test_cross_columns_give_the_systems_of_the_materialised_xxuz(name, compat, streaming):
    rc, p, z = _runner(name, False, compat, streaming)
    rm, _, _ = _runner(name, True, compat, streaming)
    assert rc.cross and not rm.cross
    sc = rc.run_trait(p.pheno[:, 0])
    sm = rm.run_trait(p.pheno[:, 0])
    assert "W" in rc.cross_loc and "UZ" not in rc.loc and "UZ" in rm.loc
    assert np.array_equal(sc.M, sm.M)
    scale = np.abs(sm.T).max(axis=(1, 2), keepdims=True)
    assert np.abs(sc.T - sm.T).max() <= 1e-9 * scale.max(), np.abs((sc.T - sm.T) / scale).max()
    assert np.allclose(sc.q, sm.q, rtol=1e-12, atol=0)
    # the materialised cross-check path of pass 2 understands the projected form too
    rc.pass2_impl = "torch"
    st = rc.pass2()
    assert np.abs(st.T - sm.T).max() <= 1e-9 * scale.max()


@pytest.mark.parametrize("name,compat,streaming,materialize", [("rhe_3bin_cov", "fixed", False, False), ("srhe_3bin_cov", "reference", True, True),
                                                              ("rhe_2bin_cov_k35", "fixed", False, False), ("genie_gxe_nxe_cov", "reference", False, False),
                                                              ("genie_gxe_cov", "fixed", True, True)])
def test_xxz_uz_products_through_the_covariate_projections(name, compat, streaming, materialize):
    """<XXz_a, U Z> of the normal equations (base.py:611-618) as sum_{s, m} <XXz_a[s], W_m> (Q W^T Z)[m, s] -- U Z = W (Q W^T Z)
    lies in span(W) -- instead of a product with the U Z rows: what pass 2 does when those rows would be the 17th fixed row of
    its one-read kernel (16 estimates: RHE_DOM with 8 bins + covariates).  Same systems, both covariate forms, every family."""
    ra, p, _ = _runner(name, materialize, compat, streaming)
    rb, _, _ = _runner(name, materialize, compat, streaming)
    rb.force_tk_proj = True
    sa = ra.run_trait(p.pheno[:, 0])
    sb = rb.run_trait(p.pheno[:, 0])
    scale = np.abs(sa.T).max()
    assert np.abs(sa.T - sb.T).max() <= 1e-10 * scale
    assert np.array_equal(sa.q, sb.q)


def test_cross_gram_equals_projection_of_the_reference_xxuz_blocks():
    """S_a (Q W^T z_b) per block against W^T applied to the REFERENCE's own per-block XXUz arrays."""
    rc, p, z = _runner("rhe_3bin_cov", False, "fixed", False)
    rc.run_trait(p.pheno[:, 0])
    ref = z["ref_t0_XXUz_blocks"].astype(np.float64)                 # [ne, jn, k, N]
    W = np.asarray(p.cov, np.float64)
    for j in range(p.n_jack):
        _, C, _ = rc._block_estimates(*rc._stored_block(j))          # [ne, k, c]
        want = np.einsum("akn,nc->akc", ref[:, j], W)
        assert np.abs(C.numpy() - want).max() <= 2e-6 * np.abs(want).max()


def test_cross_columns_fall_back_when_they_do_not_pay():
    from pyrhe_amd.streaming import cross_columns_pay, plan_passes
    assert cross_columns_pay(10, 10, False) and cross_columns_pay(10, 19, False)
    assert not cross_columns_pay(10, 20, False)        # a pass of their own against one pass with the U Z columns
    assert not cross_columns_pay(10, 31, False)
    assert cross_columns_pay(10, 10, True)             # two passes either way, the second one without a GEMM 2
    passes, at = plan_passes(10, 2, 2, cross=[10, 10])
    assert len(passes) == 2 and at == [(1, 0), (1, 10)] and passes[1] == []
    passes, at = plan_passes(10, 1, 1, cross=[10])
    assert len(passes) == 1 and at == [(0, 0)]


def test_wide_int8_passes_take_cross_columns_beyond_thirty():
    """An int8 engine takes up to 46 columns while the packed digit rows of the pass fit six tiles (3 per sketch / norm column,
    `cross_digits` per cross column, 1 for the ones column <= 96): GENIE G x E with 10 one-digit covariates is ONE pass."""
    from pyrhe_amd.streaming import cross_columns_pay, plan_passes
    passes, at = plan_passes(10, 2, 2, cross=[10, 10], cross_digits=1, wide=True)          # 20 + 2 + 20 columns, 87 digit rows
    assert len(passes) == 1 and at == [(0, 0), (0, 10)]
    passes, at = plan_passes(10, 2, 2, cross=[10, 10], cross_digits=3, wide=True)          # 127 digit rows: a pass of their own (61)
    assert len(passes) == 2 and at == [(1, 0), (1, 10)]
    passes, at = plan_passes(10, 2, 2, cross=[15, 15], cross_digits=1, wide=True)          # 52 columns: the second group moves on
    assert len(passes) == 2 and at == [(0, 0), (1, 0)]
    assert cross_columns_pay(10, 25, False, cross_digits=1, wide=True) and not cross_columns_pay(10, 25, False)
    assert not cross_columns_pay(10, 25, False, cross_digits=3, wide=True)                 # 33 + 75 + 1 digit rows


@pytest.mark.parametrize("name,streaming", [("rhe_dom_cov", False), ("srhe_dom_cov", True)])
def test_rhe_dom_with_cross_columns_matches_the_reference_and_the_materialised_form(name, streaming):
    """RHE_DOM + covariates on the stand-in engine: S and S_dom (both encodings' cross Grams) in place of the U Z columns --
    the reference's variance components at 2e-6 and the materialised form's systems at 1e-9."""
    from pyrhe_amd import estimate as est
    from pyrhe_amd.streaming import BlockPlan, SketchRunner
    from tests.cpu_stub_engine import CpuStubEngine
    z = load_golden(name)
    p, c = oracle_problem(z)
    plan = BlockPlan(c.annot, p.n_jack)
    out = {}
    for materialize in (False, True):
        eng = CpuStubEngine(c.n_indv, p.n_indv)
        geno = torch.zeros((c.packed.shape[0], eng.pitch), dtype=torch.uint8)
        geno[:, : c.packed.shape[1]] = torch.from_numpy(c.packed.copy())
        r = SketchRunner(eng, geno, plan, (0, p.n_jack), n_orig=c.n_indv, missing_indv=p.missing_indv, Z=p.Z, cov=p.cov,
                         impute=p.impute, seed=p.seed, streaming=streaming, compat="reference", dom=True, materialize_xxuz=materialize)
        assert r.cross == (not materialize)
        out[materialize] = r.run_trait(p.pheno[:, 0])
    a, b = out[False], out[True]
    assert np.abs(a.T - b.T).max() <= 1e-9 * np.abs(b.T).max() and np.allclose(a.q, b.q, rtol=1e-12, atol=0)
    sig = est.solve_all(a.T, a.q)
    assert np.allclose(sig[-1], z["ref_t0_res_sigma_ests_total"], rtol=2e-6)
    assert np.allclose(est.jackknife_se(sig[:-1], p.n_jack), z["ref_t0_res_sig_errs"], rtol=2e-5)


def test_digits_per_cross_column_follow_the_dynamic_range():
    """column_digits: one int8 digit only for columns it represents to <= 2 % of their rms (max / rms <= 8) or exactly (an
    intercept, indicators however rare, small counts); a skewed covariate takes two, an extreme one three; cohorts below 16 384
    samples (base 3) keep three everywhere (advisor, round 3: one digit for every covariate quantised heavy-tailed columns to
    zeros almost everywhere)."""
    from pyrhe_amd.streaming import column_digits, plan_passes
    rng = np.random.default_rng(5)
    n = 20000
    gauss = rng.normal(size=n)
    cols = np.stack([np.ones(n), gauss, (rng.random(n) < 0.001).astype(float), rng.integers(0, 3, n).astype(float),
                     np.exp(2.5 * gauss), gauss ** 2, np.where(np.arange(n) == 7, 1e6, gauss), rng.integers(0, 5, n) * 0.37], axis=1)
    got = column_digits(cols, 1)
    ratio = np.abs(cols).max(axis=0) / np.sqrt((cols ** 2).mean(axis=0))
    # (max / rms <= sqrt(N): a single outlier of any size stays below the 2048 of the third digit until N = 4.2 M)
    assert ratio[4] > 8 and ratio[5] > 8 and 100 < ratio[6] <= np.sqrt(n) and ratio[7] <= 8
    assert list(got) == [1, 1, 1, 1, 2, 2, 2, 1]
    assert list(column_digits(cols, 3)) == [3] * 8 and list(column_digits(cols, 2)) == [2] * 8
    big = np.zeros((5_000_000, 1))
    big[::2] = 1e-3
    big[7] = 5.0
    assert list(column_digits(big, 1)) == [3]                   # max / rms = 2132
    # the planner counts the rows the columns really take: 10 probes + phenotype + 10 covariates, one of them on two digits,
    # still fit three tiles (45 packed digit rows); all of them on three digits do not fit one pass beside the sketch columns
    passes, at = plan_passes(10, 1, 1, cross=[10], wide=True, cross_rows=[11])
    assert len(passes) == 1 and at == [(0, 0)]
    passes, at = plan_passes(10, 2, 2, cross=[10, 10], wide=True, cross_rows=[12, 20])     # 66 + 32 + 1 = 99 rows: the second group moves on
    assert len(passes) == 2 and at == [(0, 0), (1, 0)]


def test_heavy_tailed_covariate_gets_more_digits_and_keeps_the_systems():
    """A covariate set with an intercept, a Gaussian, a log-normal (max / rms = 40) and a rare indicator at N = 16 500 on the
    stand-in engine, which quantises the cross columns like the int8 first contraction: with the automatic per-column digits
    the systems agree with the all-three-digits run far better than with ONE digit forced on every column, the engine is
    handed the columns sorted by their digits, and the cross Grams come back in the covariates' own order."""
    from pyrhe_amd.streaming import BlockPlan, SketchRunner
    from oracle import fixtures
    from oracle import rhe_oracle as O
    from tests.cpu_stub_engine import CpuStubEngine
    n, m, jn, k = 16500, 48, 3, 4
    c = fixtures.make_cohort("heavy", n, m, seed=11, nbin=2, with_cov=0)
    rng = np.random.default_rng(9)
    g = rng.normal(size=n)
    cov = np.stack([np.ones(n), np.exp(2.0 * g), rng.normal(size=n), (rng.random(n) < 0.002).astype(float)], axis=1)
    y = c.pheno[:, 0] + 0.3 * cov[:, 1] / cov[:, 1].std()
    y = y - y.mean()
    plan = BlockPlan(c.annot, jn)
    Z = O.make_probe_vectors(0, n, k)
    out = {}
    for tag, digits in (("auto", None), ("three", 3), ("one", 1)):
        eng = CpuStubEngine(n, n)
        geno = torch.zeros((c.packed.shape[0], eng.pitch), dtype=torch.uint8)
        geno[:, : c.packed.shape[1]] = torch.from_numpy(c.packed.copy())
        r = SketchRunner(eng, geno, plan, (0, jn), n_orig=n, missing_indv=[], Z=Z, cov=cov, seed=0, streaming=False, compat="fixed",
                         cross_digits=digits)
        assert r.cross
        out[tag] = r.run_trait(y)
        if tag == "auto":
            assert list(r.cross_col_digits["W"]) == [1, 2, 1, 1] and r.cross_digits == 1
            assert eng.cross_digits == [2, 1, 1, 1]                              # the log-normal column first
            assert r.cross_loc["W"][1].tolist() == [1, 0, 2, 3]
            S = r._cross_grams([b.sum(dim=0) for b in r.blocks_cr])              # [nbin, c, c] in the covariates' order
            Wq = torch.from_numpy(eng.cr[r.cross_loc["W"][1].numpy()][:, :n])    # what the engine was given, re-ordered back
            assert torch.allclose(Wq[[0, 2, 3]], torch.from_numpy(cov.T[[0, 2, 3]]), rtol=0, atol=np.abs(cov[:, 2]).max() / 250)
            assert S.shape == (2, 4, 4) and torch.allclose(S, S.transpose(1, 2))
    scale = np.abs(out["three"].T).max()
    d_auto = np.abs(out["auto"].T - out["three"].T).max() / scale
    d_one = np.abs(out["one"].T - out["three"].T).max() / scale
    assert d_auto <= 2e-8 and d_auto < 0.2 * d_one, (d_auto, d_one)
