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
