"""Kernel-level parity of every model family's sketches, block by block, against the per-block arrays the REFERENCE produced
(tests/golden: ``ref_t0_XXz_blocks`` / ``XXUz_blocks`` / ``UXXz_blocks`` / ``yXXy_blocks``, shape [estimate, block, k, N]).

The model-level tests (test_models_gpu.py) pin these families through the solve of the normal equations, which for GENIE is
ill-conditioned; the comparisons here are conditioning-independent and test the kernels directly:
  * G x E estimates of GENIE        -- e.Z columns in, e (.) rows out       (streaming_genie.py:33-49, genie.py:59-75)
  * XXUz with covariates            -- the U Z sketch columns               (base.py:410-412)
  * UXXz = U (block XXz)            -- pyrhe_engine_apply_projector         (base.py:407-408)
  * dominance estimates of RHE_DOM  -- both genotype views                  (rhe_dom.py:15-41)
Tolerance: 5e-6 of the block's largest entry (the reference's own float32 accumulation-order noise), 1e-5 for yXXy.

Also here: StreamingGENIE("G+GxE") -- BASELINE config 5's model, which the reference cannot run -- against its oracle-side
definition (oracle Problem.compat = "fixed").
"""
import numpy as np
import pytest
import torch

from oracle import fixtures, plink, synth
from oracle import rhe_oracle as O
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu
BLOCK_TOL = 5e-6


def _model(z, tmp_path, cls_name, **extra):
    from pyrhe_amd import models
    from pyrhe_amd.logger import Logger
    c = fixtures.cohort_from_npz(z)
    paths = fixtures.write_cohort_files(c, str(tmp_path))
    kw = dict(geno_file=paths["geno"], annot_file=paths["annot"], pheno_file=paths["pheno"], cov_file=paths.get("cov"),
              num_jack=int(z["param_num_jack"]), num_random_vec=int(z["param_num_random_vec"]), seed=int(z["param_seed"]),
              device="cuda", log=Logger(suppress=True, debug_mode=False))
    if "param_genie_model" in z:
        kw.update(env_file=paths["env"], genie_model=str(z["param_genie_model"]))
    kw.update(extra)
    return getattr(models, cls_name)(**kw)


@pytest.mark.parametrize("materialize", [False, True])
@pytest.mark.parametrize("name,cls", [("genie_gxe_2bin", "GENIE"), ("genie_gxe_cov", "GENIE"), ("genie_gxe_nxe_cov", "GENIE"),
                                      ("rhe_3bin_cov", "RHE"), ("rhe_dom_2bin", "RHE_DOM"), ("rhe_dom_cov", "RHE_DOM")])
def test_block_sketches_of_every_family_vs_reference_blocks(name, cls, materialize, tmp_path):
    """``materialize``: the U Z sketch columns as the reference has them (XXUz rows compared entry by entry); default: the
    covariates as cross columns, XXUz kept as its projection W^T XXUz = S (Q W^T Z) [k, c] -- compared with W^T applied to
    the reference's XXUz blocks (all that base.py:589-591 reads of them, since UXXz lies in span(W))."""
    z = load_golden(name)
    if materialize and "ref_t0_XXUz_blocks" not in z:
        pytest.skip("no covariates: nothing to materialise")
    m = _model(z, tmp_path, cls, materialize_xxuz=materialize)
    m(trait=0)
    r = m._runner
    assert r.cov is None or r.cross == (not materialize)
    keep = r.keep_idx
    ref = z["ref_t0_XXz_blocks"].astype(np.float64)               # [ne, jn, k, N]
    refy = z["ref_t0_yXXy_blocks"]
    ne_g = 2 * r.plan.n_bin if r.dom else r.plan.n_bin + r.n_gxe    # the N x E estimate has no per-block sketch
    jn = r.plan.n_jack
    assert ref.shape[0] >= ne_g and ref.shape[1] == jn
    worst = 0.0
    for j in range(jn):
        A, C, rr = r._block_estimates(*r._stored_block(j))
        for a in range(ne_g):
            want = ref[a, j]
            scale = np.abs(want).max()
            got = A[a].cpu().numpy()[:, keep]
            if scale == 0:
                assert not got.any()
                continue
            dev_ = np.abs(got - want).max() / scale
            worst = max(worst, dev_)
            assert dev_ <= BLOCK_TOL, ("XXz", name, a, j, dev_)
            assert abs(float(rr[a, 0]) - refy[a, j]) <= 1e-5 * abs(refy[a, j]), ("yXXy", name, a, j)
        if "ref_t0_XXUz_blocks" in z:
            refc = z["ref_t0_XXUz_blocks"].astype(np.float64)
            refd = z["ref_t0_UXXz_blocks"].astype(np.float64)
            D = r._apply_U(A, f32=True)                            # pyrhe_engine_apply_projector on the block's XXz
            for a in range(ne_g):
                if r.cross:
                    want = refc[a, j] @ np.asarray(r.cov, np.float64)                   # [k, c]
                    assert np.abs(C[a].cpu().numpy() - want).max() <= BLOCK_TOL * np.abs(want).max(), ("W^T XXUz", name, a, j)
                else:
                    sc = np.abs(refc[a, j]).max()
                    assert np.abs(C[a].cpu().numpy()[:, keep] - refc[a, j]).max() <= BLOCK_TOL * sc, ("XXUz", name, a, j)
                sd = np.abs(refd[a, j]).max()
                assert np.abs(D[a].cpu().numpy()[:, keep] - refd[a, j]).max() <= BLOCK_TOL * sd, ("UXXz", name, a, j)
    print(f"[{name}] worst block deviation {worst:.2e} of the block maximum")


def test_streaming_genie_gxe_definition_vs_oracle():
    """BASELINE config 5's model, StreamingGENIE("G+GxE"): the reference raises (base_streaming.py:45-46), its non-streaming
    class overwrites the last G x E estimate with the N x E sketch (base.py:467-481).  The product defines the model as its
    equations state it -- G bins + G x E bins, every total the sum over its blocks, stochastic trace for the G x E rows
    (genie.py:46-94) -- and that definition is the oracle's ``compat="fixed"``.  N = 20 000, two bins, the int8 kernels."""
    from pyrhe_amd import engine, estimate as est
    from pyrhe_amd.streaming import BlockPlan, SketchRunner, n_engines_needed
    dev = torch.device("cuda", 0)
    n, m, jn, k, nbin = 20000, 2000, 10, 10, 2
    geno = engine.recode_rows(engine.synth_genotypes(m, n, 20251212, 3, dev), "value")
    g = synth.synth_rows(20251212, 3, m, n)
    rng = np.random.default_rng(13)
    env = rng.integers(0, 2, n).astype(np.float64)
    mu = g.mean(axis=1, keepdims=True)
    X = ((g - mu) / np.sqrt(mu * (1 - mu / 2))).T
    y = X @ rng.normal(0, np.sqrt(0.3 / m), m) + (X * env[:, None]) @ rng.normal(0, np.sqrt(0.2 / m), m) + rng.normal(0, np.sqrt(0.5), n)
    y -= y.mean()
    annot = np.zeros((m, nbin), np.int64)
    annot[np.arange(m), np.arange(m) % nbin] = 1
    p = O.Problem(packed=plink.encode_bed_rows(g), n_indv_orig=n, annot=annot, pheno=y.reshape(-1, 1), n_jack=jn, k=k, seed=0,
                  env=env.reshape(-1, 1), genie_model="G+GxE", compat="fixed")
    p.Z = O.make_probe_vectors(0, n, k)
    ref = O.genie_results(p, O.precompute(p))
    plan = BlockPlan(annot, jn)
    engs = []
    for _ in range(n_engines_needed(k, False, True)):
        e_ = engine.SketchEngine(n, n, plan.max_block, precision="i8x3", device=dev)
        e_.set_coding("value")
        engs.append(e_)
    runner = SketchRunner(engs, geno, plan, (0, jn), n_orig=n, missing_indv=[], Z=p.Z, env=env, genie_model="G+GxE", seed=0,
                          streaming=True)
    assert not runner.quirk_last_total and runner.n_est == 2 * nbin
    sysm = runner.run_trait(y, check_missing=False)
    sig = est.solve_all(sysm.T, sysm.q)
    Tref = ref["T"]
    assert np.abs(sysm.T - Tref).max() <= 2e-6 * np.abs(Tref).max()
    r_sig = ref["sigma_ests_total"]
    assert np.abs(sig[-1] - r_sig).max() <= 1e-5 * np.abs(r_sig).max(), (sig[-1], r_sig)
    se = np.asarray(est.jackknife_se(sig[:-1], jn))
    assert np.abs(se - ref["sig_errs"]).max() <= 1e-5 * np.abs(ref["sig_errs"]).max(), (se, ref["sig_errs"])
    for e_ in engs:
        e_.close()
