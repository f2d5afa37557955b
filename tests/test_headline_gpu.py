"""Oracle parity of the EXACT kernels and flows the headline number comes from, at the sizes where the engine selects them.

bench.py's `value` is measured on ``gemm1_i8w_kernel<3, .., VC>`` (128 rows per wave; cohorts of >= 196 608 samples, else
``gemm1_i8_kernel<3>``) + ``gemm2_i8p_kernel<1>`` in the segmented flow (several
jackknife blocks per engine call), value-coded genotypes from the device generator, k = 10, one bin.  Those instantiations
only engage for large cohorts / batched calls, so the small golden fixtures never reach them.  Here the same runner
configuration as bench.py is driven at N = 500 000 (BASELINE config 3's cohort size), N = 50 000 (config 2) and N = 200 000
with 8 bins + 10 covariates (config 4's shape: four-tile GEMM 1, two-column-group packed GEMM 2), on few enough SNPs that the
reference-order oracle (oracle/rhe_oracle.py, torch CPU float32 GEMV pairs) finishes in seconds on the box's host cores.

Tolerances: see _check -- per block within 2e-6 of the exactly summed float32 chain and within the reference's own
accumulation noise of the reference-order oracle; sigma^2 and jackknife SE within 1e-5 relative (the north star's bound).
Every test asserts, through pyrhe_engine_info, WHICH kernels ran -- a silent fall to the fp16 kernels fails the test.
"""
import numpy as np
import pytest
import torch

from oracle import plink, synth
from oracle import rhe_oracle as O

pytestmark = pytest.mark.gpu
SEED = 20251212          # bench.py's SYNTH_SEED


def _setup(n, m, jn, k, nbin=1, n_cov=0, row0=0, materialize=False, genie=None, cross_digits=None, dom=False, heavy=False):
    from pyrhe_amd import engine
    from pyrhe_amd.streaming import BlockPlan, SketchRunner, engine_capacity, n_engines_needed
    dev = torch.device("cuda", 0)
    geno = engine.recode_rows(engine.synth_genotypes(m, n, SEED, row0, dev), "value")      # as bench.py: generated, then value-coded
    g = synth.synth_rows(SEED, row0, m, n)                                                  # CPU twin of the generator
    annot = np.zeros((m, nbin), np.int64)
    annot[np.arange(m), (np.arange(m) * 7 + 3) % nbin] = 1
    rng = np.random.default_rng(1)
    y = rng.normal(size=n)
    # a heritable part so that sigma^2_g is not statistically zero (relative errors of ~0 +- SE mean nothing)
    mu = g[:: max(1, m // 50)].mean(axis=1, keepdims=True)
    Xs = (g[:: max(1, m // 50)] - mu) / np.sqrt(mu * (1 - mu / 2))
    y = y + Xs.T @ rng.normal(0, 0.2, Xs.shape[0])
    y -= y.mean()
    cov = None
    if n_cov:
        cov = np.concatenate([np.ones((n, 1)), np.random.default_rng(2).normal(size=(n, n_cov - 1))], axis=1)
        if heavy:            # a log-normal covariate (max / rms ~ 50) that also drives the phenotype, and a rare indicator
            cov[:, 1] = np.exp(2.0 * cov[:, 1])
            cov[:, 2] = (np.random.default_rng(4).random(n) < 0.002).astype(np.float64)
            y = y + 0.5 * cov[:, 1] / cov[:, 1].std()
            y -= y.mean()
    env = None
    kw = {}
    if genie:
        env = np.random.default_rng(3).integers(0, 2, n).astype(np.float64)      # Bernoulli(0.5), SURVEY 8d's config 5
        kw = dict(env=env.reshape(-1, 1), genie_model=genie, compat="fixed")
    p = O.Problem(packed=plink.encode_bed_rows(g), n_indv_orig=n, annot=annot, pheno=y.reshape(-1, 1), n_jack=jn, k=k, seed=0, cov=cov, **kw)
    p.Z = O.make_probe_vectors(0, n, k)
    p.dom = bool(dom)
    plan = BlockPlan(annot, jn)
    engs = []
    for _ in range(n_engines_needed(k, cov is not None, bool(genie))):
        e_ = engine.SketchEngine(n, n, max_entries=engine_capacity(plan, n), precision="i8x3", device=dev, sub_snps=engine_capacity(plan, n))
        e_.set_coding("value")
        engs.append(e_)
    runner = SketchRunner(engs, geno, plan, (0, jn), n_orig=n, missing_indv=[], Z=p.Z, cov=cov, seed=0, streaming=bool(genie),
                          compat="fixed", materialize_xxuz=materialize, env=env, genie_model=genie, cross_digits=cross_digits, dom=dom)
    return p, runner, engs


def _check(p, runner, engs, want_path, blocks_per_call, with_exact=True):
    """Per block: X X^T z, X X^T (U z), ||X^T y||^2 against the oracle in the reference's operation order AND against the
    oracle with the same float32 operands summed exactly (O.ACCUM = "exact": each product of the chain correctly rounded).
    The distance between those two oracles is the reference's own accumulation noise (it grows with N: 6e-6 of the block
    maximum at N = 500k); the engine accumulates in exact integers and must sit at the exact end:
        X X^T z:      |engine - exact| <= 2e-6 of the block maximum (measured 2.7e-7),
                      |engine - reference| <= max(5e-6, 1.5 x |reference - exact| + 2e-6)
        X X^T (U z):  2e-5 against either oracle (see the comment at the assertion: the reference's float32 rounding of X)
    sigma^2 and jackknife SE: 1e-5 relative against the reference-order oracle (the north star's bound).
    ``with_exact=False`` (the covariate / GENIE / RHE_DOM shapes, whose oracle passes are the expensive ones): the reference-order
    oracle alone -- per block within 1.5e-5 of the block maximum (its own noise is 5e-6 at N = 200k, 6e-6 at N = 500k; the kernels
    are the ones held to the exact sum above)."""
    from pyrhe_amd import estimate as est
    jn, N, nb = p.n_jack, p.n_indv, p.nbin
    runner.begin_trait(p.pheno[:, 0], check_missing=False)
    assert min(runner._blocks_per_call(), jn) == blocks_per_call         # the blocks of this cohort go in ONE engine call
    runner.pass1()
    torch.cuda.synchronize()
    path = engs[0].last_path()
    assert {k: path[k] for k in want_path} == want_path, path
    st = O.precompute(p)                                      # the oracle: every block in the reference's operation order
    st_x = st
    if with_exact:
        try:
            O.ACCUM = "exact"
            st_x = O.precompute(p)
        finally:
            O.ACCUM = "reference"
    use_cov = p.cov is not None
    ne_g = nb * (2 if (p.genie_model or p.dom) else 1)       # G bins, then G x E (or dominance) bins
    W = None if not use_cov else np.asarray(p.cov, np.float64)
    # covariates as cross columns (the default): the engine returns S = W^T X X^T W, the runner keeps XXUz as its projection
    # W^T XXUz = S (Q W^T Z) [k, c] -- all the normal equations read of it (base.py:589-591); compared with W^T of the oracle's
    proj = (lambda a: a @ W) if (use_cov and runner.cross) else (lambda a: a)
    worst = {}

    def compare(what, got, ref, exact):
        scale = max(np.abs(exact).max(), 1e-30)
        d = (np.abs(got - exact).max() / scale, np.abs(got - ref).max() / scale, np.abs(ref - exact).max() / scale)
        w = worst.setdefault(what[0], [0.0, 0.0, 0.0])
        for i_ in range(3):
            w[i_] = max(w[i_], float(d[i_]))
        return what, d

    checks = []
    for j in range(jn):
        A, C, r = runner._block_estimates(*runner._stored_block(j))
        blk, blx = st.blocks[j], st_x.blocks[j]
        for b in range(ne_g):
            checks.append(compare(("XXz", j, b), A[b, :, :N].cpu().numpy(), blk.XXz[b], blx.XXz[b]))
            assert abs(float(r[b, 0]) - blk.yXXy[b]) <= 1e-5 * abs(blk.yXXy[b]), ("yXXy", j, b)
            assert not with_exact or abs(float(r[b, 0]) - blx.yXXy[b]) <= 2e-6 * abs(blx.yXXy[b]), ("yXXy vs exact", j, b)
            if use_cov:
                got = C[b].cpu().numpy() if runner.cross else C[b, :, :N].cpu().numpy()
                checks.append(compare(("XXUz", j, b), got, proj(blk.XXUz[b]), proj(blx.XXUz[b])))
        assert not A[:, :, N:].any()
    for name, w in worst.items():
        print(f"[N={N}, {nb} bins] {name}: worst deviation of the block maximum -- engine vs exact sum {w[0]:.1e}, engine vs reference order "
              f"{w[1]:.1e}, reference order vs exact sum (the reference's own noise) {w[2]:.1e}")
    for what, (d_x, d_r, fl) in checks:
        if what[0] == "XXz" and not with_exact:
            assert d_r <= 1.5e-5, (what, "engine vs reference order", d_r)
        elif what[0] == "XXz":
            assert d_x <= 2e-6, (what, "engine vs exact", d_x)
            assert d_r <= max(5e-6, 1.5 * fl + 2e-6), (what, "engine vs reference order", d_r, "reference noise", fl)
        else:
            # X X^T (U z): the reference multiplies float32-ROUNDED standardized genotypes, and a SNP has only three distinct
            # ones -- their rounding errors do not average out over the samples, they add up N-fold against the column mean of
            # the right-hand side.  Negligible for z (mean ~ 1/sqrt(N) beside a result ~ sqrt(N)), visible for U z, whose
            # X^T (U z) is ~150x smaller at the same mean: measured 6e-6 of the block maximum at N = 200k for BOTH oracles
            # (the exact-sum one rounds X the same way).  The engine never rounds X; it sits that far from both.  These
            # sketches only enter the O(c / N) covariate corrections of T; sigma^2 and SE below are held to 1e-5.
            # Covariates as cross columns fed with ONE int8 digit (the default from N = 16 384 on): W^T XXUz = S (Q W^T Z)
            # carries the 8-bit quantisation of W, about 1 % -- harmless where it is used (the O(c / N) terms of T; sigma^2 and
            # SE are held to 1e-5 below) and held to 2e-5 with all three digits (cross_digits = 3).
            lim = 3e-2 if (runner.cross and runner.cross_digits == 1) else 2e-5
            assert d_r <= lim and d_x <= lim, (what, d_x, d_r, fl)
    sysm = runner.pass2()
    sig = est.solve_all(sysm.T, sysm.q)
    se = np.asarray(est.jackknife_se(sig[:-1], jn))
    ref = O.genie_results(p, st) if p.genie_model else (O.rhe_dom_results(p, st) if p.dom else O.rhe_results(p, st))
    assert np.array_equal(sysm.M, st.M)
    # 1e-5 relative per entry; a component below 5 % of the largest one (a bin whose sigma^2 is statistically zero) is
    # judged against 5 % of the largest instead of against itself
    # (GENIE: 1e-5 of the largest component, the bound of tests/test_families_gpu.py -- its normal equations amplify the
    # reference's own float32 accumulation noise 100x more than RHE's, tests/helpers.py:accumulation_noise_floor)
    floor = (lambda v: np.full(np.shape(v), np.abs(v).max())) if p.genie_model else (lambda v: np.maximum(np.abs(v), 0.05 * np.abs(v).max()))
    rel = np.abs(sig[-1] - ref["sigma_ests_total"]) / floor(ref["sigma_ests_total"])
    rel_se = np.abs(se - ref["sig_errs"]) / floor(ref["sig_errs"])
    print(f"[N={N}, {nb} bins] sigma^2 rel {rel.max():.1e}, SE rel {rel_se.max():.1e}")
    assert rel.max() <= 1e-5, (rel, sig[-1], ref["sigma_ests_total"])
    assert rel_se.max() <= 1e-5, (rel_se, se, ref["sig_errs"])
    for e_ in engs:
        e_.close()


def test_headline_kernels_at_n500k_vs_oracle():
    """BASELINE config 3's cohort size: three 200-SNP blocks in ONE engine call (bench.py batches three 10 000-SNP blocks)."""
    p, runner, engs = _setup(500_000, 600, 3, 10)
    _check(p, runner, engs, {"gemm1": 203, "gemm2": 11, "segmented": True}, blocks_per_call=3)       # 203: gemm1_i8w_kernel<3>


def test_headline_kernels_at_n500k_narrow_gemm1_vs_oracle():
    """The 64-rows-per-wave GEMM 1 (`g1_wide = 0`; what cohorts below 196 608 samples and four or more tiles run) at the same
    cohort size (three 128-SNP blocks): both int8 GEMM 1 kernels are held to the oracle at N = 500 000."""
    p, runner, engs = _setup(500_000, 384, 3, 10, row0=1000)
    for e_ in engs:
        e_.set_option("g1_wide", 0)
    _check(p, runner, engs, {"gemm1": 103, "gemm2": 11, "segmented": True}, blocks_per_call=3)


def test_headline_kernels_at_n50k_vs_oracle():
    """BASELINE config 2's cohort size: three 1000-SNP blocks per call, the batched small-cohort flow of bench.py."""
    p, runner, engs = _setup(50_000, 3000, 3, 10, row0=17)
    _check(p, runner, engs, {"gemm1": 103, "gemm2": 11, "segmented": True}, blocks_per_call=3)


def test_headline_kernels_at_n50k_wide_gemm1_vs_oracle():
    """The 128-rows-per-wave GEMM 1 at N = 50 000: bench.py's calls there (26 blocks = 130 000 rows) have rows enough for it
    and take it by the automatic rule; three 1000-SNP blocks do not, so the test forces it (`g1_wide = 2`)."""
    p, runner, engs = _setup(50_000, 3000, 3, 10, row0=29)
    for e_ in engs:
        e_.set_option("g1_wide", 2)
    _check(p, runner, engs, {"gemm1": 203, "gemm2": 11, "segmented": True}, blocks_per_call=3)


@pytest.mark.parametrize("materialize", [False, True])
def test_c4_shape_8bins_covariates_vs_oracle(materialize):
    """BASELINE config 4's shape (8 bins + 10 covariates incl. the intercept) at N = 100 000, both forms of the covariate terms.
    Cross columns carrying all three digits (what cohorts below 16 384 samples run; forced here): 10 probes + phenotype + 10
    covariates = 21 right-hand columns = 64 packed digit rows = the four-tile int8 GEMM 1, 10 sketch columns = the ONE-column-group
    packed int8 GEMM 2 with merged lists; XXz, yXXy and the projection W^T XXUz = S (Q W^T Z) per block (held to 2e-5).
    ``materialize``: the U Z sketch columns of the reference (`materialize_xxuz`; what covariate sets too wide for cross columns
    run): 20 sketch columns = the two-column-group packed GEMM 2, XXUz rows entry by entry."""
    p, runner, engs = _setup(100_000, 1200, 3, 10, nbin=8, n_cov=10, row0=5, cross_digits=3, materialize=materialize)
    assert runner.cross == (not materialize) and (materialize or runner.cross_digits == 3)
    _check(p, runner, engs, {"gemm1": 104, "gemm2": 12 if materialize else 11, "segmented": True}, blocks_per_call=3, with_exact=False)


def test_c4_8bins_covariates_at_n500k_vs_oracle():
    """BASELINE config 4 at ITS cohort size, N = 500 000: 8 bins + 10 covariates, three blocks of 320 SNPs (40 per list) in one
    call -- the kernels bench.py --n-bin 8 --n-cov 10 times: the covariate columns enter GEMM 1 with ONE int8 digit
    (PYRHE_COL_DIGITS, the default from N = 16 384 on), 10 x 3 + 3 + 10 + 1 = 44 packed digit rows = THREE tiles = the
    128-rows-per-wave GEMM 1; one-group packed GEMM 2 with merged lists.  XXz and yXXy per block at the usual bounds, W^T XXUz
    within the 8-bit quantisation of W (3e-2), sigma^2 and SE within 1e-5 of the oracle."""
    p, runner, engs = _setup(500_000, 960, 3, 10, nbin=8, n_cov=10, row0=11)
    assert runner.cross and runner.cross_digits == 1
    _check(p, runner, engs, {"gemm1": 203, "gemm2": 11, "segmented": True, "g2_merged": 7}, blocks_per_call=3, with_exact=False)


def test_heavy_tailed_covariates_get_their_digits_vs_oracle():
    """Covariates with a log-normal column and a rare 0 / 1 indicator at N = 20 000 (one int8 digit per cross column is the
    default from N = 16 384 on): the log-normal column -- max / rms = 50, one digit would quantise nearly all of it to zero --
    takes two digits (streaming.column_digits), the indicator is exact in one; sigma^2 and SE within 1e-5 of the oracle, the
    projections W^T XXUz per block within the 8-bit quantisation of the Gaussian columns (advisor, round 3)."""
    p, runner, engs = _setup(20_000, 600, 3, 10, nbin=2, n_cov=5, row0=71, heavy=True)
    assert runner.cross and list(runner.cross_col_digits["W"]) == [1, 2, 1, 1, 1]
    runner.begin_trait(p.pheno[:, 0], check_missing=False)
    assert int(engs[0].lib.pyrhe_engine_info(engs[0]._h, 11)) == 3 * 11 + 1 + 6          # 40 packed digit rows: three tiles
    _check(p, runner, engs, {"gemm1": 103, "gemm2": 11, "segmented": True}, blocks_per_call=3, with_exact=False)


def test_c5_streaming_genie_gxe_at_n500k_vs_oracle():
    """BASELINE config 5 at its cohort size: StreamingGENIE("G+GxE"), N = 500 000, one Bernoulli environment, three blocks of
    128 SNPs in one call -- Z and e.Z = 20 sketch columns + y, e.y = 22 right-hand columns = 67 digit rows = the FIVE-tile
    int8 GEMM 1 and the two-column-group packed GEMM 2, unmerged (one list per block); per-block X X^T z, e.(X X^T (e.z)),
    both norms, sigma^2 and SE against the oracle's definition of the model (Problem.compat = "fixed", streaming_genie.py:33-49)."""
    p, runner, engs = _setup(500_000, 384, 3, 10, row0=23, genie="G+GxE")
    assert not runner.quirk_last_total and runner.n_est == 2
    _check(p, runner, engs, {"gemm1": 105, "gemm2": 12, "segmented": True, "g2_merged": 1}, blocks_per_call=3)


def test_genie_gxe_covariates_one_wide_pass_vs_oracle():
    """StreamingGENIE("G+GxE") with 6 covariates at N = 100 000: Z, e.Z (20 sketch columns), y, e.y and the one-digit cross columns
    W, e.W (12) = 34 right-hand columns in ONE int8 engine pass (PYRHE_MAX_COLS = 46; 79 packed digit rows = the five-tile GEMM 1)
    where round 2's engine needed two passes; per-block sketches, the projections W^T XXUz of both estimate families, sigma^2 and
    SE against the oracle (Problem.compat = "fixed")."""
    p, runner, engs = _setup(100_000, 384, 3, 10, n_cov=6, row0=53, genie="G+GxE")
    assert runner.cross and runner.cross_digits == 1 and runner.wide_cols
    runner.begin_trait(p.pheno[:, 0], check_missing=False)
    assert runner.n_pass == 1 and runner.pass_cross == [12]
    _check(p, runner, engs, {"gemm1": 105, "gemm2": 12, "segmented": True}, blocks_per_call=3, with_exact=False)


def test_rhe_dom_segmented_flow_at_n200k_vs_oracle():
    """RHE_DOM through the segmented flow (round 3): three blocks in ONE engine call -- one GEMM 1 launch for both genotype views
    (gemm1_i8_kernel<3, .., GV = 2>), then ONE packed GEMM 2 launch over all segments for both estimates (gemm2_i8p_kernel<2, ..,
    GV = 2>: additive tiles, dominance tiles under the G view and under the 2 [g == 2] view from one read of the genotypes;
    rhe_dom.py:15-41, streaming_rhe_dom.py:9-68).  Additive and
    dominance sketches and norms per block, sigma^2 and SE against the oracle."""
    p, runner, engs = _setup(200_000, 384, 3, 10, row0=37, dom=True)
    _check(p, runner, engs, {"gemm1": 103, "gemm2": 13, "segmented": True}, blocks_per_call=3)


def test_rhe_dom_bins_and_covariates_segmented_at_n200k_vs_oracle():
    """The same with 2 bins and 5 covariates as cross columns under BOTH encodings (S and S_dom per block and bin)."""
    p, runner, engs = _setup(200_000, 384, 3, 10, nbin=2, n_cov=5, row0=41, dom=True)
    assert runner.cross and runner.cross_digits == 1         # 10 x 3 + 3 + 5 + 1 = 39 packed digit rows: three tiles
    _check(p, runner, engs, {"gemm1": 103, "gemm2": 13, "segmented": True}, blocks_per_call=3, with_exact=False)


def test_blocks_beyond_one_call_are_split_into_row_pieces():
    """A jackknife block with more rows than one engine call addresses (32-bit row byte offsets: N = 500k, over 34 308 SNPs)
    goes as row sub-ranges whose later pieces accumulate.  Forced here with a small row limit: same systems as unsplit."""
    from pyrhe_amd import engine
    from pyrhe_amd.streaming import BlockPlan, SketchRunner
    dev = torch.device("cuda", 0)
    n, m, jn, k = 3000, 900, 3, 10
    geno = engine.recode_rows(engine.synth_genotypes(m, n, SEED, 41, dev), "value")
    rng = np.random.default_rng(6)
    annot = np.zeros((m, 3), np.int64)
    annot[np.arange(m), rng.integers(0, 3, m)] = 1
    annot[rng.choice(m, 60, replace=False), 1] = 1
    plan = BlockPlan(annot, jn)
    np.random.seed(0)
    Z = np.random.randn(n, k)
    y = rng.normal(size=n); y -= y.mean()
    cov = np.concatenate([np.ones((n, 1)), rng.normal(size=(n, 2))], axis=1)
    out = []
    for limit, recompute in ((None, False), (128, False), (77, True)):
        eng = engine.SketchEngine(n, n, plan.max_block, device=dev)
        eng.set_coding("value")
        runner = SketchRunner(eng, geno, plan, (0, jn), n_orig=n, missing_indv=[], Z=Z, cov=cov, seed=0, streaming=True,
                              compat="reference", recompute_pass2=recompute)
        runner.row_limit = limit
        out.append(runner.run_trait(y, check_missing=False))
        eng.close()
    for o in out[1:]:
        assert np.abs(o.T - out[0].T).max() <= 2e-6 * np.abs(out[0].T).max()
        assert np.allclose(o.q, out[0].q, rtol=2e-6)


@pytest.mark.parametrize("nbin,n_cov,m,jn,n_pieces,dom", [(1, 0, 72_000, 2, 3, False), (3, 4, 72_000, 2, 2, False), (1, 0, 60_000, 3, 2, False),
                                                            (1, 3, 60_000, 3, 2, True)])
def test_long_blocks_row_pieces_in_one_call_equal_one_call_per_piece(nbin, n_cov, m, jn, n_pieces, dom):
    """N = 500 000 with 36 000-SNP blocks (4.5 GB of rows: beyond the 32-bit row span of a call) and with 20 000-SNP blocks (one
    list longer than the 15 104 entries a kernel chunk takes of a list) -- what M of several millions gives at jn = 100: the
    row pieces of the blocks of a call as list windows of ONE engine call (one GEMM 1 launch over all its rows,
    PYRHE_OPT_LIST_SPAN; three 20 000-SNP blocks = six windows) against the engine's own handling (`batch_pieces = False`: one
    call per piece beyond the row span, later pieces accumulating; long lists cut into chunks).  Same systems."""
    from pyrhe_amd import engine
    from pyrhe_amd.streaming import BlockPlan, SketchRunner, engine_capacity
    dev = torch.device("cuda", 0)
    n, k = 500_000, 10
    geno = engine.recode_rows(engine.synth_genotypes(m, n, SEED, 77, dev), "value")
    annot = np.zeros((m, nbin), np.int64)
    annot[np.arange(m), (np.arange(m) * 5 + 1) % nbin] = 1
    plan = BlockPlan(annot, jn)
    Z = np.random.default_rng(5).standard_normal((n, k)).astype(np.float32)
    y = np.random.default_rng(6).standard_normal(n); y -= y.mean()
    cov = None
    if n_cov:
        cov = np.random.default_rng(7).standard_normal((n, n_cov)); cov[:, 0] = 1.0
    out = []
    for batch in (True, False):
        eng = engine.SketchEngine(n, n, max_entries=engine_capacity(plan, n), precision="i8x3", device=dev, sub_snps=engine_capacity(plan, n))
        eng.set_coding("value")
        runner = SketchRunner([eng], geno, plan, (0, jn), n_orig=n, missing_indv=[], Z=Z, cov=cov, seed=0, streaming=True, compat="fixed", dom=dom)
        runner.batch_pieces = batch
        runner.begin_trait(y, check_missing=False)
        pcs = runner._pieces(0)
        assert (len(pcs) == n_pieces) if batch else ((pcs is None) == (m // jn <= 34_308))
        runner.pass1()
        out.append(runner.pass2())
        eng.close()
        del runner
        torch.cuda.empty_cache()
    a, b = out
    assert np.array_equal(a.M, b.M)
    assert np.abs(a.T - b.T).max() <= 2e-6 * np.abs(b.T).max()
    assert np.allclose(a.q, b.q, rtol=2e-6)


def test_projector_kernel_vs_oracle_float32_chain():
    """pyrhe_engine_apply_projector (Base._compute_UXXz, base.py:407-408) against the oracle's float32
    mat_mul chain W (Q (W^T v)) and against a float64 evaluation."""
    from pyrhe_amd import engine
    dev = torch.device("cuda", 0)
    n, c, rows = 20_000, 11, 7
    rng = np.random.default_rng(8)
    W = np.concatenate([np.ones((n, 1)), rng.normal(size=(n, c - 1))], axis=1)
    Q = np.linalg.pinv(W.T @ W)
    v = rng.normal(size=(rows, n)) * 1000.0
    eng = engine.SketchEngine(n, n, 64, device=dev)
    n_pad = eng.n_pad
    x = torch.zeros((rows, n_pad), dtype=torch.float64, device=dev)
    x[:, :n] = torch.from_numpy(v).to(dev)
    Wt = torch.zeros((c, n_pad), dtype=torch.float32, device=dev)
    Wt[:, :n] = torch.from_numpy(W.T.astype(np.float32)).to(dev)
    got = eng.apply_projector(x, Wt, torch.from_numpy(Q.astype(np.float32)).to(dev)).cpu().numpy()
    assert not got[:, n:].any()
    exact = (W @ (Q @ (W.T @ v.T))).T
    scale = np.abs(exact).max()
    for r in range(rows):
        ref32 = O.project_U(v[r], W, Q)
        assert np.abs(got[r, :n] - ref32).max() <= 1e-5 * scale     # torch's float32 GEMV order over 20 000 samples: 2e-6 of noise
        assert np.abs(got[r, :n] - exact[r]).max() <= 1e-6 * scale
    eng.close()


@pytest.mark.parametrize("nbin,n_cov,with_missing,materialize", [(1, 0, False, False), (8, 10, False, False), (8, 10, False, True),
                                                                 (2, 0, True, False)])
def test_calls_beyond_4gb_of_rows_equal_three_block_calls(nbin, n_cov, with_missing, materialize):
    """PYRHE_OPT_LIST_SPAN: at N = 500k a call of more than 34 308 rows passes the 32-bit row offsets; the packed int8
    GEMM 2 addresses the lists of a workgroup relative to their lowest row, so the runner batches by entries alone (GEMM 1 pays
    a fixed cost per launch).  Four 10 000-SNP blocks (5 GB of rows) in ONE call against the same blocks at three per call:
    the per-block sketches are the same bits (exact int32 accumulation, per-list float32 epilogue)."""
    outs = _span_runs(500_000, 40_000, 4, nbin, n_cov, with_missing, materialize, (None, 34_000), (4, 3), 65536)
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_merged_lists_never_straddle_blocks_in_calls_beyond_4gb():
    """Blocks of 18 000 SNPs at N = 500k (2.25 GB each): ONE block fits the 32-bit row span, TWO do not.  Lists that share a
    workgroup of the packed GEMM 2 are addressed from their common lowest row, so in a two-block call the engine must never
    merge the last bins of block 0 with the first of block 1 (round 2's grouping did: offsets wrapped, wrong rows, silently
    wrong sketches).  Against the same blocks one per call."""
    outs = _span_runs(500_000, 36_000, 2, 8, 10, False, False, (None, 34_000), (2, 1), 65536)
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _span_runs(N, M, jn, nbin, n_cov, with_missing, materialize, limits, want_blocks_per_call, cap):
    from pyrhe_amd import engine
    from pyrhe_amd.streaming import BlockPlan, SketchRunner
    dev = torch.device("cuda", 0)
    k = 10
    outs = []
    for limit, want_b in zip(limits, want_blocks_per_call):
        eng = engine.SketchEngine(N, N, max_entries=cap, precision="i8x3", device=dev, sub_snps=cap)
        geno = engine.synth_genotypes(M, N, 31, 0, dev)
        if with_missing:                           # missing calls in every block: the kernels' fill path (row | fill << 24 lists)
            rows_m = torch.as_tensor(np.random.default_rng(3).choice(M, 500, replace=False), device=dev)
            geno[rows_m, 123] = (geno[rows_m, 123] & 0xF3) | 0x04
        geno = engine.recode_rows(geno, "value")
        eng.set_coding("value")
        annot = np.zeros((M, nbin), dtype=np.int64)
        annot[np.arange(M), np.random.default_rng(4).integers(0, nbin, M)] = 1
        plan = BlockPlan(annot, jn)
        Z = np.random.default_rng(5).standard_normal((N, k)).astype(np.float32)
        cov = None
        if n_cov:                                  # 8 bins + covariates: merged segments, four-tile GEMM 1
            cov = np.random.default_rng(7).standard_normal((N, n_cov))
            cov[:, 0] = 1.0
        runner = SketchRunner([eng], geno, plan, (0, jn), n_orig=N, missing_indv=[], Z=Z, cov=cov, seed=0, streaming=True,
                              compat="fixed" if n_cov else "reference", materialize_xxuz=materialize)
        runner.row_limit = limit
        y = np.random.default_rng(6).standard_normal(N)
        runner.begin_trait(y - y.mean(), check_missing=with_missing)
        assert min(runner._blocks_per_call(), jn) == want_b
        runner.pass1()
        torch.cuda.synchronize()
        path = eng.last_path()
        assert path["gemm2"] == (12 if (n_cov and materialize) else 11) and path["segmented"]
        assert (path["g2_merged"] > 1) == (nbin == 8)        # 5000-entry lists with 10 columns: two of them do not fit the LDS
        outs.append([t.clone() for t in runner.blocks_sk[0][:jn]] + [runner.blocks_nm[0][:jn].clone()] +
                    [t[:jn].clone() for t in runner.blocks_cr if t.shape[2]])
        eng.close()
        del runner, geno
        torch.cuda.empty_cache()
    return outs
