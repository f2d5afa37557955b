"""CPU-side tests: the C-ABI library loads and exports every symbol the header declares (no compute calls),
host logic (block plan, sharding, imputation replay, parsers, estimator) against the oracle and the goldens."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import fixtures
from oracle import rhe_oracle as O
from tests.helpers import RESULT_KEYS_RHE, RHE_CASES, load_golden, oracle_problem

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from pyrhe_amd.build import build_library
    return build_library()


def test_library_exports_every_header_symbol(lib_path):
    header = open(os.path.join(REPO, "include", "pyrhe_sketch.h")).read()
    declared = set(re.findall(r"\b(pyrhe_[a-z_0-9]+)\s*\(", header))
    declared.discard("pyrhe_engine_t")
    assert len(declared) >= 12
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/pyrhe_sketch.h but not exported"
    # the Python binding covers exactly the same set
    from pyrhe_amd import engine
    assert set(engine.SYMBOLS) == declared
    bound = engine.load_library()
    assert bound.pyrhe_abi_version() == 1
    assert bound.pyrhe_last_error() is not None


def test_engine_refuses_to_run_without_gpu():
    import torch
    from pyrhe_amd import engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no ROCm GPU"):
        engine.SketchEngine(100, 100, 10)
    with pytest.raises(RuntimeError):
        engine.upload_packed(np.zeros((2, 25), np.uint8), "cuda")


def test_product_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "pyrhe_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_block_plan_matches_oracle_indexing():
    from pyrhe_amd.streaming import BlockPlan, block_range, shard_ranges
    for m, jn in [(240, 10), (247, 10), (1000, 7), (240, 1), (13, 13)]:
        for j in range(jn):
            assert block_range(j, m, jn) == O.block_range(j, m, jn)
    assert shard_ranges(100, 8) == O.distribute_work(100, 8)
    assert shard_ranges(3, 8)[3:] == [(3, 3)] * 5            # more GPUs than blocks: empty shards
    z = load_golden("rhe_3bin_nocov_rem")
    p, c = oracle_problem(z)
    plan = BlockPlan(c.annot, p.n_jack)
    M = np.zeros((p.n_jack + 1, p.nbin), np.int64)
    M[-1] = plan.len_bin
    for j in range(p.n_jack):
        ent, offs = plan.entries(j)
        s, e = plan.range(j)
        for b in range(p.nbin):
            assert np.array_equal(ent[offs[b]:offs[b + 1]], np.nonzero(c.annot[s:e, b])[0])
        M[j] = M[-1] - plan.block_counts(j)
    assert np.array_equal(M, z["ref_t0_M"])                    # bin membership incl. overlapping / uncovered SNPs


def test_hwe_fill_values_replay_numpy_stream():
    """Product-side imputation replay == the oracle's per-SNP loop (reference base.py:265-289)."""
    from pyrhe_amd.streaming import hwe_fill_values
    z = load_golden("rhe_1bin_missing")
    p, c = oracle_problem(z)
    for j in range(p.n_jack):
        s, e = O.block_range(j, p.n_snp, p.n_jack)
        X = O.read_geno(p.packed, p.n_indv_orig, s, e, p.missing_indv)
        miss = np.isnan(X)
        np.random.seed(p.seed)
        Xi = O.impute_geno(X.copy(), "binary")
        sums = np.nansum(X, axis=0).astype(np.int64)
        nobs = (~miss).sum(axis=0)
        u = np.random.RandomState(p.seed).random_sample(e - s)
        f = hwe_fill_values(sums, nobs, u)
        for m in range(e - s):
            if miss[:, m].any():
                assert f[m] == Xi[miss[:, m], m][0]


def test_parsers_match_reference_parsed_values(tmp_path):
    from pyrhe_amd import io as rio
    z = load_golden("rhe_3bin_cov")
    c = fixtures.cohort_from_npz(z)
    paths = fixtures.write_cohort_files(c, str(tmp_path))
    n, fam = rio.read_fam(paths["geno"] + ".fam")
    assert n == c.n_indv and rio.count_bim(paths["geno"] + ".bim") == c.n_snp
    nbin, annot, len_bin = rio.read_annot(paths["annot"])
    assert nbin == 3 and np.array_equal(annot, c.annot) and np.array_equal(len_bin, z["ref_len_bin"])
    y, missing, binary = rio.read_pheno(paths["pheno"])
    assert missing == [] and not binary and np.array_equal(y, c.pheno)
    cov, miss2 = rio.read_cov(paths["cov"], missing_indvs=missing)
    assert np.array_equal(cov, z["ref_cov_matrix"]) and miss2 == []
    packed = rio.read_bed_packed(paths["geno"] + ".bed", c.n_indv, c.n_snp)
    assert np.array_equal(np.asarray(packed), c.packed)
    # missing phenotypes: NA rows become missing individuals
    z2 = load_golden("rhe_1bin_missing")
    c2 = fixtures.cohort_from_npz(z2)
    p2 = fixtures.write_cohort_files(c2, str(tmp_path / "m"))
    _, missing2, _ = rio.read_pheno(p2["pheno"])
    assert missing2 == [int(i) for i in z2["ref_missing_indv"]]
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.bed"
        bad.write_bytes(b"\x00\x01\x02" + b"\x00" * 10)
        rio.read_bed_packed(str(bad), 4, 1)


@pytest.mark.parametrize("name", RHE_CASES)
def test_estimator_reproduces_reference_from_its_own_systems(name):
    """Host estimator (solve, jackknife SE, h2, overlapping h2, enrichment) fed the reference's T, q."""
    from pyrhe_amd import estimate as est
    z = load_golden(name)
    p, c = oracle_problem(z)
    T, q, M = z["ref_t0_T"], z["ref_t0_q"], z["ref_t0_M"]
    jn, ne = p.n_jack, p.nbin
    sig = np.array([est.solve_normal_equations(T[j], q[j]) for j in range(jn + 1)])
    sj, st = sig[:-1], sig[-1]
    assert np.allclose(sj, z["ref_t0_sigma_jack"], rtol=1e-12, atol=0)
    h2j, h2t = est.h2_nonoverlapping(sj, st, ne)
    ej, et = est.enrichment(h2j, h2t, M, ne)
    hoj, hot = est.h2_overlapping(sj, st, ne, c.annot, M, p.n_snp, jn)
    eoj, eot = est.enrichment(hoj, hot, M, ne)
    got = {"sigma_ests_total": st, "sig_errs": est.jackknife_se(sj, jn), "h2_total": h2t, "h2_errs": est.jackknife_se(h2j, jn),
           "enrichment_total": et, "enrichment_errs": est.jackknife_se(ej, jn), "h2_total_overlap": hot,
           "h2_errs_overlap": est.jackknife_se(hoj, jn), "enrichment_total_overlap": eot,
           "enrichment_errs_overlap": est.jackknife_se(eoj, jn)}
    for k in RESULT_KEYS_RHE:
        assert np.allclose(got[k], z["ref_t0_res_" + k], rtol=1e-10, atol=1e-13), k


def test_gemm_form_emulation_bounds_the_precision_modes():
    """The device algorithm (raw genotypes + rank-1 corrections) emulated in NumPy: hi+lo fp16 reaches float32-level
    agreement with the reference-order sketch, one fp16 rounding ~3e-4 per entry."""
    from oracle import gemm_form as GF
    z = load_golden("rhe_1bin_n1024")
    p, c = oracle_problem(z)
    ref = z["ref_t0_XXz_blocks"][0, 0].astype(np.float64)          # block 0: [k, N]
    s, e = O.block_range(0, p.n_snp, p.n_jack)
    R = np.concatenate([p.Z, p.pheno], axis=1)
    for mode, tol in (("f16x2", 2e-6), ("f16", 1e-3), ("f32", 2e-6)):
        V, A = GF.sketch_gemm_form(c.geno[s:e].astype(np.int64), R, p.k, mode)
        assert np.abs(A.T - ref).max() <= tol * np.abs(ref).max(), mode
        yx = float((V[:, p.k].astype(np.float64) ** 2).sum())
        assert abs(yx - z["ref_t0_yXXy_blocks"][0, 0]) <= 2 * tol * z["ref_t0_yXXy_blocks"][0, 0]


def test_solve_all_equals_per_system_qr_and_keeps_error_behaviour():
    """estimate.solve_all (one stacked QR + vectorised back-substitution) against the reference's per-system
    scipy QR + solve_triangular (base.py:306-312); a singular system surfaces the same LinAlgError."""
    from pyrhe_amd import estimate as est
    rng = np.random.default_rng(3)
    for n in (2, 3, 9, 17):
        A = rng.normal(size=(101, n, n))
        T = A @ A.transpose(0, 2, 1) + 0.1 * np.eye(n)
        q = rng.normal(size=(101, n))
        ref = np.array([est.solve_normal_equations(T[j], q[j].reshape(-1, 1)) for j in range(101)])
        got = est.solve_all(T, q)
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
        assert np.allclose(est.solve_all(T, q, "lstsq"), ref, rtol=1e-8)
    with pytest.raises(np.linalg.LinAlgError):
        est.solve_all(np.zeros((3, 2, 2)), np.ones((3, 2)))


def test_annot_fast_reader_equals_the_reference_line_loop(tmp_path):
    """read_annot parses a plain annotation file with pandas' C reader and everything else with the reference's own line loop
    (file_processing.py:37-69): same matrix, dtype and counts on a plain file, with '#' lines, with one column; lines the C
    reader does not take (trailing spaces) fall through to the loop; a blank line raises as it does in the reference."""
    from pyrhe_amd import io as rio
    rng = np.random.default_rng(0)
    mat = np.zeros((500, 5), np.int64)
    mat[np.arange(500), rng.integers(0, 5, 500)] = 1
    mat[::7, 2] = 1                                                   # overlapping annotations
    body = ["# header comment"] + [" ".join(str(v) for v in row) for row in mat]
    body.insert(100, "# a comment in the middle")
    cases = {"plain": "\n".join(body[1:100] + body[101:]) + "\n", "comments": "\n".join(body) + "\n",
             "one_column": "\n".join(str(v) for v in mat[:, 0]) + "\n",
             "trailing_spaces": "\n".join(l + " " for l in body[1:100]) + "\n"}
    for name, text in cases.items():
        p = tmp_path / (name + ".annot")
        p.write_text(text)
        a, b = rio.read_annot(str(p)), rio._read_annot_lines(str(p))
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[1].dtype == b[1].dtype and np.array_equal(a[2], b[2]), name
    assert np.array_equal(rio.read_annot(str(tmp_path / "comments.annot"))[1], mat)
    p = tmp_path / "blank.annot"
    p.write_text("1 0\n\n0 1\n")
    with pytest.raises(ValueError):
        rio.read_annot(str(p))
