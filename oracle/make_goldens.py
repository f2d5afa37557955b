#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE in this container (never on the GPU box).

TEST INFRASTRUCTURE.  Usage (from the repo root, in the build container only):

    python oracle/make_goldens.py            # writes tests/golden/*.npz

How: the reference (``/root/reference``, pure Python) is imported from where it lies;
its one missing third-party dependency on this path, ``bed_reader`` (``pyrhe/setup.py:13``,
used at ``pyrhe/src/base/base.py:10,100,341,343``), is provided as an in-memory module backed
by ``oracle/plink.py`` (published PLINK1 layout + bed_reader's documented defaults:
float32, Fortran order, count_A1, NaN for missing).  Nothing from the reference is copied:
only *data* (inputs we generate + outputs the reference computes) is stored.

Reference modes used as oracle:
  * no covariates  -> non-streaming ``RHE(..., multiprocessing=False)``
  * covariates     -> non-streaming ``RHE(+cov)``  (= "fixed" semantics) and
                      ``StreamingRHE(+cov, multiprocessing=True, num_workers=1)`` (bug-compatible Q4)
  * GxE            -> non-streaming ``GENIE(genie_model="G+GxE", multiprocessing=False)``
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import fixtures, plink  # noqa: E402

REFERENCE = "/root/reference"
OUT_DIR = os.path.join(REPO, "tests", "golden")


def _install_bed_reader_standin():
    mod = types.ModuleType("bed_reader")

    class open_bed:  # noqa: N801  (name fixed by the reference's import)
        def __init__(self, path, *a, **kw):
            self.path = path
            prefix = path[:-4]
            with open(prefix + ".fam") as f:
                self.iid_count = sum(1 for _ in f)
            with open(prefix + ".bim") as f:
                self.sid_count = sum(1 for _ in f)
            self._packed = plink.read_bed_packed(path, self.iid_count, self.sid_count)

        def read(self, index=None, dtype="float32", order="F", **kw):
            rows, cols = index
            assert rows == slice(None, None, 1), rows
            sub = self._packed[cols]
            return plink.decode_count_a1(sub, self.iid_count)

    mod.open_bed = open_bed
    sys.modules["bed_reader"] = mod


def _import_reference():
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    _install_bed_reader_standin()
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    from pyrhe.src.models.rhe import RHE, StreamingRHE
    from pyrhe.src.models.genie import GENIE
    from pyrhe.src.models.rhe_dom import RHE_DOM
    from pyrhe.src.util.logger import Logger
    import pyrhe.src.models.genie.streaming_genie as sg
    import pyrhe.src.models.rhe_dom.streaming_rhe_dom as sd
    global _STREAMING_EXTRA
    _STREAMING_EXTRA = {"StreamingGENIE": sg.StreamingGENIE, "StreamingRHE_DOM": sd.StreamingRHE_DOM}
    return RHE, StreamingRHE, GENIE, Logger, RHE_DOM


CASES = [
    # name, cohort kwargs, model, model kwargs
    dict(name="rhe_1bin_nocov", cohort=dict(n_indv=403, n_snp=240, seed=11), model="RHE",
         kw=dict(num_jack=10, num_random_vec=10, seed=0)),
    dict(name="rhe_3bin_nocov_rem", cohort=dict(n_indv=403, n_snp=247, seed=12, nbin=3, overlap=True), model="RHE",
         kw=dict(num_jack=10, num_random_vec=10, seed=1)),
    dict(name="rhe_3bin_cov", cohort=dict(n_indv=403, n_snp=247, seed=13, nbin=3, overlap=True, with_cov=3), model="RHE",
         kw=dict(num_jack=10, num_random_vec=10, seed=2)),
    dict(name="srhe_3bin_cov", cohort=dict(n_indv=403, n_snp=247, seed=13, nbin=3, overlap=True, with_cov=3),
         model="StreamingRHE", kw=dict(num_jack=10, num_random_vec=10, seed=2)),
    # (StreamingRHE without covariates raises AttributeError in the reference's aggregate, base_streaming.py:29)
    dict(name="srhe_2bin_missing_cov", cohort=dict(n_indv=403, n_snp=240, seed=25, nbin=2, missing_geno=0.02, missing_pheno=5, with_cov=3),
         model="StreamingRHE", kw=dict(num_jack=10, num_random_vec=10, seed=3)),
    # num_random_vec beyond one engine pass (30 right-hand columns): the groups Z and UZ are split over three passes
    dict(name="rhe_2bin_cov_k35", cohort=dict(n_indv=403, n_snp=240, seed=26, nbin=2, with_cov=3), model="RHE",
         kw=dict(num_jack=10, num_random_vec=35, seed=14)),
    dict(name="genie_gxe_k33", cohort=dict(n_indv=403, n_snp=240, seed=27, nbin=1, with_env=True), model="GENIE",
         kw=dict(num_jack=10, num_random_vec=33, seed=15, genie_model="G+GxE+NxE")),
    dict(name="sgenie_gxe_nxe_cov", cohort=dict(n_indv=403, n_snp=240, seed=28, nbin=2, with_env=True, with_cov=3), model="StreamingGENIE",
         kw=dict(num_jack=10, num_random_vec=10, seed=16, genie_model="G+GxE+NxE")),
    dict(name="srhe_dom_cov", cohort=dict(n_indv=403, n_snp=240, seed=29, nbin=2, with_cov=3), model="StreamingRHE_DOM",
         kw=dict(num_jack=10, num_random_vec=10, seed=17)),
    dict(name="rhe_1bin_missing", cohort=dict(n_indv=403, n_snp=240, seed=14, missing_geno=0.02, missing_pheno=5),
         model="RHE", kw=dict(num_jack=10, num_random_vec=10, seed=3)),
    dict(name="rhe_1bin_missing_mean", cohort=dict(n_indv=403, n_snp=240, seed=14, missing_geno=0.02, missing_pheno=5),
         model="RHE", kw=dict(num_jack=10, num_random_vec=10, seed=3, geno_impute_method="mean")),
    dict(name="rhe_1bin_n1024", cohort=dict(n_indv=1024, n_snp=1000, seed=15), model="RHE",
         kw=dict(num_jack=10, num_random_vec=10, seed=4)),
    dict(name="rhe_2bin_jn1", cohort=dict(n_indv=403, n_snp=240, seed=16, nbin=2), model="RHE",
         kw=dict(num_jack=1, num_random_vec=7, seed=5)),
    dict(name="rhe_1bin_2traits", cohort=dict(n_indv=403, n_snp=240, seed=17, n_traits=2), model="RHE",
         kw=dict(num_jack=5, num_random_vec=10, seed=6)),
    dict(name="genie_gxe_2bin", cohort=dict(n_indv=403, n_snp=240, seed=18, nbin=2, with_env=True), model="GENIE",
         kw=dict(num_jack=10, num_random_vec=10, seed=7, genie_model="G+GxE")),
    dict(name="genie_gxe_nxe_2bin", cohort=dict(n_indv=403, n_snp=240, seed=20, nbin=2, with_env=True), model="GENIE",
         kw=dict(num_jack=10, num_random_vec=10, seed=9, genie_model="G+GxE+NxE")),
    dict(name="genie_gxe_nxe_cov", cohort=dict(n_indv=403, n_snp=240, seed=21, nbin=1, with_env=True, with_cov=3), model="GENIE",
         kw=dict(num_jack=10, num_random_vec=10, seed=10, genie_model="G+GxE+NxE")),
    # genie_model="G" has no reference behaviour: GENIE.aggregate indexes env[:, -1 - 2*nbin] and raises IndexError
    dict(name="rhe_dom_2bin", cohort=dict(n_indv=403, n_snp=240, seed=23, nbin=2), model="RHE_DOM",
         kw=dict(num_jack=10, num_random_vec=10, seed=12)),
    dict(name="rhe_dom_cov", cohort=dict(n_indv=403, n_snp=240, seed=24, nbin=1, with_cov=3), model="RHE_DOM",
         kw=dict(num_jack=10, num_random_vec=10, seed=13)),
    dict(name="genie_gxe_cov", cohort=dict(n_indv=403, n_snp=240, seed=19, nbin=1, with_env=True, with_cov=3), model="GENIE",
         kw=dict(num_jack=10, num_random_vec=10, seed=8, genie_model="G+GxE")),
]


def run_case(case, classes):
    RHE, StreamingRHE, GENIE, Logger, RHE_DOM = classes
    cohort = fixtures.make_cohort(case["name"], **case["cohort"])
    out = fixtures.cohort_to_npz_dict(cohort)
    kw = dict(case["kw"])
    with tempfile.TemporaryDirectory() as tmp:
        paths = fixtures.write_cohort_files(cohort, tmp)
        log = Logger(output_file=None, suppress=True, debug_mode=False)
        params = dict(model="rhe", geno_file=paths["geno"], annot_file=paths["annot"], pheno_file=paths["pheno"],
                      cov_file=paths.get("cov"), device="cpu", log=log, **{k: v for k, v in kw.items() if k != "genie_model"})
        streaming = case["model"].startswith("Streaming")
        if case["model"] == "StreamingGENIE":
            params.update(multiprocessing=True, num_workers=1, env_file=paths["env"], genie_model=kw["genie_model"], model="genie")
            cls = _STREAMING_EXTRA["StreamingGENIE"]
        elif case["model"] == "StreamingRHE_DOM":
            params.update(multiprocessing=True, num_workers=1, model="rhe_dom")
            cls = _STREAMING_EXTRA["StreamingRHE_DOM"]
        elif streaming:
            params.update(multiprocessing=True, num_workers=1)
            cls = StreamingRHE
        elif case["model"] == "GENIE":
            params.update(multiprocessing=False, num_workers=1, env_file=paths["env"], genie_model=kw["genie_model"], model="genie")
            cls = GENIE
        elif case["model"] == "RHE_DOM":
            params.update(multiprocessing=False, num_workers=1, model="rhe_dom")
            cls = RHE_DOM
        else:
            params.update(multiprocessing=False, num_workers=1)
            cls = RHE
        cwd = os.getcwd()
        os.chdir(tmp)  # the reference may drop side files in CWD
        try:
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                model = cls(**params)
                out["ref_all_zb"] = model.all_zb.copy()
                if model.use_cov:
                    out["ref_all_Uzb"] = model.all_Uzb.copy()
                    out["ref_cov_matrix"] = np.asarray(model.cov_matrix, dtype=np.float64).copy()
                out["ref_missing_indv"] = np.asarray(model.missing_indv, dtype=np.int64)
                out["ref_len_bin"] = np.asarray(model.len_bin, dtype=np.int64)
                n_traits = model.num_traits
                for t in range(n_traits):
                    rec = {"T": [], "q": []}
                    if not streaming:
                        orig_setup = model.setup_lhs_rhs_jackknife
                        orig_agg = model.aggregate

                        def setup(j, trace_sums, is_streaming=False, _o=orig_setup, _r=rec):
                            T, q = _o(j, trace_sums, is_streaming)
                            _r["T"].append(T.copy()); _r["q"].append(q.copy())
                            return T, q

                        def agg(_o=orig_agg, _r=rec, _m=model):
                            _r["XXz_blocks"] = _m.XXz[:, :_m.num_jack].astype(np.float32)
                            _r["yXXy_blocks"] = _m.yXXy[:, :_m.num_jack].copy()
                            if _m.use_cov:
                                _r["XXUz_blocks"] = _m.XXUz[:, :_m.num_jack].astype(np.float32)
                                _r["UXXz_blocks"] = _m.UXXz[:, :_m.num_jack].astype(np.float32)
                            _o()
                            _r["XXz_total"] = _m.XXz[:, _m.num_jack].copy()
                            _r["yXXy_total"] = _m.yXXy[:, _m.num_jack].copy()
                            if _m.use_cov:
                                _r["XXUz_total"] = _m.XXUz[:, _m.num_jack].copy()
                                _r["UXXz_total"] = _m.UXXz[:, _m.num_jack].copy()

                        model.setup_lhs_rhs_jackknife = setup
                        model.aggregate = agg
                    orig_est = model.estimate

                    def est(method="lstsq", _o=orig_est, _r=rec):
                        res = _o(method=method)
                        _r["sigma_jack"] = np.asarray(res[0]); _r["sigma_total"] = np.asarray(res[1])
                        if len(res) == 4:
                            _r["sigma_jack_adj"] = np.asarray(res[2]); _r["sigma_total_adj"] = np.asarray(res[3])
                        return res

                    model.estimate = est
                    n_msgs = len(log.msgs)
                    res = model(trait=t)
                    pre = f"ref_t{t}_"
                    out[pre + "M"] = np.asarray(model.M, dtype=np.int64).copy()
                    for k2, v in rec.items():
                        out[pre + k2] = np.asarray(v)
                    for k2, v in res.items():
                        out[pre + "res_" + k2] = np.asarray(v, dtype=np.float64)
                    out[pre + "log"] = np.array("".join(log.msgs[n_msgs:]))
                    if not streaming:
                        model.setup_lhs_rhs_jackknife = orig_setup
                        model.aggregate = orig_agg
                    model.estimate = orig_est
                out["ref_log_init"] = np.array("".join(log.msgs[:0]))
                if hasattr(model, "_finalize"):
                    model._finalize()
        finally:
            os.chdir(cwd)
    # params (data, not code)
    for k2, v in kw.items():
        out["param_" + k2] = np.array(v)
    out["param_model"] = np.array(case["model"])
    return out


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    classes = _import_reference()
    only = set(sys.argv[1:])
    for case in CASES:
        if only and case["name"] not in only:
            continue
        out = run_case(case, classes)
        path = os.path.join(OUT_DIR, case["name"] + ".npz")
        np.savez_compressed(path, **out)
        sig = out["ref_t0_res_sigma_ests_total"]
        print(f"{case['name']}: sigma={sig} -> {path} ({os.path.getsize(path)/1024:.0f} KiB)")


if __name__ == "__main__":
    main()
