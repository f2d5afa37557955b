"""world_size-2 gloo test of the multi-GPU path on CPU: jackknife blocks sharded by the reference's contiguous
rule, ONE all-reduce of the per-bin accumulators, leave-one-out systems built where the block lives and gathered.
The sketches come from a CPU stand-in engine built on oracle math (tests/cpu_stub_engine.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import load_golden, oracle_problem


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build_runner(name, rank_blocks=None, compat="reference", streaming=True):
    from pyrhe_amd.streaming import BlockPlan, SketchRunner, shard_ranges
    from tests.cpu_stub_engine import CpuStubEngine
    z = load_golden(name)
    p, c = oracle_problem(z)
    plan = BlockPlan(c.annot, p.n_jack)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    j0, j1 = shard_ranges(p.n_jack, world)[rank]
    eng = CpuStubEngine(c.n_indv, p.n_indv)
    r0 = plan.range(j0)[0] if j1 > j0 else 0
    r1 = plan.range(j1 - 1)[1] if j1 > j0 else 0
    geno = torch.zeros((r1 - r0, eng.pitch), dtype=torch.uint8)
    geno[:, : c.packed.shape[1]] = torch.from_numpy(c.packed[r0:r1].copy())
    runner = SketchRunner(eng, geno, plan, (j0, j1), n_orig=c.n_indv, missing_indv=p.missing_indv, Z=p.Z, cov=p.cov,
                          impute=p.impute, seed=p.seed, streaming=streaming, compat=compat)
    return runner, p, z


def _worker(rank, world, port, name, compat, streaming, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        runner, p, z = _build_runner(name, compat=compat, streaming=streaming)
        sysm = runner.run_trait(p.pheno[:, 0])
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), T=sysm.T, q=sysm.q, M=sysm.M)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,compat,streaming", [("rhe_3bin_nocov_rem", "reference", True),
                                                  ("rhe_3bin_cov", "fixed", True),
                                                  ("rhe_1bin_missing", "reference", False)])
def test_two_ranks_equal_one_rank_and_reference(name, compat, streaming, tmp_path):
    from pyrhe_amd import estimate as est
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, compat, streaming, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "r0.npz")
    r1 = np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["T"], r1["T"]) and np.array_equal(r0["q"], r1["q"])         # every rank holds all systems
    runner, p, z = _build_runner(name, compat=compat, streaming=streaming)               # world = 1 in this process
    one = runner.run_trait(p.pheno[:, 0])
    assert np.array_equal(one.M, r0["M"]) and np.array_equal(one.M, z["ref_t0_M"])
    assert np.allclose(one.T, r0["T"], rtol=1e-12) and np.allclose(one.q, r0["q"], rtol=1e-12)
    # and the sharded result reproduces the reference's variance components (stub = float64 maths: 2e-6)
    sig = np.array([est.solve_normal_equations(r0["T"][j], r0["q"][j].reshape(-1, 1)) for j in range(p.n_jack + 1)])
    assert np.allclose(sig[-1], z["ref_t0_res_sigma_ests_total"], rtol=2e-6)
    assert np.allclose(est.jackknife_se(sig[:-1], p.n_jack), z["ref_t0_res_sig_errs"], rtol=2e-5)


def test_streaming_q4_weighted_form_close_to_sequential(tmp_path):
    """Bug-compatible streaming + covariates across 2 ranks uses U(sum (jn-j) XXz_j); single rank applies U to the
    running sum block by block like the reference.  Same value up to float32 rounding order."""
    from pyrhe_amd import estimate as est
    name = "srhe_3bin_cov"
    mp.spawn(_worker, args=(2, _free_port(), name, "reference", True, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npz")
    z = load_golden(name)
    jn = int(z["param_num_jack"])
    sig = np.array([est.solve_normal_equations(r0["T"][j], r0["q"][j].reshape(-1, 1)) for j in range(jn + 1)])
    assert np.allclose(sig[-1], z["ref_t0_res_sigma_ests_total"], rtol=1e-5)
    assert np.allclose(sig[:-1], z["ref_t0_sigma_jack"], rtol=1e-5)
