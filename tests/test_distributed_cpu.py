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
    from pyrhe_amd.streaming import n_engines_needed
    has_gxe = p.genie_model in ("G+GxE", "G+GxE+NxE")
    eng = [CpuStubEngine(c.n_indv, p.n_indv) for _ in range(n_engines_needed(p.k, p.cov is not None, has_gxe))]
    r0 = plan.range(j0)[0] if j1 > j0 else 0
    r1 = plan.range(j1 - 1)[1] if j1 > j0 else 0
    geno = torch.zeros((r1 - r0, eng[0].pitch), dtype=torch.uint8)
    geno[:, : c.packed.shape[1]] = torch.from_numpy(c.packed[r0:r1].copy())
    env = None if p.genie_model is None else np.delete(np.asarray(p.env, np.float64).reshape(-1), p.missing_indv)
    runner = SketchRunner(eng, geno, plan, (j0, j1), n_orig=c.n_indv, missing_indv=p.missing_indv, Z=p.Z, cov=p.cov,
                          impute=p.impute, seed=p.seed, streaming=streaming, compat=compat, dom=bool(getattr(p, "dom", False)),
                          env=env, genie_model=p.genie_model)
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
                                                  ("rhe_1bin_missing", "reference", False),
                                                  ("srhe_dom_cov", "reference", True),
                                                  ("genie_gxe_nxe_cov", "reference", False)])   # GENIE G + GxE + NxE + covariates: a second, cross-only engine pass per rank      # RHE_DOM + covariates: both encodings' totals, cross Grams and quirk Q4's weighted sums exchanged
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


# ----------------------------------------------------------------------------------------------------------------------
# More ranks than the golden fixtures exercise: uneven shards (jn = 100 over 8 ranks: 13, ..., 13, 9), ranks without a
# block (world > jn), covariates with the reference-compatible (weighted) UXXz -- all against the one-process run.
# ----------------------------------------------------------------------------------------------------------------------
def _synth_runner(n, m, jn, nbin, with_cov, compat, blocks_per_call=0, dom=False):
    from oracle import fixtures
    from oracle import rhe_oracle as O
    from pyrhe_amd.streaming import BlockPlan, SketchRunner, shard_ranges
    from tests.cpu_stub_engine import CpuStubEngine
    c = fixtures.make_cohort("dist", n, m, seed=77, nbin=nbin, with_cov=with_cov)
    y = c.pheno[:, 0] - c.pheno[:, 0].mean()
    plan = BlockPlan(c.annot, jn)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    j0, j1 = shard_ranges(jn, world)[rank]
    eng = CpuStubEngine(n, n)
    r0 = plan.range(j0)[0] if j1 > j0 else 0
    r1 = plan.range(j1 - 1)[1] if j1 > j0 else 0
    geno = torch.zeros((r1 - r0, eng.pitch), dtype=torch.uint8)
    geno[:, : c.packed.shape[1]] = torch.from_numpy(c.packed[r0:r1].copy())
    Z = O.make_probe_vectors(0, n, 4)
    runner = SketchRunner(eng, geno, plan, (j0, j1), n_orig=n, missing_indv=[], Z=Z, cov=c.cov, seed=0, streaming=True,
                          compat=compat, dom=bool(dom))
    if blocks_per_call:          # several engine calls per rank: one round of collectives per call (SketchRunner.pass1)
        runner.row_limit = blocks_per_call * plan.max_block
    return runner, y, (j0, j1)


def _synth_worker(rank, world, port, shape, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        runner, y, local = _synth_runner(*shape)
        sysm = runner.run_trait(y)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), T=sysm.T, q=sysm.q, M=sysm.M, local=np.asarray(local),
                 rounds=int(runner.timings.get("exchange_rounds", 0)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(4, (96, 300, 100, 2, 0, "reference")),
                                         (8, (96, 300, 100, 3, 3, "reference")),      # 13 x 7 + 9 blocks, weighted UXXz (Q4)
                                         (8, (96, 300, 100, 1, 3, "fixed")),
                                         (4, (64, 90, 3, 2, 2, "reference")),         # world > jn: one rank has no block
                                         (8, (64, 100, 10, 1, 0, "reference")),       # ceil rule: ranks 5..7 empty
                                         # four blocks per engine call: 13-block ranks exchange in 4 rounds (4, 4, 4, 1), the 9-block rank's
                                         # fourth round carries zeros; quirk Q4's weighted totals travel per call too
                                         (8, (96, 300, 100, 2, 3, "reference", 4)),
                                         (4, (64, 90, 3, 2, 2, "fixed", 1)),          # a rank without blocks in a 1-round exchange
                                         # 50 blocks per rank in calls of 2 = 25 calls: grouped into 4 rounds of 7, 7, 7, 4 calls
                                         # (SketchRunner.MAX_EXCHANGE_ROUNDS: buffers and wire bytes do not grow with the calls)
                                         (2, (96, 300, 100, 2, 3, "reference", 2)),
                                         # RHE_DOM with covariates in the reference-compatible mode: plain and (jn - j)-weighted
                                         # partial totals of both encodings from one read per round (sum_blocks_both)
                                         (3, (64, 120, 10, 2, 2, "reference", 2, True))])
def test_many_ranks_uneven_and_empty_shards_equal_one_rank(world, shape, tmp_path):
    from pyrhe_amd.streaming import shard_ranges
    mp.spawn(_synth_worker, args=(world, _free_port(), shape, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    jn = shape[2]
    assert [tuple(p["local"]) for p in parts] == shard_ranges(jn, world)
    sizes = [b - a for a, b in shard_ranges(jn, world)]
    assert sum(sizes) == jn and (world <= jn or 0 in sizes)
    for p in parts[1:]:
        assert np.array_equal(parts[0]["T"], p["T"]) and np.array_equal(parts[0]["q"], p["q"])     # every rank holds all systems
    from pyrhe_amd.streaming import SketchRunner
    per_call = shape[6] if len(shape) > 6 else None
    if len(shape) > 7 and shape[7]:
        per_call = 1                 # RHE_DOM on the stand-in engine: one block per call (the segmented flow is the int8 engine's)
    if per_call:
        calls = -(-max(sizes) // per_call)
        want = -(-calls // -(-calls // SketchRunner.MAX_EXCHANGE_ROUNDS))
        assert [int(p["rounds"]) for p in parts] == [want] * world and want <= SketchRunner.MAX_EXCHANGE_ROUNDS
    runner, y, _ = _synth_runner(*shape)                                                            # world = 1 in this process
    one = runner.run_trait(y)
    assert np.array_equal(one.M, parts[0]["M"])
    assert np.allclose(one.T, parts[0]["T"], rtol=1e-11, atol=0) and np.allclose(one.q, parts[0]["q"], rtol=1e-12, atol=0)
