"""The driver's multi-GPU launch line rehearsed on ONE GPU: two ranks (gloo instead of RCCL, which refuses two ranks on one
device) shard the jackknife blocks, all-reduce the accumulators, gather the leave-one-out systems -- sigma^2 and SEs must
equal the single-process run of the same cohort."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--n-indv", "20000", "--n-snp", "20000", "--jn", "10", "--steps", "1", "--warmup", "1", "--verify",
        "--no-cpu-baseline", "--no-alt"]


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(out: str) -> dict:
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("extra", [[], ["--n-bin", "4", "--n-cov", "3"],
                                   ["--n-bin", "2", "--n-cov", "2", "--missing-rate", "0.01", "--drop-indv", "37"]])
def test_bench_two_ranks_equal_one_rank(extra):
    env = dict(os.environ, PYRHE_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + ARGS + extra, cwd=REPO, env=env, capture_output=True,
                         text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_port()), "bench.py", "--gpus", "2"] + ARGS + extra,
                         cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = _last_json(one.stdout), _last_json(two.stdout)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["scaling"] == "strong"
    assert b["config"]["dist"]["world"] == 2 and sum(b["config"]["dist"]["blocks_per_rank"]) == 10 and b["config"]["dist"]["backend"] == "gloo"
    # the cross-rank float64 sum changes the summation order only
    assert np.allclose(a["config"]["sigma_total"], b["config"]["sigma_total"], rtol=1e-9)
    assert np.allclose(a["config"]["sigma_se"], b["config"]["sigma_se"], rtol=1e-7)


def test_bench_four_ranks_uneven_shards_equal_one_rank():
    """Four ranks on one GPU (gloo), jn = 10: shards of 3, 3, 3, 1 blocks -- the driver's launch line with an uneven split."""
    env = dict(os.environ, PYRHE_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    args = ["--n-indv", "12000", "--n-snp", "10000", "--jn", "10", "--steps", "1", "--warmup", "0", "--verify", "--no-cpu-baseline",
            "--no-alt", "--n-bin", "2", "--n-cov", "3"]
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + args, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    four = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                           "127.0.0.1", "--master-port", str(_port()), "bench.py", "--gpus", "4"] + args, cwd=REPO, env=env,
                          capture_output=True, text=True, timeout=900)
    assert four.returncode == 0, four.stderr[-2000:]
    a, b = _last_json(one.stdout), _last_json(four.stdout)
    assert b["n_gpus"] == 4
    assert np.allclose(a["config"]["sigma_total"], b["config"]["sigma_total"], rtol=1e-9)
    assert np.allclose(a["config"]["sigma_se"], b["config"]["sigma_se"], rtol=1e-7)


def test_two_ranks_without_a_seed_agree(tmp_path):
    """seed=None (run_rhe.py's default) under a multi-rank launch: rank 0's clock-derived seed is broadcast, so every rank
    draws the same probe vectors and generated annotation (written once, by rank 0); both ranks report the same estimates."""
    from oracle import fixtures
    c = fixtures.make_cohort("noseed", 500, 360, seed=9)
    paths = fixtures.write_cohort_files(c, str(tmp_path))
    env = dict(os.environ, PYRHE_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_port()), os.path.join(REPO, "tests", "dist_model_worker.py"),
                        str(tmp_path), paths["geno"], paths["pheno"]], cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = (json.load(open(tmp_path / f"rank{i}.json")) for i in (0, 1))
    assert a["seed"] == b["seed"] and a["zsum"] == b["zsum"] and a["len_bin"] == b["len_bin"] and sum(a["len_bin"]) == 360
    assert a["sigma"] == b["sigma"] and a["se"] == b["se"]
    assert np.all(np.isfinite(a["sigma"])) and np.all(np.isfinite(a["se"]))


def test_rccl_executes_every_collective_of_the_path_on_one_rank():
    """RCCL itself on the GPU box: the driver's launch line with ONE rank under the "nccl" backend (= RCCL on ROCm) and
    ``--force-collectives``, so that the in-place float64 all-reduce of every bucket, the (jn - j)-weighted ones of quirk Q4's
    totals, the packed norm / cross-Gram all-reduce, the tensor all-gather of the leave-one-out systems, barrier(device_ids) and
    the MAX all-reduce of the timing all go through RCCL (base_streaming.py:21-58's aggregate is what they stand for).  Same
    sigma^2 and SE as the run without a process group, and the line says what the collectives saw."""
    args = ["--n-indv", "20000", "--n-snp", "12000", "--jn", "10", "--steps", "1", "--warmup", "1", "--verify", "--no-cpu-baseline",
            "--no-alt", "--n-bin", "3", "--n-cov", "4", "--compat", "reference"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("PYRHE_DIST_BACKEND", None)
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + args, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    rccl = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                           "127.0.0.1", "--master-port", str(_port()), "bench.py", "--gpus", "1", "--force-collectives"] + args,
                          cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert rccl.returncode == 0, rccl.stderr[-3000:]
    a, b = _last_json(one.stdout), _last_json(rccl.stdout)
    assert "dist" not in a["config"]
    assert b["config"]["dist"] == {"backend": "nccl", "world": 1, "blocks_per_rank": [10], "collectives_forced": True}
    assert np.allclose(a["config"]["sigma_total"], b["config"]["sigma_total"], rtol=1e-12)
    assert np.allclose(a["config"]["sigma_se"], b["config"]["sigma_se"], rtol=1e-10)
