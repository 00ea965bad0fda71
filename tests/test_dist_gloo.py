"""world_size-2 gloo test of the sharded mean log-prob (the N>1 path of bench.py).

The per-rank evaluation here is the CPU oracle (this is a CPU test of the sharding and the
reduction, not of the kernels); on the GPU box the same driver gets the HIP epilogue's sum."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import recipes
from helpers import c2_layers


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, rows, out):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import flow_oracle as O
    from torch_mnf_amd.dist import shard_bounds, sharded_mean_log_prob, sharded_mean_log_prob_async

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    layers = c2_layers(64, 3)
    x = recipes.gaussian(99, rows, 64)
    lo, hi = shard_bounds(rows, world, rank)

    def local_sum(xl):
        _, lp = O.mean_log_prob(xl, layers)
        return lp.double().sum().reshape(1)

    mean = sharded_mean_log_prob(local_sum, x[lo:hi])
    # pipelined form (bench.py's N > 1 step): the reduction of batch k is collected after batch k + 1 is enqueued
    first = sharded_mean_log_prob_async(local_sum, x[lo:hi])
    second = sharded_mean_log_prob_async(local_sum, 2 * x[lo:hi])
    piped = (float(first.result()), float(second.result()))
    if rank == 0:
        out.put((float(mean), lo, hi, piped))
    dist.destroy_process_group()


def test_sharded_mean_matches_single_process():
    from oracle import flow_oracle as O

    rows, world = 1001, 2  # odd: uneven shards
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    mean, lo, hi, piped = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref, _ = O.mean_log_prob(recipes.gaussian(99, rows, 64), c2_layers(64, 3))
    assert (lo, hi) == (0, 501)
    assert abs(mean - ref) <= 1e-9 * abs(ref)
    ref2, _ = O.mean_log_prob(2 * recipes.gaussian(99, rows, 64), c2_layers(64, 3))
    assert abs(piped[0] - ref) <= 1e-9 * abs(ref) and abs(piped[1] - ref2) <= 1e-9 * abs(ref2)


def test_shard_bounds_cover_rows_exactly():
    from torch_mnf_amd.dist import shard_bounds

    for rows in (0, 1, 7, 1 << 20, (1 << 22) + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(rows, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == rows
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_spawns_its_ranks_as_child_processes():
    """`python bench.py --gpus 2` with no launcher: the parent must start the ranks itself (torch.distributed.run as
    a child process, before any GPU call) and pass the ranks' failure on as its own exit code.  On this CPU-only
    container every rank stops at "needs an MI355X" -- that message next to the launcher's per-rank failure report
    proves the ranks were started by torch.distributed.run (the launcher ends the other rank as soon as one has
    failed); the full run is the -m gpu test tests/test_hip_round2.py::test_bench_starts_its_own_ranks."""
    import subprocess
    import sys

    if torch.cuda.is_available():
        import pytest

        pytest.skip("CPU-container check; the GPU box runs the real thing")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert out.stderr.count("needs an MI355X") >= 1 and "local_rank" in out.stderr, out.stderr[-1500:]


def _actnorm_worker(rank, world, port, rows, dim, out):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch_mnf_amd as amd
    from torch_mnf_amd.dist import shard_bounds

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    x = 1.7 * recipes.gaussian(123, rows, dim) + 0.4
    lo, hi = shard_bounds(rows, world, rank)  # uneven shards (rows is odd)
    torch.manual_seed(7)  # the same random s, t on every rank (replicated parameters)
    f = amd.ActNormFlow(dim)
    f._maybe_init(x[lo:hi])  # the data-dependent initialisation proper: plain tensor ops, runs on the CPU shards
    out.put((rank, f.s.detach().clone(), f.t.detach().clone(), f.data_dep_init_done))
    dist.destroy_process_group()


def test_actnorm_data_dependent_init_is_rank_consistent():
    """ActNormFlow initialises s, t from its first batch (affine_constant_flow.py:42-50).  With that batch sharded over
    two ranks (uneven shards) every rank must get the GLOBAL batch's statistics: bit-equal across ranks, and equal to
    what one process computes from the concatenated batch."""
    import torch_mnf_amd as amd

    rows, dim, world = 1001, 12, 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_actnorm_worker, args=(r, world, port, rows, dim, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((out.get(timeout=120) for _ in range(world)), key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, s0, t0, d0), (_, s1, t1, d1) = got
    assert d0 and d1
    assert torch.equal(s0, s1) and torch.equal(t0, t1), "the replicas' parameters differ"
    x = 1.7 * recipes.gaussian(123, rows, dim) + 0.4
    torch.manual_seed(7)
    ref = amd.ActNormFlow(dim)
    ref._maybe_init(x)  # single process, whole batch: the reference's formulas as they stand
    assert float((s0 - ref.s.detach()).abs().max()) <= 1e-6 and float((t0 - ref.t.detach()).abs().max()) <= 1e-6
    # and NOT the local statistics (which is what an unsynchronised init would give)
    local = amd.ActNormFlow(dim)
    local._maybe_init(x[:501])
    assert float((s0 - local.s.detach()).abs().max()) > 1e-4
