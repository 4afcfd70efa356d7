"""CPU tests of the multi-GPU path: world_size-2 gloo run of the design-point
sharded objective with a NumPy stand-in for the per-rank HIP compute, checked against the oracle
on the whole matrix; plus the dealing of instances / columns to ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from accbpg_and_fw_amd.sharded import ShardedDOptimalObj, shard_bounds, split_instances, _DistSum


class NumpyLocal:
    """Stand-in for DOptimalObj over a column block: same staged interface, NumPy arithmetic."""

    def __init__(self, V_local):
        self.V = V_local
        self.L = None

    def gram_into(self, x_local, gram):
        G = (self.V * x_local.numpy()) @ self.V.T
        gram.copy_(torch.from_numpy(np.tril(G)))               # lower triangle significant

    def factor(self, gram):
        G = np.tril(gram.numpy())
        G = G + np.tril(G, -1).T
        self.L = np.linalg.cholesky(G)
        return -2.0 * np.sum(np.log(np.diag(self.L)))

    def grad_from_factor(self, g_local):
        Y = np.linalg.solve(self.L, self.V)
        g_local.copy_(torch.from_numpy(-np.sum(Y * Y, axis=0)))


class NumpyMessages:
    """Stand-in for the HIP message kernels (accbpg_tri_pack / _unpack / accbpg_vec_count_bad)."""

    @staticmethod
    def pack(gram, packed):
        m = gram.shape[0]
        r, c = np.tril_indices(m)
        packed[:r.size] = gram[torch.from_numpy(r), torch.from_numpy(c)]

    @staticmethod
    def unpack(packed, gram):
        m = gram.shape[0]
        r, c = np.tril_indices(m)
        gram[torch.from_numpy(r), torch.from_numpy(c)] = packed[:r.size]

    @staticmethod
    def count_bad(x_local, slot):
        slot[0] = float((~(x_local >= 0)).sum())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, m, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        np.random.seed(42)
        V = np.random.randn(m, n)
        rng = np.random.RandomState(7)
        x = rng.rand(n) + 0.01
        x /= x.sum()
        lo, hi = shard_bounds(n, world, rank)
        f = ShardedDOptimalObj(NumpyLocal(V[:, lo:hi].copy()), m, n, lo, hi, torch.device("cpu"), reduce=_DistSum(),
                               messages=NumpyMessages, world=world)
        xt = torch.from_numpy(x)
        fx, g = f.func_grad(xt, 2)
        f0 = f(xt)
        g1 = f.gradient(xt)
        # a negative entry that lives on ONE rank only: both ranks must raise the reference's assertion
        # (accbpg/functions.py:45), whatever the summed Gram matrix looks like
        xbad = xt.clone()
        xbad[n - 1] = -1e-3
        try:
            f(xbad)
            raised = False
        except AssertionError:
            raised = True
        # the summing fallback of the gradient assembly (reducers without a gather)
        f2 = ShardedDOptimalObj(NumpyLocal(V[:, lo:hi].copy()), m, n, lo, hi, torch.device("cpu"), reduce=_DistSum(),
                                messages=NumpyMessages, world=None)
        g2 = f2.gradient(xt)
        out[rank] = (fx, g.numpy().copy(), f0, g1.numpy().copy(), (lo, hi), raised, g2.numpy().copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape", [(12, 101), (20, 64)])
def test_sharded_objective_world2_gloo(shape):
    from oracle import np_oracle as O
    m, n = shape
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, m, n, out), nprocs=world, join=True)
    np.random.seed(42)
    V = np.random.randn(m, n)
    rng = np.random.RandomState(7)
    x = rng.rand(n) + 0.01
    x /= x.sum()
    fr, gr = O.DOptOracle(V).func_grad(x, 2)
    covered = []
    for rank in range(world):
        fx, g, f0, g1, (lo, hi), raised, g2 = out[rank]
        assert abs(fx - fr) < 1e-11 * max(1, abs(fr)) and f0 == fx
        np.testing.assert_allclose(g, gr, rtol=1e-10)           # every rank holds the FULL gradient
        np.testing.assert_array_equal(g, g1)
        np.testing.assert_array_equal(g, g2)                    # all-gather and padded sum agree exactly
        assert raised
        covered += list(range(lo, hi))
    assert covered == list(range(n))


def test_shard_bounds_and_instance_split():
    for n, world in [(10, 3), (262144, 8), (7, 8), (64, 1)]:
        spans = [shard_bounds(n, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    got = sorted(sum((split_instances(64, 8, r) for r in range(8)), []))
    assert got == list(range(64))
    assert all(len(split_instances(64, 8, r)) == 8 for r in range(8))


def _deal_worker(rank, world, port, out):
    from accbpg_and_fw_amd.batched import solve_instances
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        built = []

        def make(i):
            built.append(i)
            return (float(i), None, 1.0, np.full(3, float(i)))

        def solver(f, h, L, x0, scale=1.0):
            return x0 * scale, np.array([f, f + 1.0]), torch.tensor([f])

        res = solve_instances(make, 7, solver, world=world, rank=rank, concurrent=False, scale=3.0)
        out[rank] = (built, [(r[0].tolist(), r[1].tolist(), r[2].tolist()) for r in res])
    finally:
        dist.destroy_process_group()


def test_instances_dealt_solved_gathered_world2_gloo():
    """BASELINE config 4 plumbing: every rank builds only its own instances, and every rank ends up with all
    results in instance order."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_deal_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0][0] == [0, 2, 4, 6] and out[1][0] == [1, 3, 5]
    want = [([3.0 * i] * 3, [float(i), i + 1.0], [float(i)]) for i in range(7)]
    assert out[0][1] == want and out[1][1] == want


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` outside a torch.distributed environment starts two ranks itself (before any
    GPU call) and relays rank 0's JSON line; a rank count that disagrees with WORLD_SIZE is refused."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    line = [ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["instances"] == 16 and rec["backend"] == "gloo"
    env["WORLD_SIZE"] = "4"
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                         capture_output=True, text=True, env=env, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)
