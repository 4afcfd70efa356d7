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
        f = ShardedDOptimalObj(NumpyLocal(V[:, lo:hi].copy()), m, n, lo, hi, torch.device("cpu"), reduce=_DistSum())
        xt = torch.from_numpy(x)
        fx, g = f.func_grad(xt, 2)
        f0 = f(xt)
        g1 = f.gradient(xt)
        out[rank] = (fx, g.numpy().copy(), f0, g1.numpy().copy(), (lo, hi))
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
        fx, g, f0, g1, (lo, hi) = out[rank]
        assert abs(fx - fr) < 1e-11 * max(1, abs(fr)) and f0 == fx
        np.testing.assert_allclose(g, gr, rtol=1e-10)           # every rank holds the FULL gradient
        np.testing.assert_array_equal(g, g1)
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
