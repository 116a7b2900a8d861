"""CPU, world_size 2, gloo: the row sharding + all-gather re-assembly of feos_torch_amd.dist
(the N>1 path of bench.py) with the CPU oracle injected as the per-shard compute."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from feos_torch_amd import dist as pdist
    from feos_torch_amd.synthetic import pure_batch
    from oracle import pyoracle as orc

    dist.init_process_group("gloo", rank=rank, world_size=world)

    def compute(par, T):
        p, st = orc.pure_vapor_pressure(par.numpy(), T.numpy(), prec=0)
        return torch.from_numpy(p), torch.from_numpy(st)

    P, T = pure_batch(n, seed=3)
    p, st = pdist.sharded_vapor_pressure(torch.from_numpy(P), torch.from_numpy(T), compute=compute)
    q.put((rank, p.numpy(), st.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1000, 1001, 1])
def test_sharded_vapor_pressure_matches_unsharded(oracle, n):
    from feos_torch_amd.synthetic import pure_batch

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P, T = pure_batch(n, seed=3)
    want, st = oracle.pure_vapor_pressure(P, T, prec=0)
    for _, got, gst in results:
        assert got.shape == (n,) and np.array_equal(gst, st)
        assert np.array_equal(got, want)  # same arithmetic per row -> bit identical


def _worker_mix(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from feos_torch_amd import dist as pdist
    from feos_torch_amd.synthetic import mix_batch
    from oracle import pyoracle as orc

    dist.init_process_group("gloo", rank=rank, world_size=world)

    def compute(par, kij, T, z, p0):
        p, rho4, st = orc.mix_bubble_dew(par.numpy(), kij.numpy(), T.numpy(), z.numpy(), p0.numpy(), False)
        return torch.from_numpy(p), torch.from_numpy(rho4), torch.from_numpy(st)

    b = [torch.from_numpy(np.ascontiguousarray(v)) for v in mix_batch(n, seed=4)]
    p, rho4, st = pdist.sharded_rows(compute, *b)
    q.put((rank, p.numpy(), rho4.numpy(), st.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_rows_generic_bubble_points(oracle):
    """sharded_rows with a multi-output, multi-input row-wise solve ([n], [n,4] and bool outputs), uneven shards."""
    from feos_torch_amd.synthetic import mix_batch

    n, world = 301, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mix, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_p, want_rho, want_st = oracle.mix_bubble_dew(*mix_batch(n, seed=4), False)
    for _, p_, rho_, st_ in results:
        assert st_.dtype == np.bool_ and np.array_equal(st_, want_st)
        assert np.array_equal(p_, want_p) and np.array_equal(rho_, want_rho)


def test_shard_bounds_cover_every_row_once():
    from feos_torch_amd.dist import shard_bounds

    for n in (0, 1, 7, 8, 10_000_001):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
