"""N > 1 path on CPU: world_size-2 gloo process group (LPT assignment + variable-length gather)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from points2surf_amd import sharding


def test_assign_lpt_properties():
    costs = [5, 1, 9, 3, 3, 7, 2]
    for world in (1, 2, 3, 8):
        parts = sharding.assign_lpt(costs, world)
        assert len(parts) == world
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(costs)))                       # a partition
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) <= sum(costs) / world + max(costs)          # LPT bound
        assert parts == sharding.assign_lpt(costs, world)             # deterministic
    assert sharding.assign_lpt([], 2) == [[], []]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    w, r, lr = sharding.init_process_group('gloo')
    assert (w, r) == (world, rank)
    # every rank owns different "shapes" with different query counts
    costs = [300, 120, 570, 499, 50]
    mine = sharding.assign_lpt(costs, world)[rank]
    res = []
    for s in mine:
        n = costs[s]
        res.append(torch.full((n,), float(s)) + torch.arange(n, dtype=torch.float32) * 1e-3)
    local = torch.cat(res) if res else torch.zeros(0)
    got = sharding.gather_variable(local, dst=0)
    sharding.barrier()
    if rank == 0:
        assert got is not None and len(got) == world
        total = sum(int(g.shape[0]) for g in got)
        assert total == sum(costs)
        # reassemble per shape in dataset order and check content
        parts = sharding.assign_lpt(costs, world)
        for rk, g in enumerate(got):
            off = 0
            for s in parts[rk]:
                seg = g[off:off + costs[s]]
                assert torch.allclose(seg, torch.full((costs[s],), float(s)) + torch.arange(costs[s]) * 1e-3)
                off += costs[s]
        np.save(os.path.join(outdir, 'ok.npy'), np.array([total]))
    else:
        assert got is None
    dist.destroy_process_group()


def test_gloo_world2_gather(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert int(np.load(os.path.join(str(tmp_path), 'ok.npy'))[0]) == 1539


def test_query_range_is_an_ordered_balanced_partition():
    from points2surf_amd import sharding
    for Q in (0, 1, 7, 8, 307237, 757499):
        for world in (1, 2, 3, 8):
            r = [sharding.query_range(Q, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == Q
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
