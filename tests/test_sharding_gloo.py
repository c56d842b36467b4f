"""N > 1 path on CPU: world_size-2 gloo process group (LPT assignment + variable-length gather)."""
import os
import time
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


def test_assign_shapes_is_lpt_with_owner_list():
    costs = [307237, 571597, 499408, 307237, 571597, 499408, 10]
    for world in (1, 2, 4, 8):
        parts, owner = sharding.assign_shapes(costs, world)
        assert parts == sharding.assign_lpt(costs, world)
        assert all(owner[i] == r for r, p in enumerate(parts) for i in p) and len(owner) == len(costs)


class _FakeRng:
    """stands in for engine.Rng (get_state / set_state): the 'stream' is a counter in mt[0], 'position' in pos"""

    def __init__(self, seed):
        self.mt = np.zeros(624, np.uint32)
        self.mt[0] = seed
        self.pos = 624

    def get_state(self):
        return self.mt.copy(), self.pos

    def set_state(self, mt, pos):
        self.mt, self.pos = np.array(mt, dtype=np.uint32), int(pos)

    def consume(self, n):                      # a shape that draws n words
        self.mt[0] = np.uint32((int(self.mt[0]) * 1664525 + n) & 0xffffffff)
        self.pos = (self.pos + n) % 625


DRAWS = [11, 7, 5, 13, 3, 17, 19, 2, 23]          # words each shape consumes
OWNER = [0, 1, 1, 2, 0, 0, 2, 1, 0]


def _handoff_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    sharding.init_process_group('gloo')
    assert sharding.stream_handoff_enabled()
    a, b = _FakeRng(40938661), _FakeRng(99)       # two generators (fixed-radius models hand over both)
    h = sharding.StreamHandoff('test', OWNER, rank=rank)
    seen, published_at = {}, {}
    for i in [k for k, o in enumerate(OWNER) if o == rank]:
        with h.guard(i):
            h.begin(i, [a, b])
            seen[i] = (int(a.mt[0]), a.pos, int(b.mt[0]), b.pos)
            if h.must_publish(i):
                def advance(k):
                    a.consume(DRAWS[k])
                    b.consume(2 * DRAWS[k])
                h.publish_after(i, [a, b], advance)
                published_at[h.published] = i
                assert (int(a.mt[0]), a.pos, int(b.mt[0]), b.pos) == seen[i]     # put back to the start of shape i
            a.consume(DRAWS[i])                       # "inference" of shape i
            b.consume(2 * DRAWS[i])
            h.done(i)
    np.save(os.path.join(outdir, 'seen_%d.npy' % rank), np.array([[k] + list(v) for k, v in seen.items()], dtype=np.int64))
    np.save(os.path.join(outdir, 'pub_%d.npy' % rank), np.array(sorted(published_at.items()), dtype=np.int64).reshape(-1, 2))
    h.finish()                                        # every rank is through: returns
    sharding.barrier()
    dist.destroy_process_group()


def test_stream_handoff_gives_every_owner_the_single_process_state(tmp_path):
    """sharding.StreamHandoff over the rendezvous store, 3 gloo ranks: every shape's owner starts from exactly the state
    a single process walking all shapes in order would have (also across consecutive shapes of one owner, and with the
    owner of shape i+1 still busy when shape i publishes)"""
    port = _free_port()
    mp.spawn(_handoff_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    a, b = _FakeRng(40938661), _FakeRng(99)
    want = {}
    for i, n in enumerate(DRAWS):
        want[i] = (int(a.mt[0]), a.pos, int(b.mt[0]), b.pos)
        a.consume(n)
        b.consume(2 * n)
    got = {}
    for r in range(3):
        for row in np.load(os.path.join(str(tmp_path), 'seen_%d.npy' % r)):
            got[int(row[0])] = tuple(int(x) for x in row[1:])
    assert got == want
    # consecutive shapes of one owner: the start of the next FOREIGN shape is published when the FIRST of them is
    # reached, not after their inferences (OWNER = [0, 1, 1, 2, 0, 0, 2, 1, 0]: rank 1 publishes shape 3 at shape 1,
    # rank 0 publishes shape 6 at shape 4)
    pub = {r: {int(j): int(i) for j, i in np.load(os.path.join(str(tmp_path), 'pub_%d.npy' % r))} for r in range(3)}
    assert pub[1][3] == 1 and pub[0][6] == 4 and pub[0][1] == 0 and pub[2][4] == 3 and pub[2][7] == 6 and pub[1][8] == 7


def _handoff_failing_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    sharding.init_process_group('gloo')
    a = _FakeRng(40938661)
    h = sharding.StreamHandoff('failing', [0, 1, 2, 0, 1, 2], rank=rank, timeout_s=120.0)
    t0 = time.time()
    what = 'finished'
    try:
        for i in [k for k, o in enumerate(h.owner) if o == rank]:
            with h.guard(i):
                h.begin(i, [a])
                if rank == 1 and i == 1:
                    raise ValueError('injected failure of shape 1')          # BEFORE publishing the start of shape 2
                if h.must_publish(i):
                    h.publish_after(i, [a], lambda k: a.consume(DRAWS[k]))
                a.consume(DRAWS[i])
                h.done(i)
    except ValueError as e:
        what = 'raised ValueError: %s' % e
    except RuntimeError as e:
        what = 'raised RuntimeError: %s' % e
    with open(os.path.join(outdir, 'out_%d.txt' % rank), 'w') as f:
        f.write('%.2f\n%s\n' % (time.time() - t0, what))
    # no barrier: a crashed rank would not reach it either
    dist.destroy_process_group()


def test_stream_handoff_failure_reaches_the_waiting_ranks(tmp_path):
    """a rank that raises while it holds the token leaves a 'failed' record in the store; the ranks
    waiting for a later token raise within seconds, naming the failed shape and rank, instead of blocking for the 7200 s
    time-out"""
    port = _free_port()
    mp.spawn(_handoff_failing_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    out = {}
    for r in range(3):
        with open(os.path.join(str(tmp_path), 'out_%d.txt' % r)) as f:
            secs, what = f.read().split('\n')[:2]
        out[r] = (float(secs), what)
    assert out[1][1].startswith('raised ValueError: injected failure')
    for r in (0, 2):       # rank 2 waits for shape 2, rank 0 for shape 3: both learn of rank 1's failure
        assert out[r][1].startswith('raised RuntimeError: stream hand-off (failing)'), out[r]
        assert 'rank 1 failed at shape 1' in out[r][1] and 'injected failure' in out[r][1]
        assert out[r][0] < 30.0, out[r]


def _handoff_late_failure_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    sharding.init_process_group('gloo')
    a = _FakeRng(7)
    h = sharding.StreamHandoff('late', [0, 1], rank=rank, timeout_s=120.0)
    t0 = time.time()
    what = 'finished'
    try:
        for i in [k for k, o in enumerate(h.owner) if o == rank]:
            with h.guard(i):
                h.begin(i, [a])
                if h.must_publish(i):
                    h.publish_after(i, [a], lambda k: a.consume(DRAWS[k]))
                if rank == 1:
                    time.sleep(1.0)                      # rank 0 is through with its only shape by now
                    raise ValueError('injected late failure')
                a.consume(DRAWS[i])
                h.done(i)
        h.finish()
    except ValueError as e:
        what = 'raised ValueError: %s' % e
    except RuntimeError as e:
        what = 'raised RuntimeError: %s' % e
    with open(os.path.join(outdir, 'late_%d.txt' % rank), 'w') as f:
        f.write('%.2f\n%s\n' % (time.time() - t0, what))
    dist.destroy_process_group()


def test_stream_handoff_finish_raises_on_a_rank_that_failed_after_the_others_were_done(tmp_path):
    """a rank that is through with its shapes does not walk into the closing barrier while another rank has failed (it would
    sit out the process group's time-out there): StreamHandoff.finish polls the failure record"""
    port = _free_port()
    mp.spawn(_handoff_late_failure_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out = {}
    for r in range(2):
        with open(os.path.join(str(tmp_path), 'late_%d.txt' % r)) as f:
            secs, what = f.read().split('\n')[:2]
        out[r] = (float(secs), what)
    assert out[1][1].startswith('raised ValueError: injected late failure')
    assert out[0][1].startswith('raised RuntimeError: stream hand-off (late): rank 0 is through with its shapes, but rank 1 failed at shape 1'), out[0]
    assert out[0][0] < 30.0


def _handoff_early_failure_worker(rank, world, port, outdir):
    """the drop-in's order of events (points_to_surf_eval): the hand-off object exists BEFORE the per-rank loop that loads
    and counts every shape; rank 1 fails inside that loop (a corrupt cloud file), rank 0 gets as far as waiting for a token"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    sharding.init_process_group('gloo')
    a = _FakeRng(7)
    h = sharding.StreamHandoff('early', [], rank=rank, timeout_s=120.0)
    t0 = time.time()
    what = 'finished'
    try:
        with h.guard(-1):
            if rank == 1:
                raise OSError('cloud file truncated')
        h.owner = [1, 0]                                   # rank 0 owns shape 1: waits for rank 1's token
        for i in [k for k, o in enumerate(h.owner) if o == rank]:
            with h.guard(i):
                h.begin(i, [a])
                a.consume(DRAWS[i])
                h.done(i)
        h.finish()
    except OSError as e:
        what = 'raised OSError: %s' % e
    except RuntimeError as e:
        what = 'raised RuntimeError: %s' % e
    with open(os.path.join(outdir, 'early_%d.txt' % rank), 'w') as f:
        f.write('%.2f\n%s\n' % (time.time() - t0, what))
    dist.destroy_process_group()


def test_stream_handoff_failure_before_the_first_shape(tmp_path):
    """a rank that fails while it loads / counts the data set (before any shape is its turn) leaves a record too;
    the rank waiting for its token raises within seconds and says where the other one failed"""
    port = _free_port()
    mp.spawn(_handoff_early_failure_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out = {}
    for r in range(2):
        with open(os.path.join(str(tmp_path), 'early_%d.txt' % r)) as f:
            secs, what = f.read().split('\n')[:2]
        out[r] = (float(secs), what)
    assert out[1][1].startswith('raised OSError: cloud file truncated')
    assert 'rank 1 failed before its first shape' in out[0][1] and 'cloud file truncated' in out[0][1], out[0]
    assert out[0][0] < 30.0


def _range_split_worker(rank, world, port, outdir):
    """three ranks skip ONE shape -- each walks every candidate start of a window around the predicted
    start of its query range (tests/wchoice_model.py: range_map), publishes the map through the rendezvous store, rank 0
    composes the maps"""
    import pickle
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    sharding.init_process_group('gloo')
    from torch.distributed.distributed_c10d import _get_default_store
    from tests import wchoice_model as wm
    from tests import test_wchoice_algorithm as T
    nsel, W = 40, 64
    pts, qs, tbs, st, xs = T._split_fixture(nsel=nsel)          # every rank has the cloud, the queries and the stream
    nq = len(tbs)
    a, b = sharding.query_range(nq, world, rank)
    pred = T._predicted_start(tbs, a, nsel)
    starts = range(max(0, pred - W // 2), pred + W // 2) if rank else [0]
    mp_, evals, _ = wm.range_map(tbs[a:b], xs, starts, nsel)
    store = _get_default_store()
    store.set('p2s/split/%d' % rank, pickle.dumps(mp_))
    if rank == 0:
        s = 0
        for r in range(world):
            s = pickle.loads(store.get('p2s/split/%d' % r))[s]           # KeyError = the window missed the true start
        path = T._single_stream(pts, qs, tbs, st, xs, nsel)              # numpy-checked single stream
        with open(os.path.join(outdir, 'split.txt'), 'w') as f:
            f.write('%d %d\n' % (s, path[-1]))
    dist.barrier()
    dist.destroy_process_group()


def test_range_split_skip_world3_equals_the_single_stream(tmp_path):
    port = _free_port()
    mp.spawn(_range_split_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    got, want = open(os.path.join(str(tmp_path), 'split.txt')).read().split()
    assert got == want and int(got) > 3 * 30 * 40
