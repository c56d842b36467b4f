"""Shape-level sharding across the GPUs of one node (one process per GPU, torch.distributed).

The path has no cross-query state, so shapes shard embarrassingly: longest-processing-time-first
assignment by query count (``assign_shapes``: ONE policy for bench.py and the drop-in), no data-path collective.
Two things cross ranks:
  * the dataset-wide sub-sample RNG stream (reference source/data_loader.py:274-277), which keeps results
    bit-identical to the single-process run.  ``StreamHandoff`` (default when a process group exists): the owner of
    shape i consumes ITS shape's draws once more ahead of the inference (``skip_shape_stream`` on a snapshot) and
    hands the generator state (2.5 KB) to the owner of shape i+1 through the rendezvous store -- control plane, no
    collective; a rank never touches a shape it does not own.  Without a process group (ranks run one after the
    other, tests) or with ``P2S_STREAM_HANDOFF=replicate`` every rank consumes the draws of every foreign shape itself;
  * the final variable-length gather of per-shape results to rank 0 (``gather_variable``; RCCL over
    xGMI with the nccl backend, gloo in the CPU tests).
Replaces the reference's only multi-GPU mechanism, ``torch.nn.DataParallel`` (reference
source/points_to_surf_eval.py:168), which re-broadcasts all 21.5 MB of weights on every forward.
"""
import os

import numpy as np


def dist_env():
    """(world_size, rank, local_rank) from the torchrun environment."""
    return (int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')),
            int(os.environ.get('LOCAL_RANK', '0')))


def assign_lpt(costs, world):
    """Longest-processing-time-first: returns ``world`` lists of item indices (each ascending).
    Deterministic (ties broken by index) so every rank computes the same assignment."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += float(costs[i])
    return [sorted(x) for x in out]


def assign_shapes(query_counts, world):
    """THE shape -> rank policy of bench.py and the drop-in: longest-processing-time-first over the shapes' query counts
    (Cloud.count_queries: the cost of a shape is proportional to it).  Returns (lists of shape indices per rank, each
    ascending; owner[i] = rank of shape i)."""
    parts = assign_lpt(query_counts, world)
    owner = [0] * len(query_counts)
    for r, items in enumerate(parts):
        for i in items:
            owner[i] = r
    return parts, owner


def is_initialized():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def init_process_group(backend=None):
    """idempotent; 127.0.0.1 rendezvous (the container hostname may not resolve)"""
    import torch
    import torch.distributed as dist
    world, rank, local_rank = dist_env()
    # P2S_DIST_FORCE=1: create the process group also at world size 1 (one GPU: RCCL then really executes the
    # all_gather / gather / barrier of this module -- tests/test_gpu_rccl_world1.py)
    if (world == 1 and not os.environ.get('P2S_DIST_FORCE')) or dist.is_initialized():
        return world, rank, local_rank
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
        # P2S_DIST_BACKEND=gloo: rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks (RCCL
        # refuses two ranks on one device); the collectives then run on host memory (collective_device)
        backend = os.environ.get('P2S_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    kw = {}
    if backend == 'nccl':
        torch.cuda.set_device(local_rank)
        kw['device_id'] = torch.device('cuda', local_rank)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return world, rank, local_rank


def local_device_index(local_rank):
    """the device of a rank: its LOCAL_RANK, or (rehearsal with P2S_DIST_BACKEND=gloo on fewer GPUs) LOCAL_RANK modulo
    the number of visible devices"""
    import torch
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n > 0 and local_rank >= n and os.environ.get('P2S_DIST_BACKEND', 'nccl') != 'nccl':
        return local_rank % n
    return local_rank


def collective_device(device):
    """where tensors of a collective live: the GPU with RCCL (backend nccl), host memory with gloo (CPU tests, and the
    2-ranks-on-one-GPU rehearsal of bench.py)"""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() != 'nccl':
        return torch.device('cpu')
    return device


def gather_variable(t, dst=0):
    """Gather 1-D tensors of different lengths to ``dst``.  Returns the list (by rank) on ``dst``,
    None elsewhere.  One all_gather of the sizes + one padded gather (the path's only collective)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [t]
    world, rank = dist.get_world_size(), dist.get_rank()      # world size 1 with a group: the collectives still run
    t = t.to(collective_device(t.device))
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    padded = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    padded[:t.shape[0]] = t
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst)
    if rank != dst:
        return None
    return [b[:s] for b, s in zip(bufs, sizes)]


def query_range(n_queries, world, rank):
    """contiguous, ordered split of a shape's queries (intra-shape sharding for few, large shapes): [begin, end)"""
    base, rem = divmod(int(n_queries), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def skip_queries(cloud, rng_dev, cfg, queries, sub_sample_size, chunk=4096, rng_patch=None):
    """advance the RNG stream past ``queries`` ([m,3] device tensor, in order) without inference.  Fixed-radius models:
    ``rng_patch`` (the data set's first generator) is advanced past the patch choices of the same queries.
    ``chunk``: DEPRECATED and ignored (the library chooses its own batches); passing anything but the default warns."""
    if chunk != 4096:
        import warnings
        warnings.warn('sharding.skip_queries / skip_shape_stream: the chunk argument is ignored', DeprecationWarning, stacklevel=2)
    m = int(queries.shape[0])
    if m == 0:
        return
    if float(cfg.get('patch_radius', 0.0) or 0.0) > 0.0:
        from . import engine
        if rng_patch is None:
            raise ValueError('fixed-radius model: the generator of the patch choice is required')
        if cloud.n < sub_sample_size:
            # the inference path refuses such a cloud (p2s_infer_shape_ball: P2S_EINVAL) -- refuse here as well, BEFORE
            # either generator is advanced, so that all shard paths behave alike
            raise ValueError('fixed-radius model: a cloud with fewer points (%d) than the sub-sample (%d) is not supported'
                             % (cloud.n, sub_sample_size))
        engine.ball_skip(cloud, rng_patch, queries, float(cfg['patch_radius']), int(cfg.get('points_per_patch', 300)))
    # fixed_subsample: every query re-seeds the generator, nothing carries over -- but ``rng.seed(42)`` sits INSIDE the
    # N >= sub_sample_size branch (reference source/base/utils.py:210-211): a cloud with fewer points still shuffles
    # shape.pts from the dataset-wide stream, also in fixed mode (the NULL-ids skip takes the shuffle + pad path)
    if cfg.get('fixed_subsample') and cloud.n >= sub_sample_size:
        return
    if cloud.n < sub_sample_size:
        rng_dev.skip(cloud, sub_sample_size, n_queries=m)
        return
    if cfg.get('uniform_subsample'):
        rng_dev.skip(cloud, sub_sample_size, n_queries=m)
    else:
        rng_dev.skip(cloud, sub_sample_size, query_ms=queries)      # one call: the library walks the queries in batches of 4096


def skip_shape_stream(cloud, rng_dev, cfg, grid_resolution, epsilon, sub_sample_size, chunk=4096, rng_patch=None):
    """Advance the dataset-wide RNG stream past one shape without running the encoders and without
    materialising ids (NULL-ids path of the C ABI: p2s_max = a count of session values, p2s_vanilla = tables +
    offsets pass only)."""
    import torch
    q = cloud.query_grid(grid_resolution, epsilon)
    skip_queries(cloud, rng_dev, cfg, q, sub_sample_size, chunk=chunk, rng_patch=rng_patch)
    torch.cuda.synchronize()
    rng_dev.check()
    if rng_patch is not None:
        rng_patch.check()
    return int(q.shape[0])


class StreamHandoff:
    """The dataset-wide generator state(s) handed from the owner of shape i to the owner of shape i+1 through the
    process group's rendezvous store (TCPStore; keys are written once): the exact single-process stream without any
    rank consuming the draws of a foreign shape.

        h = StreamHandoff('rec/p2s_max', owner)              # owner[i] = rank of shape i, same list on every rank
        for i in my shapes, ascending:
            with h.guard(i):                                 # any exception -> the 'failed' key: peers raise, not hang
                h.begin(i, rngs)                             # my generators -> start of shape i (waits for the token)
                if h.must_publish(i):                        # a later shape belongs to another rank, not yet served
                    h.publish_after(i, rngs, advance)        # advance(k): consume shape k's draws (skip_shape_stream)
                ... infer shape i with rngs ...
                h.done(i)
        h.finish()                                           # all ranks through, or the failure of one of them raised here

    ``rngs``: list of engine.Rng (the sub-sample generator; fixed-radius models add the patch-choice generator).
    Consecutive shapes of one owner: the token for the next FOREIGN shape j is published as soon as the first of them
    is reached (``advance`` runs for i .. j-1), not after their inferences -- the ring never waits for an inference.
    Failure: the owner of a shape that raises writes ``p2s/stream/<tag>/failed`` (rank, shape, message); ``begin``
    polls its key and that one (slices of at most ``poll_s`` seconds) and raises, naming the failed shape and rank."""

    def __init__(self, tag, owner, rank=None, store=None, timeout_s=7200.0, poll_s=0.05):
        import torch.distributed as dist
        self.owner = list(owner)
        self.rank = dist.get_rank() if rank is None else int(rank)
        if store is None:
            from torch.distributed.distributed_c10d import _get_default_store
            store = _get_default_store()
        self.store = store
        self.tag = str(tag)
        self.timeout_s = float(timeout_s)
        self.poll_s = min(float(poll_s), 5.0)
        self.at = 0                     # my generators are positioned at the start of this shape
        self.published = 0              # the start of every shape <= this index is known to its owner (as far as I know)
        self.waited_s = 0.0

    def _key(self, i):
        return 'p2s/stream/%s/%d' % (self.tag, i)

    def _failed_key(self):
        return 'p2s/stream/%s/failed' % self.tag

    @staticmethod
    def pack(rngs):
        out = []
        for r in rngs:
            mt, pos = r.get_state()
            out.append(np.concatenate([np.ascontiguousarray(mt, dtype=np.uint32), np.array([pos], dtype=np.uint32)]))
        return np.concatenate(out).tobytes()

    @staticmethod
    def unpack(blob, rngs):
        a = np.frombuffer(blob, dtype=np.uint32)
        if a.size != 625 * len(rngs):
            raise RuntimeError('stream hand-off: %d words for %d generators' % (a.size, len(rngs)))
        for k, r in enumerate(rngs):
            r.set_state(a[625 * k:625 * k + 624].copy(), int(a[625 * k + 624]))

    def failed(self):
        """the failure record a rank left (str), or None"""
        if self.store.check([self._failed_key()]):
            return self.store.get(self._failed_key()).decode('utf-8', 'replace')
        return None

    def fail(self, i, exc):
        """tell the ring that shape ``i`` of this rank raised: every rank waiting in ``begin`` raises instead of hanging"""
        try:
            if not self.store.check([self._failed_key()]):          # the first failure stays on record
                where = 'at shape %d' % i if i >= 0 else 'before its first shape (loading / counting the data set)'
                self.store.set(self._failed_key(), ('rank %d failed %s: %s: %s'
                                                    % (self.rank, where, type(exc).__name__, exc))[:2000])
        except Exception:
            pass                         # the store itself is gone: the process group's own timeout takes over

    def guard(self, i):
        """context manager around everything a rank does for its shape ``i``"""
        import contextlib

        @contextlib.contextmanager
        def _g():
            try:
                yield self
            except BaseException as e:
                self.fail(i, e)
                raise
        return _g()

    def begin(self, i, rngs):
        """position ``rngs`` at the first draw of shape ``i`` (which this rank owns)"""
        import time
        if self.owner[i] != self.rank:
            raise ValueError('shape %d belongs to rank %d' % (i, self.owner[i]))
        if i == self.at or i == 0:
            self.at = i
            return
        t0 = time.time()
        key, nap = self._key(i), 0.001
        while not self.store.check([key]):
            why = self.failed()
            if why is not None:
                raise RuntimeError('stream hand-off (%s): waiting for the start of shape %d on rank %d, but %s'
                                   % (self.tag, i, self.rank, why))
            if time.time() - t0 > self.timeout_s:
                raise TimeoutError('stream hand-off (%s): the start of shape %d (owner of shape %d: rank %d) did not arrive '
                                   'within %.0f s' % (self.tag, i, i - 1, self.owner[i - 1], self.timeout_s))
            time.sleep(nap)
            nap = min(nap * 1.5, self.poll_s)
        self.unpack(self.store.get(key), rngs)
        self.waited_s += time.time() - t0
        self.at = i
        self.published = max(self.published, i)

    def next_foreign(self, i):
        """the first shape after ``i`` that belongs to another rank (None: the rest of the dataset is mine)"""
        for j in range(i + 1, len(self.owner)):
            if self.owner[j] != self.rank:
                return j
        return None

    def must_publish(self, i):
        j = self.next_foreign(i)
        return j is not None and j > self.published

    def publish_after(self, i, rngs, advance):
        """``advance(k)`` consumes the draws of my shape k: run it for k = i .. j-1 (j = the next foreign shape), publish the
        state as the start of shape j, then put the generators back to the start of shape i"""
        j = self.next_foreign(i)
        if j is None or j <= self.published:
            return
        snap = [r.get_state() for r in rngs]
        for k in range(i, j):
            advance(k)
        self.store.set(self._key(j), self.pack(rngs))
        self.published = j
        for r, (mt, pos) in zip(rngs, snap):
            r.set_state(mt, pos)

    def done(self, i):
        self.at = i + 1

    def finish(self):
        """after a rank's last shape, before the closing collective: returns when every rank has got here, raises when one
        left a failure record instead -- a plain barrier would sit out the process group's time-out on a rank that raised
        after the others were through with their shapes"""
        import time
        world = max(self.owner) + 1 if self.owner else 1
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                world = dist.get_world_size()
        except Exception:
            pass
        self.store.set('p2s/stream/%s/finished/%d' % (self.tag, self.rank), b'1')
        keys = ['p2s/stream/%s/finished/%d' % (self.tag, r) for r in range(world)]
        t0, nap = time.time(), 0.001
        while not self.store.check(keys):
            why = self.failed()
            if why is not None:
                raise RuntimeError('stream hand-off (%s): rank %d is through with its shapes, but %s' % (self.tag, self.rank, why))
            if time.time() - t0 > self.timeout_s:
                raise TimeoutError('stream hand-off (%s): not every rank finished within %.0f s' % (self.tag, self.timeout_s))
            time.sleep(nap)
            nap = min(nap * 1.5, self.poll_s)
        why = self.failed()
        if why is not None:
            raise RuntimeError('stream hand-off (%s): %s' % (self.tag, why))


def stream_handoff_enabled():
    """token hand-off needs a process group (its store); P2S_STREAM_HANDOFF=replicate forces the replicate mode"""
    return is_initialized() and os.environ.get('P2S_STREAM_HANDOFF', 'token') != 'replicate'
