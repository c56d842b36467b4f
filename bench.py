#!/usr/bin/env python
"""Headline benchmark: SDF queries/s per GPU, p2s_max, 256^3 query grid (BASELINE.json metric).

A *step* = one complete shape: near-surface query grid of a 256^3 volume -> for every query the
300-NN patch (fp64-exact), the 1000-point global sub-sample (numpy-legacy MT19937 stream), the
PointNet encoders + decoder -> SDF.  The cloud is resident in HBM when the timed region starts
(config 3 of BASELINE.json with a synthetic stand-in cloud: the Famous set cannot be downloaded).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; shapes shard across ranks (every rank processes its own shape per
step -> weak scaling, no data-path collective); the per-shape SDF arrays are gathered to rank 0
over RCCL at the end of the timed region (the path's only exchange step).  Rank 0 prints ONE JSON
line.  ``roofline`` is for the dominant kernel (p2s_chain_kernel, MFMA-bound) from HIP events
recorded on the launch stream during the timed steps; ``cpu_baseline`` times the torch-CPU port of
the reference's path (oracle/torch_port.py) on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_QUERY = 776773632          # p2s_max, SURVEY.md 8(d) (torch FlopCounter == hand count)
# FLOP of the per-point layers + max-pool inputs (everything the chain kernel replaces): total minus the
# per-query FC layers (2 STN heads 1,703,936 MAC each, decoder 1,343,744 MAC), SURVEY.md 8(a) table
FLOP_CHAIN_PER_QUERY = FLOP_PER_QUERY - 2 * (2 * 1703936 + 1343744)
BYTES_PER_QUERY = 15620             # minimal HBM traffic per query, SURVEY.md 8(d)
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md
GRID_RES, EPSILON = 256, 3
SEED_DATA = 40938661


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--points', type=int, default=50000, help='points of the synthetic cloud')
    ap.add_argument('--cpu-seconds', type=float, default=25.0, help='target CPU-baseline duration (0 = skip)')
    ap.add_argument('--chunk', type=int, default=0)
    ap.add_argument('--bf16', action='store_true',
                    help='secondary mode (BASELINE configs[3]): bf16 encoder + fp32 decoder; NOT the headline metric')
    ap.add_argument('--res', type=int, default=GRID_RES,
                    help='query-grid resolution; the headline metric is quoted at 256 (other values: BASELINE configs 1/4)')
    return ap.parse_args()


def cpu_baseline(w, cfg, cloud, queries, target_seconds, grid_res=GRID_RES):
    import torch
    from oracle.torch_port import TorchPort
    port = TorchPort(w, cfg)
    threads = torch.get_num_threads()
    rng = np.random.RandomState(SEED_DATA)
    # probe, then size the sample for ~target_seconds of CPU work
    n0 = 64
    t0 = time.time()
    port.infer_queries(cloud, queries[:n0], rng, batch=n0)
    dt0 = time.time() - t0
    n = int(min(max(target_seconds / max(dt0 / n0, 1e-6), n0), 4096, queries.shape[0]))
    rng = np.random.RandomState(SEED_DATA)
    t0 = time.time()
    port.infer_queries(cloud, queries[:n], rng, batch=500)
    dt = time.time() - t0
    return {'value': n / dt, 'unit': 'queries/s', 'cores': int(threads), 'kind': 'port',
            'sample': 'first %d of the same %d^3-grid queries (kNN cKDTree + RandomState sub-sample + torch-CPU '
                      'forward, batch 500), %.1f s' % (n, grid_res, dt),
            'host_cpus': os.cpu_count()}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from points2surf_amd import engine, synth

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the HIP engine has no CPU fallback')
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    w, cfg = synth.make_weights('p2s_max')
    if args.bf16:
        cfg = dict(cfg, encoder_bf16=True)
    model = engine.Model(w, cfg)
    model.set_profiling(True)
    # every rank owns its own shapes (seeded by rank): shape-level sharding
    pts = synth.make_cloud(args.points, seed=1000 + rank)
    cloud = engine.Cloud(pts)
    rng = engine.Rng(SEED_DATA)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sdf = None
    for _ in range(args.warmup):
        sdf, _ = engine.infer_shape(model, cloud, rng, args.res, EPSILON, chunk=args.chunk, want_queries=False)
    barrier()
    t0 = time.time()
    n_queries = 0
    acc = {}
    for _ in range(args.steps):
        sdf, _ = engine.infer_shape(model, cloud, rng, args.res, EPSILON, chunk=args.chunk, want_queries=False)
        n_queries += int(sdf.shape[0])
        for k, v in model.counters().items():
            acc[k] = acc.get(k, 0) + v
        if world > 1:
            # final gather of the variable-length per-shape SDF to rank 0 (RCCL over xGMI)
            sizes = [torch.zeros(1, dtype=torch.int64, device='cuda') for _ in range(world)]
            dist.all_gather(sizes, torch.tensor([sdf.shape[0]], dtype=torch.int64, device='cuda'))
            cap = int(max(int(s.item()) for s in sizes))
            padded = torch.zeros(cap, dtype=torch.float32, device='cuda')
            padded[:sdf.shape[0]] = sdf
            bufs = [torch.empty(cap, dtype=torch.float32, device='cuda') for _ in range(world)] if rank == 0 else None
            dist.gather(padded, bufs, dst=0)
    barrier()
    dt = time.time() - t0

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        nq = torch.tensor([n_queries], dtype=torch.int64, device='cuda')
        dist.all_reduce(nq, op=dist.ReduceOp.SUM)
        total_queries = int(nq.item())
    else:
        total_queries = n_queries

    if rank == 0:
        value = total_queries / dt
        launches = max(acc.get('launches_chain', 0), 1)
        chain_ms = acc.get('ms_chain_stn', 0.0) + acc.get('ms_chain_main', 0.0)
        avg_launch_ms = chain_ms / launches
        flop_per_launch = FLOP_CHAIN_PER_QUERY * n_queries / launches
        achieved = flop_per_launch / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
        peak = 2500.0 if args.bf16 else PEAK_FP32_MFMA_TFLOPS     # dense MFMA peak of the compute dtype
        # HBM bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE),
        # committed under profiles/; they cannot be collected from inside this process
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(REPO, 'profiles', 'r01', 'pmc_summary.json')) as f:
                ck = json.load(f)['chain_kernel']
            traffic = ck['hbm_traffic_bytes_per_launch'] * (2.0 * n_queries / launches) / ck['queries_per_launch']
            if args.bf16:
                traffic, traffic_src = None, None        # counters were collected for the fp32 kernel only
            traffic_src = 'profiles/r01/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, scaled to this launch size)'
        except Exception:
            pass
        out = {
            'metric': 'SDF queries/sec/GPU (p2s_max, %d^3 grid%s)' % (args.res, ', bf16 encoder' if args.bf16 else ''),
            'value': value, 'unit': 'queries/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / max(args.steps, 1) * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16' if args.bf16 else 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE.json configs[2]: p2s_max, grid_res=%d, eps=3, kNN patch=300 / global ' % args.res +
                                   'sub=1000, fp32; synthetic %d-point cloud per rank (Famous set not available '
                                   'offline), seeded random-init weights' % args.points,
                       'queries_per_shape_rank0': int(sdf.shape[0]), 'parallelism': 'shape-sharded x%d' % world,
                       'shapes_per_hour': world * args.steps / dt * 3600.0,
                       'queries_per_s_per_gpu': value / world},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': 'p2s_chain_bf16_kernel' if args.bf16 else 'p2s_chain_kernel', 'launches': int(launches), 'avg_launch_ms': avg_launch_ms,
                         'algorithmic_flop_per_launch': flop_per_launch,
                         'hbm_algorithmic_GBps': value / world * BYTES_PER_QUERY / 1e9},
            'stage_ms_rank0': {k: acc[k] for k in sorted(acc) if k.startswith('ms_')},
        }
        if args.cpu_seconds > 0 and world == 1:
            q = cloud.query_grid(args.res, EPSILON).cpu().numpy()
            out['cpu_baseline'] = cpu_baseline(w, cfg, pts, q, args.cpu_seconds, args.res)
        elif world == 1:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
