#!/usr/bin/env python
"""Headline benchmark: SDF queries/s per GPU, p2s_max, 256^3 query grid (BASELINE.json metric).

A *step* = one complete shape per rank: near-surface query grid of a 256^3 volume -> for every query the
300-NN patch (fp64-exact), the 1000-point global sub-sample (numpy-legacy MT19937 stream), the
PointNet encoders + decoder -> SDF.  The workload is the committed ``abc_minimal`` test shape
(tests/golden/abc_minimal/04_pts/00994122..., 34,693 points, Q = 307,237 queries at 256^3, eps 3) -- the input the
parity tests pin against the unmodified reference -- with seeded random-init weights (no pretrained weights offline).
The cloud is resident in HBM when the timed region starts.

  python bench.py --gpus N --steps K --warmup W [--rng-mode dataset|per_shape]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU (``--gpus N`` without a torchrun environment re-executes itself under
``torch.distributed.run`` with N ranks; fewer than N visible devices is an error, never a silent 1-GPU run).  The
dataset is ``N x (W + K)`` copies of the shape in one list; shape i belongs to rank i mod N (weak scaling).
  --rng-mode dataset (default, exact): ONE sub-sample stream over the whole dataset, as the reference's
      ``--workers 0`` run: every rank also consumes the draws of the shapes it does not own
      (sharding.skip_shape_stream), so every shape's SDF is bit-identical to the single-process run.
  --rng-mode per_shape: shape i is seeded with seed + i (no cross-shape dependency; a declared deviation from the
      reference from the second shape on).
The per-shape SDF arrays are gathered to rank 0 over RCCL at the end of every step (sharding.gather_variable, the
path's only exchange).  Rank 0 prints ONE JSON line.  ``roofline`` is for the dominant kernel (p2s_chain_kernel,
MFMA-bound) from HIP events recorded on the launch stream during the timed steps; ``cpu_baseline`` times the
torch-CPU port of the reference's path (oracle/torch_port.py) on this box's host cores on a bounded sample;
``self_check`` (after the timed region, rank 0) compares a fresh device run of the same shape with that port's
output and with the full-grid golden written by the unmodified reference.
"""
import argparse
import json
import os
import subprocess
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
FIXTURE_SHAPE = '00994122_57d9d4755722f9d2d7436f0a_trimesh_000'
FIXTURE_CLOUD = os.path.join(REPO, 'tests', 'golden', 'abc_minimal', '04_pts', FIXTURE_SHAPE + '.xyz.npy')
GOLDEN_FMT = os.path.join(REPO, 'tests', 'golden', 'ref_rec_p2s_max_testset_grid%d.npz')      # 128, 256 (default), 512


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--points', type=int, default=0,
                    help='0 (default): the abc_minimal fixture cloud; > 0: synthetic cloud of that many points')
    ap.add_argument('--cpu-seconds', type=float, default=25.0, help='target CPU-baseline duration (0 = skip)')
    ap.add_argument('--chunk', type=int, default=0)
    ap.add_argument('--rng-mode', choices=['dataset', 'per_shape'], default='dataset')
    ap.add_argument('--bf16', nargs='?', const=1, default=0, type=int, choices=[0, 1, 2, 3],
                    help='secondary modes (BASELINE configs[3]), NOT the headline metric: bf16 encoder + fp32 decoder '
                         '(--bf16 or --bf16 1); split precision with 2 / 3 bf16 pieces per operand (--bf16 2 / 3)')
    ap.add_argument('--res', type=int, default=GRID_RES,
                    help='query-grid resolution; the headline metric is quoted at 256 (other values: BASELINE configs 1/4)')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL)')
    return ap.parse_args()


def cpu_baseline(w, cfg, cloud, queries, target_seconds, grid_res=GRID_RES):
    """the torch-CPU port on the first n queries of the same workload; returns (record, sdf[:n])"""
    import torch
    from oracle.torch_port import TorchPort
    # torch's default intra-op pool = the physical cores (128 on the MI355X host); all 256 SMT threads measured 13x
    # slower (5.6 vs 75 queries/s).  Pinned explicitly so the figure does not depend on the environment.
    torch.set_num_threads(int(os.environ.get('P2S_CPU_THREADS', max(1, (os.cpu_count() or 2) // 2))))
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
    sdf = port.infer_queries(cloud, queries[:n], rng, batch=500)
    dt = time.time() - t0
    rec = {'value': n / dt, 'unit': 'queries/s', 'cores': int(threads), 'kind': 'port',
           'sample': 'first %d of the same %d^3-grid queries (kNN cKDTree + RandomState sub-sample + torch-CPU '
                     'forward, batch 500), %.1f s' % (n, grid_res, dt),
           'host_cpus': os.cpu_count(), 'torch': torch.__version__,
           'torch_threads': int(threads), 'blas': 'mkl' if torch.backends.mkl.is_available() else 'other'}
    return rec, sdf


def respawn(args):
    """``python bench.py --gpus N`` without a torchrun environment: N ranks on this node, or a loud error"""
    import socket
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and (args.backend or 'nccl') == 'nccl' and not os.environ.get('P2S_BENCH_SHARE_GPU'):
        raise SystemExit('bench.py --gpus %d: %d ranks requested but %d device(s) visible -- refusing to report a '
                         '%d-GPU number from fewer GPUs' % (args.gpus, args.gpus, have, args.gpus))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        respawn(args)
    import torch
    import torch.distributed as dist
    from points2surf_amd import engine, parity, synth, sharding

    world, rank, local_rank = sharding.dist_env()
    if args.gpus != world:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the HIP engine has no CPU fallback')
    share = bool(os.environ.get('P2S_BENCH_SHARE_GPU'))      # rehearsal of the N > 1 control flow on a 1-GPU box (gloo)
    if world > torch.cuda.device_count() and not share:
        raise SystemExit('bench.py: %d ranks, %d device(s)' % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if world > 1:
        if share and (args.backend or 'nccl') == 'nccl':
            raise SystemExit('P2S_BENCH_SHARE_GPU needs --backend gloo (RCCL refuses two ranks on one device)')
        if share:        # init by hand: sharding.init_process_group binds LOCAL_RANK to its own device
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        else:
            sharding.init_process_group(args.backend)
    cdev = sharding.collective_device(torch.device('cuda', torch.cuda.current_device()))

    w, cfg = synth.make_weights('p2s_max')
    if args.bf16:
        cfg = dict(cfg, encoder_bf16=int(args.bf16))
    model = engine.Model(w, cfg)
    model.set_profiling(True)
    if args.points > 0:
        pts = synth.make_cloud(args.points, seed=1000)
        workload = 'synthetic %d-point cloud' % args.points
    else:
        pts = np.ascontiguousarray(np.load(FIXTURE_CLOUD)[:, :3], dtype=np.float32)
        workload = 'abc_minimal test shape %s (%d points; tests/golden/abc_minimal)' % (FIXTURE_SHAPE, pts.shape[0])
    cloud = engine.Cloud(pts)
    n_sub = model.sub_sample_size
    rng = engine.Rng(SEED_DATA)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # dataset = world x (warmup + steps) copies of the shape; shape i -> rank i mod world
    def run_step(step):
        """all shapes of dataset round ``step``: mine is inferred, the others' draws are skipped (exact mode)"""
        out = None
        for r in range(world):
            shape_ind = step * world + r
            if args.rng_mode == 'per_shape':
                if r != rank:
                    continue
                key = np.random.RandomState((SEED_DATA + shape_ind) & 0xffffffff).get_state()[1]   # init_genrand
                rng.set_state(key, 624)
            if r == rank:
                out, _ = engine.infer_shape(model, cloud, rng, args.res, EPSILON, chunk=args.chunk, want_queries=False)
            else:
                sharding.skip_shape_stream(cloud, rng, cfg, args.res, EPSILON, n_sub)
        return out

    sdf = None
    for s in range(args.warmup):
        sdf = run_step(s)
    barrier()
    t0 = time.time()
    n_queries = 0
    acc = {}
    gathered = 0
    for s in range(args.steps):
        sdf = run_step(args.warmup + s)
        n_queries += int(sdf.shape[0])
        for k, v in model.counters().items():
            acc[k] = acc.get(k, 0) + v
        if world > 1:
            # final gather of the variable-length per-shape SDF to rank 0 (RCCL over xGMI): the only exchange
            parts = sharding.gather_variable(sdf, dst=0)
            if rank == 0:
                gathered += sum(int(p.shape[0]) for p in parts)
    barrier()
    dt = time.time() - t0

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        nq = torch.tensor([n_queries], dtype=torch.int64, device=cdev)
        dist.all_reduce(nq, op=dist.ReduceOp.SUM)
        total_queries = int(nq.item())
        if rank == 0 and gathered != total_queries:
            raise SystemExit('gather returned %d SDF values, ranks produced %d' % (gathered, total_queries))
    else:
        total_queries = n_queries

    if rank == 0:
        value = total_queries / dt
        launches = max(acc.get('launches_chain', 0), 1)
        chain_ms = acc.get('ms_chain_stn', 0.0) + acc.get('ms_chain_main', 0.0)
        avg_launch_ms = chain_ms / launches
        flop_per_launch = FLOP_CHAIN_PER_QUERY * n_queries / launches
        # split precision executes 3 (two pieces) / 6 (three pieces) bf16 MFMA passes per algorithmic product: the
        # roofline of those modes is priced in executed bf16 FLOP against the dense bf16 peak
        passes = {0: 1, 1: 1, 2: 3, 3: 6}[int(args.bf16)]
        achieved = passes * flop_per_launch / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
        peak = 2500.0 if args.bf16 else PEAK_FP32_MFMA_TFLOPS     # dense MFMA peak of the compute dtype
        # HBM bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE),
        # committed under profiles/; they cannot be collected from inside this process
        traffic, traffic_src = None, None
        if not args.bf16:
            for rnd in ('r02', 'r01'):
                try:
                    with open(os.path.join(REPO, 'profiles', rnd, 'pmc_summary.json')) as f:
                        ck = json.load(f)['chain_kernel']
                    traffic = ck['hbm_traffic_bytes_per_launch'] * (2.0 * n_queries / launches) / ck['queries_per_launch']
                    traffic_src = ('profiles/%s/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, scaled '
                                   'to this launch size)' % rnd)
                    break
                except Exception:
                    continue
        out = {
            'metric': 'SDF queries/sec/GPU (p2s_max, %d^3 grid%s)' % (args.res, (', bf16 encoder' if args.bf16 == 1 else ', split bf16x%d encoder' % args.bf16) if args.bf16 else ''),
            'value': value, 'unit': 'queries/s',
            'n_gpus': world if not share else 1, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / max(args.steps, 1) * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': ('bf16' if args.bf16 == 1 else 'bf16x%d' % args.bf16) if args.bf16 else 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE.json configs[%d]: p2s_max, grid_res=%d, eps=3, kNN patch=300 / global ' % ({128: 1, 512: 4}.get(args.res, 2), args.res) +
                                   'sub=1000, fp32; %s, one shape per rank per step; seeded random-init weights '
                                   '(Famous set / pretrained weights not available offline)' % workload,
                       'queries_per_shape_rank0': int(sdf.shape[0]),
                       'parallelism': 'shape-sharded x%d' % world + (' (REHEARSAL: all ranks share one GPU, gloo)' if share else ''),
                       'rng_mode': args.rng_mode,
                       'shapes_per_hour': world * args.steps / dt * 3600.0,
                       'queries_per_s_per_gpu': value / world},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': 'p2s_chain_bf16_kernel' if args.bf16 else 'p2s_chain_kernel', 'launches': int(launches), 'avg_launch_ms': avg_launch_ms,
                         'algorithmic_flop_per_launch': flop_per_launch, 'mfma_passes_per_product': passes,
                         'hbm_algorithmic_GBps': value / world * BYTES_PER_QUERY / 1e9},
            'stage_ms_rank0': {k: acc[k] for k in sorted(acc) if k.startswith('ms_')},
        }
        # ---- after the timed region: the output of this very workload against the checkers --------------------
        check = {}
        rng_chk = engine.Rng(SEED_DATA)
        sdf_chk, _ = engine.infer_shape(model, cloud, rng_chk, args.res, EPSILON, chunk=args.chunk, want_queries=False)
        sdf_chk = sdf_chk.cpu().numpy()
        tol = 0.25 if args.bf16 == 1 else 1e-4      # north_star: SDF within 1e-4 fp32 of the reference (plain bf16: reported only)
        golden = GOLDEN_FMT % args.res
        if args.points == 0 and os.path.isfile(golden):
            ref = np.load(golden)['rec_0']
            ok = ref.shape == sdf_chk.shape
            rec = {'file': os.path.relpath(golden, REPO), 'queries': int(ref.shape[0])}
            if ok:
                cmp_ = parity.compare_sdf(sdf_chk, ref)
                fl = cmp_['flipped']
                rec.update({'max_abs_dsdf': cmp_['max_abs_dsdf'], 'sign_flips': int(fl.size)})
                if 0 < fl.size <= 8 and not args.bf16:
                    # sign = (sign logit >= 0): a flip is an fp32 TIE iff the device's own sign logit is within the logit
                    # accuracy of zero (the reference's answer for such a query depends on its batch composition / threads)
                    q_all = cloud.query_grid(args.res, EPSILON)
                    lg = [float(engine.query_logits(model, cloud, engine.Rng(SEED_DATA), q_all, int(j))[1]) for j in fl]
                    rec['flipped_sign_logits'] = lg
                    rec['sign_flips_not_ties'] = parity.not_ties(lg)
                else:
                    rec['sign_flips_not_ties'] = int(fl.size)
            check['vs_reference_golden'] = rec
            if not ok or rec['max_abs_dsdf'] > tol or (not args.bf16 and rec['sign_flips_not_ties'] != 0):
                out['self_check'] = check
                print(json.dumps(out), flush=True)
                raise SystemExit('bench.py self-check FAILED against the reference golden: %s' % check)
        if args.cpu_seconds > 0 and world == 1:
            q = cloud.query_grid(args.res, EPSILON).cpu().numpy()
            out['cpu_baseline'], sdf_cpu = cpu_baseline(w, cfg, pts, q, args.cpu_seconds, args.res)
            n = sdf_cpu.shape[0]
            check['vs_cpu_port'] = {'queries': int(n), 'max_abs_dsdf': float(np.abs(sdf_cpu - sdf_chk[:n]).max()),
                                    'sign_flips': int((np.sign(sdf_cpu) != np.sign(sdf_chk[:n])).sum())}
            if check['vs_cpu_port']['max_abs_dsdf'] > tol or (not args.bf16 and check['vs_cpu_port']['sign_flips']):
                out['self_check'] = check
                print(json.dumps(out), flush=True)
                raise SystemExit('bench.py self-check FAILED against the CPU port: %s' % check)
        elif world == 1:
            out['cpu_baseline'] = None
        out['self_check'] = check
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
