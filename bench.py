#!/usr/bin/env python
"""Headline benchmark: SDF queries/s per GPU, p2s_max, 256^3 query grid (BASELINE.json metric).

A *step* = one COMPLETE shape per rank, host to host (reference source/points_to_surf_eval.py:358-404 for one shape):
the cloud starts in host memory -> upload -> neighbour index built on the device (the reference builds a cKDTree,
source/data_loader.py:40-42) -> near-surface query grid of the 256^3 volume (source/sdf.py:46-79) -> for every query
the 300-NN patch (fp64-exact), the 1000-point global sub-sample (numpy-legacy MT19937 stream), the PointNet encoders +
decoder -> SDF -> download to host memory.  A fresh cloud handle per step: nothing is cached between steps.
The dataset is the three committed ``abc_minimal`` clouds (tests/golden/abc_minimal/04_pts: 59,979 / 86,648 / 34,693
points; Q = 572 k / 499 k / 307 k queries at 256^3, eps 3) taken round-robin as ONE dataset with one sub-sample stream
-- the inputs the parity tests pin against the unmodified reference -- with seeded random-init weights (no pretrained
weights offline).

  python bench.py --gpus N --steps K --warmup W [--model p2s_max|p2s_vanilla] [--bf16 M] [--rng-mode dataset|per_shape]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

``--model p2s_vanilla --bf16 4`` is BASELINE configs[3] (p2s_vanilla with QSTN, reduced-precision encoder + fp32 decoder)
at whatever N the launcher gives; the default is the headline, configs[2] (p2s_max, fp32).

N > 1: one process per GPU (``--gpus N`` without a torchrun environment re-executes itself under
``torch.distributed.run`` with N ranks; fewer than N visible devices is an error, never a silent 1-GPU run).  The
dataset is ``N x (W + K)`` shapes in one list, assigned by sharding.assign_shapes (LPT over query counts); every round of
N shapes is one cloud (weak scaling).
  --rng-mode dataset (default, exact): ONE sub-sample stream over the whole dataset, as the reference's ``--workers 0``
      run.  With a process group the generator state is handed from shape to shape through the rendezvous store
      (sharding.StreamHandoff); every shape's SDF is bit-identical to the single-process run, and rank 0 verifies that.
  --rng-mode per_shape: shape i is seeded with seed + i (no cross-shape dependency; a declared deviation from the
      reference from the second shape on).
The per-shape SDF arrays are gathered to rank 0 over RCCL once, at the end of the timed region (sharding.gather_variable,
the path's only exchange).  Rank 0 prints ONE JSON line.
  roofline        the dominant kernel (p2s_chain_kernel, MFMA-bound) from HIP events recorded on the launch stream during
                  the timed steps
  cpu_baseline    the reference's CPU path timed on this box's host cores (rank 0, after the timed region, at every N): the
                  UNMODIFIED reference through oracle/ref_shims.py where a reference checkout exists (kind "reference"),
                  else the torch-CPU port oracle/torch_port.py (kind "port"); thread count chosen by a 3-point probe
  secondary       (N = 1, headline model only) configs[3]'s model and the reduced-precision encoders: per entry one warm-up
                  and THREE timed complete shapes of the test shape (median, all three values, stage_ms, non_chain_ms), each
                  checked over the full grid against the reference's golden; ``skip``: the cost of keeping the exact stream
                  under sharding (t_s, t_i, modelled efficiency); ``dropin_*``: the same workload through boundary B1
  self_check      (after the timed region, rank 0) the dataset from a fresh stream against the goldens written by the
                  unmodified reference; a flipped sign passes only as a verified tie of the encoder mode's own threshold
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

# algorithmic FLOP per query, SURVEY.md 8(a)/(d) (torch FlopCounter == hand count).  ``chain``: the per-point layers +
# max-pool inputs (everything the chain kernel replaces) = total minus the per-query FC layers (STN heads 1,703,936 MAC
# each, decoder 1,343,744 MAC; QSTN: 181,292,800 MAC per-point, 668,100 MAC of FC layers + rotations)
MODEL_FLOP = {
    'p2s_max': {'total': 776773632, 'chain': 776773632 - 2 * (2 * 1703936 + 1343744), 'configs': 2},
    'p2s_vanilla': {'total': 1140695432, 'chain': 776773632 - 2 * (2 * 1703936 + 1343744) + 2 * 181292800, 'configs': 3},
}
BYTES_PER_QUERY = 15620             # minimal HBM traffic per query, SURVEY.md 8(d)
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md
PEAK_16BIT_MFMA_TFLOPS = 2500.0     # dense bf16 / fp16
GRID_RES, EPSILON = 256, 3
SEED_DATA = 40938661
GOLDEN = os.path.join(REPO, 'tests', 'golden')
ABC3 = ['00011084_fddd53ce45f640f3ab922328_trimesh_019', '00016513_3d6966cd42eb44ab8f4224f2_trimesh_053',
        '00994122_57d9d4755722f9d2d7436f0a_trimesh_000']          # tests/golden/abc_minimal/abc3.txt (dataset order)
FIXTURE_SHAPE = ABC3[2]                                           # abc_minimal/testset.txt
MFMA_PASSES = {0: 1, 1: 1, 2: 3, 3: 6, 4: 3}                      # executed MFMA passes per algorithmic product
DTYPE = {0: 'f32', 1: 'bf16', 2: 'bf16x2', 3: 'bf16x3', 4: 'fp16x2'}
ENCODER_TEXT = {0: '', 1: ', bf16 encoder', 2: ', split bf16x2 encoder', 3: ', split bf16x3 encoder', 4: ', fp16-pair encoder'}


def cloud_path(name):
    return os.path.join(GOLDEN, 'abc_minimal', '04_pts', name + '.xyz.npy')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--model', choices=sorted(MODEL_FLOP), default='p2s_max',
                    help='p2s_max: the headline (BASELINE configs[2]); p2s_vanilla: configs[3]\'s model (QSTN + weighted sub-sample)')
    ap.add_argument('--dataset', choices=['abc3', 'fixture'], default=None,
                    help='abc3 (default at 256^3): the three abc_minimal clouds round-robin; fixture: the test shape only')
    ap.add_argument('--points', type=int, default=0,
                    help='> 0: a synthetic cloud of that many points instead of the abc_minimal clouds (no golden check)')
    ap.add_argument('--cpu-seconds', type=float, default=25.0, help='target CPU-baseline duration (0 = skip)')
    ap.add_argument('--chunk', type=int, default=0)
    ap.add_argument('--rng-mode', choices=['dataset', 'per_shape'], default='dataset')
    ap.add_argument('--bf16', nargs='?', const=1, default=0, type=int, choices=[0, 1, 2, 3, 4],
                    help='encoder arithmetic (BASELINE configs[3]), NOT the headline metric: bf16 encoder + fp32 decoder '
                         '(--bf16 or --bf16 1); split precision with 2 / 3 bf16 pieces per operand (--bf16 2 / 3); fp16 pair per operand '
                         '(--bf16 4: 3 fp16 MFMAs per product, fp32 accuracy)')
    ap.add_argument('--res', type=int, default=GRID_RES,
                    help='query-grid resolution; the headline metric is quoted at 256 (other values: BASELINE configs 1/4)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary passes')
    ap.add_argument('--secondary-reps', type=int, default=3, help='timed shapes per secondary entry (median reported)')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL)')
    # test hook (tests/test_gpu_bench_rehearsal.py): RANK:K -- that rank raises when it reaches its K-th own shape (0-based)
    ap.add_argument('--fault', default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (the checker timed, never the product)
# ------------------------------------------------------------------------------------------------------------------
def _probe_threads(port, cloud, queries):
    """3-point probe of torch's intra-op thread count (8 / cores/4 / cores/2): the one with the most queries/s.  All SMT
    threads of the MI355X host measured 13x slower than its physical cores; too few leave the GEMMs under-fed."""
    import torch
    cores = os.cpu_count() or 2
    fixed = os.environ.get('P2S_CPU_THREADS')
    cands = [int(fixed)] if fixed else sorted({max(1, min(8, cores)), max(1, cores // 4), max(1, cores // 2)})
    best, probe = None, {}
    for t in cands:
        torch.set_num_threads(t)
        rng = np.random.RandomState(SEED_DATA)
        port.infer_queries(cloud, queries[:16], rng, batch=16)                # touch the code path at this thread count
        rng = np.random.RandomState(SEED_DATA)
        t0 = time.time()
        port.infer_queries(cloud, queries[:96], rng, batch=96)
        probe[t] = 96.0 / (time.time() - t0)
        if best is None or probe[t] > probe[best]:
            best = t
    torch.set_num_threads(best)
    return best, probe


def _reference_leg(model_name, cloud_name, threads):
    """the UNMODIFIED reference's points_to_surf_eval (torch CPU) over the 32^3 query grid of one abc_minimal shape, through
    the five external shims of oracle/ref_shims.py.  Only where a reference checkout exists
    (P2S_REFERENCE_ROOT / /root/reference: the build container, never the GPU box).  -> (queries, seconds, sdf)"""
    import shutil
    import tempfile
    import torch
    from oracle import ref_shims
    from points2surf_amd import synth
    tmp = tempfile.mkdtemp(prefix='p2s_bench_ref_')
    try:
        root = os.path.join(tmp, 'abc_minimal')
        os.makedirs(os.path.join(root, '04_pts'))
        shutil.copyfile(cloud_path(cloud_name), os.path.join(root, '04_pts', cloud_name + '.xyz.npy'))
        with open(os.path.join(root, 'testset.txt'), 'w') as f:
            f.write(cloud_name + '\n')
        modeldir = os.path.join(tmp, 'models')
        synth.write_model_files(modeldir, model_name)
        torch.set_num_threads(threads)
        with ref_shims.reference():
            from source import points_to_surf_eval as ref_eval
            opt = ref_eval.parse_arguments(['--indir', root, '--outdir', os.path.join(tmp, 'out'), '--dataset', 'testset.txt',
                                            '--modeldir', modeldir, '--models', model_name, '--query_grid_resolution', '32',
                                            '--epsilon', str(EPSILON), '--certainty_threshold', '13', '--sigma', '5',
                                            '--gpu_idx', '-1', '--workers', '0', '--batchSize', '500', '--cache_capacity', '5'])
            opt.reconstruction = True
            t0 = time.time()
            ref_eval.points_to_surf_eval(opt)                                   # reference source/points_to_surf_eval.py:297-404
            dt = time.time() - t0
        sdf = np.load(os.path.join(tmp, 'out', 'rec', 'dist_ms', cloud_name + '.xyz.npy')).astype(np.float32)
        return int(sdf.shape[0]), dt, sdf
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(w, cfg, model_name, cloud_name, cloud, queries, target_seconds, grid_res=GRID_RES):
    """-> (record, sdf of the sample, 'grid32' | 'prefix'): what the sample is -- the whole 32^3 grid of the shape
    (reference leg) or the first n queries of the benchmarked grid (port)"""
    import torch
    from oracle import ref_shims
    from oracle.torch_port import TorchPort
    port = TorchPort(w, cfg)
    threads, probe = _probe_threads(port, cloud, queries)
    rec = {'unit': 'queries/s', 'cores': int(threads), 'host_cpus': os.cpu_count(), 'torch': torch.__version__,
           'torch_threads': int(threads), 'thread_probe_queries_per_s': {str(k): v for k, v in sorted(probe.items())},
           'blas': 'mkl' if torch.backends.mkl.is_available() else 'other'}
    if ref_shims.reference_available() and cloud_name is not None:
        n, dt, sdf = _reference_leg(model_name, cloud_name, threads)
        rec.update({'value': n / dt, 'kind': 'reference', 'same_box': True,
                    'sample': 'the UNMODIFIED reference (source.points_to_surf_eval.points_to_surf_eval, torch CPU, --workers 0, '
                              'batch 500) over the %d queries of the 32^3 grid of the dataset\'s shape %s -- the same per-query '
                              'work as at %d^3 (kNN 300 / sub-sample 1000 / same network), %.1f s' % (n, cloud_name[:8], grid_res, dt)})
        return rec, sdf, 'grid32'
    rng = np.random.RandomState(SEED_DATA)
    n = int(min(max(target_seconds * probe[threads], 64), 4096, queries.shape[0]))
    t0 = time.time()
    sdf = port.infer_queries(cloud, queries[:n], rng, batch=500)
    dt = time.time() - t0
    ref_note = None
    try:
        with open(os.path.join(GOLDEN, 'meta_sizes.json')) as f:
            m = json.load(f)['ref_rec_%s_testset_grid256' % model_name]
        ref_note = ('the UNMODIFIED reference (points_to_surf_eval, torch CPU, %d threads, build container) made %.1f '
                    'queries/s on the full 256^3 grid of the test shape when it wrote the golden '
                    '(tests/golden/meta_sizes.json)' % (m['threads'], m['reference_queries_per_s']))
    except Exception:
        pass
    # same_box: is `value` the REFERENCE ITSELF timed on this box?  No: no checkout exists here, `value` is the port on this
    # box's cores, and `reference_itself` quotes the reference's own rate from ANOTHER machine (the build container)
    rec.update({'value': n / dt, 'kind': 'port', 'same_box': False,
                'sample': 'oracle/torch_port.py (no reference checkout on this box): first %d of the %d^3-grid queries of the '
                          'dataset\'s first shape (kNN cKDTree + RandomState sub-sample + torch-CPU forward, batch 500), %.1f s'
                          % (n, grid_res, dt),
                'reference_itself': ref_note})
    return rec, sdf, 'prefix'


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


def reseed(rng, seed=SEED_DATA):
    """``RandomState(seed)`` on an existing device generator (init_genrand): the handle, its stream buffers and the
    weighted sub-sample's tables stay allocated -- a timed pass never allocates"""
    rng.set_state(np.random.RandomState(int(seed) & 0xffffffff).get_state()[1], 624)


def complete_shape(engine, model, pts_host, rng, res, chunk, ev=None):
    """ONE complete shape, host to host: upload + index build + query grid + inference + download.  A fresh cloud handle
    (no cached grid).  ``ev``: optional dict of lists collecting (cloud build, grid, download) milliseconds from events.
    Returns (SDF in host memory, the device tensor it was copied from)."""
    import torch
    e0 = e1 = e1b = e2 = e3 = None
    if ev is not None:
        e0, e1, e1b, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(5))
        e0.record()
    cloud = engine.Cloud(pts_host)                  # H2D + p2s_cloud_create (bbox, histogram, SAT, counting sort)
    if ev is not None:
        e1.record()
    nq = cloud.count_queries(res, EPSILON)          # a1: voxelise, dilate, count (one host sync), compact
    if ev is not None:
        e1b.record()
    sdf, _ = engine.infer_shape(model, cloud, rng, res, EPSILON, chunk=chunk, want_queries=False, n_queries=nq)
    if ev is not None:
        e2.record()
    out = sdf.cpu()                                 # D2H; synchronises
    if ev is not None:
        e3.record()
        e3.synchronize()
        ev.setdefault('ms_cloud', []).append(e0.elapsed_time(e1))
        ev.setdefault('ms_grid', []).append(e1.elapsed_time(e1b))
        ev.setdefault('ms_d2h', []).append(e2.elapsed_time(e3))
    cloud.close()
    return out, sdf


def dropin_leg(shapes, res, encoder, model='p2s_max'):
    """The hot path measured THROUGH the boundary the north_star names (B1): the drop-in's
    ``source.points_to_surf_eval.points_to_surf_eval(opt)`` in reconstruction mode over the dataset (files loaded from
    disk, all result files written), followed by ``source.sdf.implicit_surface_to_mesh_directory`` -- the sequence and the
    timed region of the reference's full_eval.py:44-64.  Returns (record for ``secondary.dropin_<encoder>``, the SDF arrays
    it wrote, in dataset order)."""
    import shutil
    import tempfile
    from points2surf_amd import synth
    dropin = os.path.join(REPO, 'points2surf_amd', 'dropin')
    if dropin not in sys.path:
        sys.path.insert(0, dropin)
    from source import points_to_surf_eval as ev
    from source import sdf as dsdf
    tmp = tempfile.mkdtemp(prefix='p2s_bench_dropin_')
    prev = os.environ.get('P2S_ENCODER')
    try:
        root = os.path.join(tmp, 'abc_minimal')
        os.makedirs(os.path.join(root, '04_pts'))
        for name, pts, _ in shapes:
            np.save(os.path.join(root, '04_pts', name + '.xyz.npy'), pts)
        with open(os.path.join(root, 'testset.txt'), 'w') as f:
            f.write('\n'.join(n for n, _, _ in shapes) + '\n')
        modeldir = os.path.join(tmp, 'models')
        synth.write_model_files(modeldir, model)
        os.environ['P2S_ENCODER'] = encoder
        stats = {}

        def run(outdir, grid):
            opt = ev.parse_arguments(['--indir', root, '--outdir', outdir, '--dataset', 'testset.txt', '--modeldir', modeldir,
                                      '--models', model, '--query_grid_resolution', str(grid), '--epsilon', str(EPSILON),
                                      '--certainty_threshold', '13', '--sigma', '5', '--workers', '7', '--batchSize', '0'])
            opt.reconstruction = True                          # full_eval.py:45
            t0 = time.time()
            ev.points_to_surf_eval(opt)                        # full_eval.py:46
            t1 = time.time()
            stats.clear()
            stats.update(ev.last_run_stats)
            rec = os.path.join(outdir, 'rec')
            dsdf.implicit_surface_to_mesh_directory(            # full_eval.py:51-64
                os.path.join(rec, 'dist_ms'), os.path.join(rec, 'query_pts_ms'), os.path.join(rec, 'vol'),
                os.path.join(rec, 'mesh'), opt.query_grid_resolution, opt.sigma, opt.certainty_threshold, opt.workers)
            return t1 - t0, time.time() - t1, rec

        import contextlib
        with contextlib.redirect_stdout(sys.stderr):            # the drop-in prints what the reference prints; stdout carries
            run(os.path.join(tmp, 'warm'), 64)                  # the ONE JSON line only.  warm-up: block cache, page cache
            t_eval, t_mesh, rec = run(os.path.join(tmp, 'out'), res)
        nq, files, sdfs = 0, 0, []
        for name, _, _ in shapes:
            sdf = np.load(os.path.join(rec, 'dist_ms', name + '.xyz.npy'))
            sdfs.append(sdf)
            nq += int(sdf.shape[0])
            for sub, ext in (('eval', '.xyz.npy'), ('eval', '.xyz.txt'), ('query_pts_ms', '.xyz.npy'), ('vis', '.ply'),
                             ('query_pts_ms_vis', '.ply'), ('vol', '.off'), ('mesh', '.ply')):
                files += int(os.path.getsize(os.path.join(rec, sub, name + ext)) > 0)
        return {'value': nq / t_eval, 'unit': 'queries/s', 'queries': nq, 'shapes': len(shapes), 'encoder': encoder,
                'seconds_points_to_surf_eval': t_eval, 'seconds_mesh_directory': t_mesh,
                'seconds_shape_loop': stats.get('seconds_shapes'), 'seconds_model_create': stats.get('seconds_model_create'),
                'value_shape_loop': nq / max(stats.get('seconds_shapes') or t_eval, 1e-9),
                'shapes_per_hour_eval': len(shapes) / t_eval * 3600.0,
                'shapes_per_hour_incl_mesh': len(shapes) / (t_eval + t_mesh) * 3600.0,
                'files_written': files,
                'what': 'source.points_to_surf_eval.points_to_surf_eval(opt) of the drop-in (reconstruction pass: clouds '
                        'loaded from .npy files, eval/.npy + .txt, dist_ms, query_pts_ms and both visualisation PLYs '
                        'written) then source.sdf.implicit_surface_to_mesh_directory (.off + mesh .ply per shape), timed '
                        'like full_eval.py:44-64; model load included'}, sdfs
    finally:
        if prev is None:
            os.environ.pop('P2S_ENCODER', None)
        else:
            os.environ['P2S_ENCODER'] = prev
        shutil.rmtree(tmp, ignore_errors=True)


def golden_check(engine, parity, model, w, cfg, shapes, res, sdfs, tol, bf16, stride=1):
    """every query of the run ``sdfs`` (one array per shape, one stream from SEED_DATA in dataset order) against the
    goldens the unmodified reference wrote.  Signs: a flipped sign is accepted as a TIE only if BOTH the device's own sign
    logit and the CPU port's sign logit for that query (same inputs) lie within the encoder mode's tie threshold of zero
    (parity.tie_logit: 1e-5 for the fp32 encoder, 2e-5 for the split-precision modes).
    ``stride`` > 1: the golden holds every ``stride``-th query of every shape (the data-path golden's network subset)."""
    rec = {'shapes': [], 'queries': 0, 'max_abs_dsdf': 0.0, 'max_abs_diff_unmasked': 0.0, 'sign_flips': 0,
           'sign_flips_not_ties': 0, 'flipped': [], 'tie_logit': parity.tie_logit(bf16)}
    ok = True
    for si, (name, pts, ref) in enumerate(shapes):
        if ref is None:
            rec['shapes'].append({'shape': name[:8], 'golden': None})
            continue
        sdf = sdfs[si][::stride]
        if ref.shape != sdf.shape:
            rec['shapes'].append({'shape': name[:8], 'error': 'shape %s vs golden %s' % (sdf.shape, ref.shape)})
            ok = False
            continue
        cmp_ = parity.compare_sdf(sdf, ref)
        fl = cmp_['flipped']
        unmasked = float(np.abs(sdf - ref).max()) if sdf.size else 0.0
        srec = {'shape': name[:8], 'queries': int(ref.shape[0]), 'max_abs_dsdf': cmp_['max_abs_dsdf'],
                'max_abs_diff_unmasked': unmasked, 'sign_flips': int(fl.size)}
        rec['queries'] += int(ref.shape[0])
        rec['max_abs_dsdf'] = max(rec['max_abs_dsdf'], cmp_['max_abs_dsdf'])
        rec['max_abs_diff_unmasked'] = max(rec['max_abs_diff_unmasked'], unmasked)
        rec['sign_flips'] += int(fl.size)
        not_ties = int(fl.size)
        if 0 < fl.size <= 32 and bf16 in (0, 3, 4):   # ~1 query in 400,000 has a sign logit within fp32 noise of zero
            # (split-precision encoders 3 / 4 claim fp32 accuracy: their flips are classified the same way)
            not_ties = 0
            from oracle.torch_port import TorchPort
            from points2surf_amd import sharding
            for j in fl:
                # position a stream at this shape's first draw: skip the shapes before it
                rng = engine.Rng(SEED_DATA)
                for name2, pts2, _ in shapes[:si]:
                    c2 = engine.Cloud(pts2)
                    sharding.skip_shape_stream(c2, rng, cfg, res, EPSILON, model.sub_sample_size)
                    c2.close()
                cloud = engine.Cloud(pts)
                q_all = cloud.query_grid(res, EPSILON)
                patch, sub, one = engine.query_inputs(model, cloud, rng, q_all, int(j) * stride)
                lg_dev = float(model.forward(patch, sub, one)[0][0, 1])
                lg_cpu = float(TorchPort(w, cfg).forward(patch.cpu().numpy(), sub.cpu().numpy(), one.cpu().numpy())[0, 1])
                tie = parity.is_tie(lg_dev, lg_cpu, bf16)
                not_ties += 0 if tie else 1
                rec['flipped'].append({'shape': name[:8], 'query': int(j) * stride, 'sdf': float(sdf[j]), 'ref': float(ref[j]),
                                       'sign_logit_device': lg_dev, 'sign_logit_cpu_port': lg_cpu, 'tie': bool(tie)})
                cloud.close()
                rng.close()
        elif fl.size:
            rec['flipped'] += [{'shape': name[:8], 'query': int(j) * stride, 'sdf': float(sdf[j]), 'ref': float(ref[j])} for j in fl[:16]]
        srec['sign_flips_not_ties'] = not_ties
        rec['sign_flips_not_ties'] += not_ties
        rec['shapes'].append(srec)
        # fp32 and the split-precision encoders 3 / 4 (advertised as fp32-accurate): any flipped sign that is not a
        # verified tie fails the check -- more than 32 flips are never classified, so they fail as well; plain bf16 /
        # two bf16 pieces (modes 1 / 2, outside the contract) are reported only
        if cmp_['max_abs_dsdf'] > tol or (bf16 in (0, 3, 4) and not_ties):
            ok = False
    return rec, ok


def datapath_check(engine, shapes, res, golden, meta, batch=4096, progress=None):
    """The DATA PATH of a whole data set against the unmodified reference's ``PointcloudPatchDataset.__getitem__``
    (tests/golden/ref_datapath_<model>_<dataset>_grid<res>.npz, written by oracle/make_golden_datapath.py): the shapes in
    dataset order from ONE generator, per block of 1024 queries the sha256 of the distance-weighted sub-sample ids
    (p2s_subsample_weighted: every ``choice(N, 1000, replace=False, p)`` draw of the run, bit for bit), of the kNN patches
    in patch space and of the radii (p2s_knn_patch), the generator state after every batch (where both sides hold it in
    the same representation) and, through the next draws, after every shape.  -> (record, ok)"""
    import hashlib
    import torch
    blk = int(meta['block'])
    assert batch % blk == 0
    rng = engine.Rng(int(meta['seed']))
    rec = {'shapes': [], 'queries': 0, 'blocks': 0, 'ids_blocks_differ': 0, 'patch_blocks_differ': 0, 'radius_blocks_differ': 0,
           'state_checkpoints': 0, 'state_checkpoints_differ': 0, 'state_checkpoints_other_representation': 0,
           'state_after_shape_equal': []}
    ok = True

    def digests(a):
        a = np.ascontiguousarray(a)
        return [hashlib.sha256(a[i:i + blk].tobytes()).digest() for i in range(0, a.shape[0], blk)]

    for si, (name, pts, _) in enumerate(shapes):
        ms = meta['shapes'][si]
        cloud = engine.Cloud(pts)
        q = cloud.query_grid(res, EPSILON)
        nq = int(q.shape[0])
        srec = {'shape': name[:8], 'queries': nq,
                'query_points_equal': hashlib.sha256(np.ascontiguousarray(q.cpu().numpy()).tobytes()).hexdigest() == ms['query_sha256']
                and nq == ms['queries'] == ms['queries_run']}
        ok = ok and srec['query_points_equal']
        want = {k: [bytes(r) for r in golden['%s_sha_%d' % (k, si)]] for k in ('ids', 'patch', 'radius', 'state')}
        bad = {'ids': 0, 'patch': 0, 'radius': 0}
        for a in range(0, nq, batch):
            b = min(a + batch, nq)
            qb = q[a:b].contiguous()
            ids = rng.subsample_weighted(cloud, qb, int(meta.get('sub_sample_size', 1000)), want_pts=False)[0]
            _, patch, rad = cloud.knn_patch(qb, int(meta.get('points_per_patch', 300)), want_ids=False)
            got = {'ids': digests(ids.cpu().numpy()), 'patch': digests(patch.cpu().numpy()), 'radius': digests(rad.cpu().numpy())}
            b0 = a // blk
            for k in bad:
                bad[k] += sum(1 for i, d in enumerate(got[k]) if d != want[k][b0 + i])
            mt, pos = rng.get_state()           # the state after the last block of this batch
            dg = hashlib.sha256(np.ascontiguousarray(mt, dtype=np.uint32).tobytes() + np.int32(pos).tobytes()).digest()
            rec['state_checkpoints'] += 1
            if dg != want['state'][b0 + len(got['ids']) - 1]:
                if pos in (0, 624):             # numpy twists lazily (pos 624), the device may hold the twisted array (pos 0)
                    rec['state_checkpoints_other_representation'] += 1
                else:
                    rec['state_checkpoints_differ'] += 1
            if progress:
                progress(si, b, nq)
        rng.check()
        # the generator after the shape, through the next draws (representation-independent): numpy from the golden's
        # state against the device from a snapshot
        rs = np.random.RandomState(0)
        rs.set_state(('MT19937', golden['state_key_%d' % si], int(golden['state_pos_%d' % si]), 0, 0.0))
        snap = rng.get_state()
        tail = rng.subsample_uniform(cloud, 1, 64, want_pts=False)[0].cpu().numpy().reshape(-1)
        rng.set_state(*snap)
        same = bool(np.array_equal(tail, rs.randint(0, pts.shape[0], 64)))
        rec['state_after_shape_equal'].append(same)
        srec.update({k + '_blocks_differ': v for k, v in bad.items()})
        rec['shapes'].append(srec)
        rec['queries'] += nq
        rec['blocks'] += (nq + blk - 1) // blk
        for k, v in bad.items():
            rec[k + '_blocks_differ'] += v
        ok = ok and same and not any(bad.values())
        cloud.close()
        torch.cuda.synchronize()
    rng.close()
    ok = ok and rec['state_checkpoints_differ'] == 0 and rec['state_checkpoints_other_representation'] <= 8
    rec['bit_identical'] = bool(ok)
    return rec, ok


def handoff_efficiency(t_i, t_s, n):
    """modelled weak-scaling efficiency of the exact dataset stream with the token hand-off: the ring passes one token per
    t_s, a rank needs t_i + t_s per own shape (multi-GPU)"""
    return 1.0 if n <= 1 else t_i / max(t_i + t_s, n * t_s)


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
    if world > 1 or os.environ.get('P2S_DIST_FORCE'):
        if share and (args.backend or 'nccl') == 'nccl':
            raise SystemExit('P2S_BENCH_SHARE_GPU needs --backend gloo (RCCL refuses two ranks on one device)')
        if share:        # init by hand: sharding.init_process_group binds LOCAL_RANK to its own device
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        else:
            sharding.init_process_group(args.backend)
    cdev = sharding.collective_device(torch.device('cuda', torch.cuda.current_device()))

    mname, bf16 = args.model, int(args.bf16)
    flop = MODEL_FLOP[mname]
    w, cfg = synth.make_weights(mname)
    if bf16:
        cfg = dict(cfg, encoder_bf16=bf16)
    model = engine.Model(w, cfg)
    model.set_profiling(True)

    # ---- the dataset: (name, host cloud, golden SDF or None) in dataset order ---------------------------------
    dataset = args.dataset or ('abc3' if args.res == GRID_RES else 'fixture')
    shapes = []
    if args.points > 0:
        shapes.append(('synthetic%d' % args.points, synth.make_cloud(args.points, seed=1000), None))
        workload = 'synthetic %d-point cloud' % args.points
        golden_file = None
    else:
        names = ABC3 if dataset == 'abc3' else [FIXTURE_SHAPE]
        golden_file = os.path.join(GOLDEN, 'ref_rec_%s_%s_grid%d.npz' % (mname, 'abc3' if dataset == 'abc3' else 'testset', args.res))
        g = np.load(golden_file) if os.path.isfile(golden_file) else None
        if g is None and dataset == 'fixture' and args.res == 32:       # the stage-wise fixture of oracle/make_golden.py
            golden_file = os.path.join(GOLDEN, 'ref_%s_grid32.npz' % mname)
            g = {'rec_0': np.load(golden_file)['sdf_full']}
        for i, n in enumerate(names):
            pts = np.ascontiguousarray(np.load(cloud_path(n))[:, :3], dtype=np.float32)
            shapes.append((n, pts, None if g is None else g['rec_%d' % i]))
        workload = ('abc_minimal clouds %s (%s points; tests/golden/abc_minimal) round-robin as one dataset'
                    % (', '.join(n[:8] for n in names), ' / '.join(str(s[1].shape[0]) for s in shapes)))
    n_sub = model.sub_sample_size
    rng = engine.Rng(SEED_DATA)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # dataset = world x (warmup + steps) shapes in one list; shape g = cloud (g // world) mod 3 (every round of ``world``
    # shapes is one cloud: weak scaling).  Owner of a shape: sharding.assign_shapes (LPT over the query counts, the
    # drop-in's policy), separately for the warm-up block and the timed block so that every rank has timed work.
    ev = {}
    n_rounds = args.warmup + args.steps
    cloud_of = [(g // world) % len(shapes) for g in range(n_rounds * world)]
    q_of_cloud = []
    for _, pts, _ in shapes:                       # untimed: the query count of each distinct cloud
        c = engine.Cloud(pts)
        q_of_cloud.append(c.count_queries(args.res, EPSILON))
        c.close()
    owner = []
    for lo, hi in ((0, args.warmup * world), (args.warmup * world, n_rounds * world)):
        _, own = sharding.assign_shapes([q_of_cloud[cloud_of[g]] for g in range(lo, hi)], world)
        owner += own
    handoff = None
    if world > 1 and args.rng_mode == 'dataset' and sharding.stream_handoff_enabled():
        handoff = sharding.StreamHandoff('bench', owner, rank=rank)
    stream_mode = ('per_shape' if args.rng_mode == 'per_shape' else
                   ('dataset/handoff' if handoff is not None else ('dataset/replicate' if world > 1 else 'dataset')))

    def skip_shape(g, r):
        c2 = engine.Cloud(shapes[cloud_of[g]][1])
        sharding.skip_shape_stream(c2, r, cfg, args.res, EPSILON, n_sub)
        c2.close()

    # N > 1: every rank leaves, per own shape, the digest of the generator state at its first draw and after its last one and
    # the sha256 of its SDF in the rendezvous store (control plane, 200 bytes): rank 0 replays the whole dataset stream after
    # the timed region and compares (self_check.stream_handoff)
    store = None
    if sharding.is_initialized():
        from torch.distributed.distributed_c10d import _get_default_store
        store = _get_default_store()
    fault = tuple(int(x) for x in args.fault.split(':')) if args.fault else None
    own_seen = [0]

    def state_digest(r):
        import hashlib
        return hashlib.sha256(sharding.StreamHandoff.pack([r])).hexdigest()

    def run_block(lo, hi, timed):
        """shapes lo..hi-1 of the dataset in order: mine are inferred (complete shapes, host to host); returns the list
        of (host SDF, device SDF, cloud index)"""
        import contextlib
        import hashlib
        outs = []
        for g in range(lo, hi):
            pts = shapes[cloud_of[g]][1]
            if owner[g] != rank:
                if args.rng_mode == 'dataset' and handoff is None:
                    skip_shape(g, rng)               # replicate mode: index + voxelise the shape to consume its draws
                continue
            with (handoff.guard(g) if handoff is not None else contextlib.nullcontext()):
                if args.rng_mode == 'per_shape':
                    reseed(rng, SEED_DATA + g)
                elif handoff is not None:
                    handoff.begin(g, [rng])
                    if handoff.must_publish(g):
                        handoff.publish_after(g, [rng], lambda k: skip_shape(k, rng))
                if fault is not None and fault[0] == rank:
                    if fault[1] == own_seen[0]:
                        raise RuntimeError('injected fault (--fault %s) at shape %d' % (args.fault, g))
                    own_seen[0] += 1
                st0 = state_digest(rng) if store is not None else None
                outs.append(complete_shape(engine, model, pts, rng, args.res, args.chunk, ev if timed else None) + (cloud_of[g],))
                if store is not None:
                    store.set('p2s/bench/rec/%d' % g, json.dumps({
                        'start': st0, 'end': state_digest(rng), 'queries': int(outs[-1][0].shape[0]),
                        'sha256': hashlib.sha256(outs[-1][0].numpy().tobytes()).hexdigest()}))
                if timed:
                    for k, v in model.counters().items():        # per pipeline call (reset at its start)
                        acc[k] = acc.get(k, 0) + v
                if handoff is not None:
                    handoff.done(g)
        return outs

    acc = {}
    run_block(0, args.warmup * world, False)
    barrier()
    clocks = ClockSampler(torch.cuda.current_device()) if rank == 0 else None        # sysfs reads on a thread: not in the way
    if clocks is not None:
        clocks.start()
    t0 = time.time()
    mine_timed = run_block(args.warmup * world, n_rounds * world, True)
    n_queries = sum(int(o[0].shape[0]) for o in mine_timed)
    per_shape_q = {shapes[ci][0][:8]: q_of_cloud[ci] for ci in sorted(set(cloud_of[args.warmup * world:]))}
    gathered = 0
    parts = None
    if handoff is not None:
        handoff.finish()        # every rank is through with its shapes -- or the failure record of one of them raises HERE,
                                # not after the process group's time-out inside the gather below
    if sharding.is_initialized():
        # the final variable-length gather of the SDF values to rank 0 (RCCL over xGMI): the path's only exchange
        mine_dev = torch.cat([o[1] for o in mine_timed]) if mine_timed else torch.empty((0,), dtype=torch.float32, device='cuda')
        parts = sharding.gather_variable(mine_dev, dst=0)
        if rank == 0:
            gathered = sum(int(p_.shape[0]) for p_ in parts)
    barrier()
    dt = time.time() - t0
    device_clock = clocks.stop() if clocks is not None else None

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
        if sharding.is_initialized() and gathered != total_queries:        # P2S_DIST_FORCE: the group at world size 1
            raise SystemExit('gather returned %d SDF values, the rank produced %d' % (gathered, total_queries))

    if rank == 0:
        value = total_queries / dt
        launches = max(acc.get('launches_chain', 0), 1)
        chain_ms = acc.get('ms_chain_stn', 0.0) + acc.get('ms_chain_main', 0.0) + acc.get('ms_chain_qstn', 0.0)
        avg_launch_ms = chain_ms / launches
        flop_per_launch = flop['chain'] * n_queries / launches
        # split precision executes 3 (two pieces) / 6 (three pieces) 16-bit MFMA passes per algorithmic product: the
        # roofline of those modes is priced in executed FLOP against the dense 16-bit peak
        passes = MFMA_PASSES[bf16]
        achieved = passes * flop_per_launch / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
        peak = PEAK_16BIT_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS     # dense MFMA peak of the compute dtype
        # HBM bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE),
        # committed under profiles/; they cannot be collected from inside this process
        traffic, traffic_src = None, None
        if not bf16 and mname == 'p2s_max':
            for rnd in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
                try:
                    with open(os.path.join(REPO, 'profiles', rnd, 'pmc_summary.json')) as f:
                        ck = json.load(f)['chain_kernel']
                    traffic = ck['hbm_traffic_bytes_per_launch'] * (2.0 * n_queries / launches) / ck['queries_per_launch']
                    traffic_src = ('profiles/%s/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, scaled '
                                   'to this launch size)' % rnd)
                    break
                except Exception:
                    continue
        stage = {k: acc[k] for k in sorted(acc) if k.startswith('ms_')}
        stage['ms_cloud'] = float(sum(ev.get('ms_cloud', [])))          # upload + index build (events around engine.Cloud)
        stage['ms_grid'] = float(sum(ev.get('ms_grid', [])))            # a1 incl. its host sync (the handle is fresh: no cached grid)
        stage['ms_d2h'] = float(sum(ev.get('ms_d2h', [])))
        shapes_per_hour = world * args.steps / dt * 3600.0
        out = {
            'metric': 'SDF queries/sec/GPU (%s, %d^3 grid%s)' % (mname, args.res, ENCODER_TEXT[bf16]),
            'value': value, 'unit': 'queries/s',
            'n_gpus': world if not share else 1, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / max(args.steps, 1) * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': DTYPE[bf16],
            'data': 'abc_minimal clouds of the reference repository (committed fixtures); seeded random-init weights' if not args.points else 'synthetic',
            'config': {'workload': 'BASELINE.json configs[%d]: %s, grid_res=%d, eps=3, kNN patch=300 / global sub=1000, %s; '
                                   % ({128: 1, 512: 4}.get(args.res, flop['configs']), mname, args.res,
                                      'fp32' if not bf16 else DTYPE[bf16] + ' encoder + fp32 decoder') +
                                   '%s, one COMPLETE shape per rank per step (host cloud -> upload -> '
                                   'device index build -> query grid -> inference -> SDF in host memory; fresh handle, '
                                   'nothing cached); seeded random-init weights (Famous / ABC test sets and pretrained weights '
                                   'not available offline)' % workload,
                       'queries_per_shape': per_shape_q,
                       'parallelism': 'shape-sharded x%d' % world + (' (REHEARSAL: all ranks share one GPU, gloo)' if share else ''),
                       'rng_mode': args.rng_mode, 'stream_mode': stream_mode,
                       'assignment': 'sharding.assign_shapes: LPT over query counts',
                       'collective': (dist.get_backend() + ' (world %d): one all_gather of sizes + one padded gather at the '
                                      'end of the timed region' % dist.get_world_size()) if sharding.is_initialized() else None,
                       'stream_wait_s_rank0': None if handoff is None else handoff.waited_s,
                       'shapes_per_hour': shapes_per_hour,
                       'queries_per_s_per_gpu': value / world},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': 'p2s_chain_bf16_kernel' if bf16 else 'p2s_chain_kernel', 'launches': int(launches), 'avg_launch_ms': avg_launch_ms,
                         'algorithmic_flop_per_launch': flop_per_launch, 'mfma_passes_per_product': passes,
                         'whole_step_frac': value / world * flop['total'] * passes / 1e12 / peak,
                         'hbm_algorithmic_GBps': value / world * BYTES_PER_QUERY / 1e9,
                         'device_clock_rank0': device_clock},
            'stage_ms_rank0': stage,
        }

        def bail(what, check):
            out['self_check'] = check
            print(json.dumps(out), flush=True)
            raise SystemExit('bench.py self-check FAILED %s: %s' % (what, json.dumps(check)))

        # ---- after the timed region -----------------------------------------------------------------------------
        tol = 0.25 if bf16 == 1 else 1e-4      # north_star: SDF within 1e-4 fp32 of the reference (plain bf16: reported only)
        check = {}
        # (1) the dataset as ONE stream from a fresh start, every query against the reference's goldens.  p2s_vanilla's
        # three-cloud run at 256^3 has no full golden (7.5 h of reference CPU); it has the DATA-PATH golden instead
        # (oracle/make_golden_datapath.py: the unmodified reference's dataset iterated over all 1,378,242 queries -- every
        # weighted-choice draw, every patch, the generator state -- and its network on every 8th query of every shape)
        chk_shapes, stride, dp = shapes, 1, None
        if golden_file is not None and not os.path.isfile(golden_file) and dataset == 'abc3':
            key = 'ref_datapath_%s_abc3_grid%d' % (mname, args.res)
            gf = os.path.join(GOLDEN, key + '.npz')
            if os.path.isfile(gf):
                golden_file = gf
                with open(os.path.join(GOLDEN, 'meta_sizes.json')) as f:
                    dp_meta = json.load(f)[key]
                dp = (np.load(gf), dp_meta)
                stride = int(dp_meta['stride'])
                chk_shapes = [(n, pts, dp[0]['sdf_sub_%d' % i]) for i, (n, pts, _) in enumerate(shapes)]
        reseed(rng)
        sdfs = [complete_shape(engine, model, pts, rng, args.res, args.chunk)[0].numpy() for _, pts, _ in chk_shapes]
        if any(ref is not None for _, _, ref in chk_shapes):
            rec, ok = golden_check(engine, parity, model, w, cfg, chk_shapes, args.res, sdfs, tol, bf16, stride=stride)
            rec['file'] = os.path.relpath(golden_file, REPO)
            if stride > 1:
                rec['every_nth_query'] = stride
            check['vs_reference_golden'] = rec
            if not ok:
                bail('against the reference golden', check)
            if dp is not None:
                drec, dok = datapath_check(engine, shapes, args.res, dp[0], dp[1])
                rec['datapath'] = drec
                if not dok:
                    bail('data path (sub-sample ids / patches / generator state) against the reference', check)
        elif golden_file is not None:
            check['vs_reference_golden'] = {'file': os.path.relpath(golden_file, REPO), 'missing': True}
        # (1b) N > 1: rank 0 replays the WHOLE dataset stream from a fresh start -- every shape's draws consumed by the
        # NULL-ids skip, one after the other -- and compares, per shape, the generator state at its first draw and after
        # its last one with what the shape's owner left in the store (start: the hand-off delivered the right state; end:
        # the inference consumed what the skip consumes).  The LAST timed shape of every other rank is also inferred again
        # from the replayed state and must be bit-identical (sha256, and the gathered values themselves).
        if store is not None and parts is not None and world > 1:
            import hashlib
            lo = args.warmup * world
            last_of = {}
            for g in range(lo, n_rounds * world):
                last_of[owner[g]] = g
            hrec = {'mode': stream_mode, 'shapes': n_rounds * world, 'state_mismatches': [], 'sdf_replayed': []}
            reseed(rng)
            for g in range(n_rounds * world):
                if args.rng_mode == 'per_shape':
                    reseed(rng, SEED_DATA + g)
                theirs = json.loads(store.get('p2s/bench/rec/%d' % g).decode())
                if state_digest(rng) != theirs['start']:
                    hrec['state_mismatches'].append({'shape': g, 'owner': owner[g], 'at': 'start'})
                if owner[g] != 0 and last_of.get(owner[g]) == g:
                    again = complete_shape(engine, model, shapes[cloud_of[g]][1], rng, args.res, args.chunk)[0].numpy()
                    before = sum(q_of_cloud[cloud_of[k]] for k in range(lo, g) if owner[k] == owner[g])
                    got = parts[owner[g]][before:before + again.shape[0]].cpu().numpy()
                    same = bool(got.shape == again.shape and np.array_equal(got, again)
                                and hashlib.sha256(again.tobytes()).hexdigest() == theirs['sha256'])
                    hrec['sdf_replayed'].append({'shape': g, 'owner': owner[g], 'queries': int(again.shape[0]), 'bit_identical': same})
                else:
                    skip_shape(g, rng)
                if state_digest(rng) != theirs['end']:
                    hrec['state_mismatches'].append({'shape': g, 'owner': owner[g], 'at': 'end'})
            hrec['bit_identical_to_single_stream'] = bool(not hrec['state_mismatches'] and hrec['sdf_replayed']
                                                          and all(r['bit_identical'] for r in hrec['sdf_replayed']))
            check['stream_handoff'] = hrec
            if not hrec['bit_identical_to_single_stream']:
                bail('stream replay: %s' % json.dumps(hrec['state_mismatches'][:4] + [r for r in hrec['sdf_replayed'] if not r['bit_identical']][:4]), check)
        # (2) the r02 measurement beside the headline: cloud handles + query grids resident, SDF left on the device
        if world == 1 and mname == 'p2s_max':
            resident = [engine.Cloud(pts) for _, pts, _ in shapes]
            reseed(rng)
            for c in resident:
                c.query_grid(args.res, EPSILON)
            engine.infer_shape(model, resident[0], rng, args.res, EPSILON, chunk=args.chunk, want_queries=False)
            torch.cuda.synchronize()
            t1 = time.time()
            nres = 0
            for c in resident:
                s_, _ = engine.infer_shape(model, c, rng, args.res, EPSILON, chunk=args.chunk, want_queries=False)
                nres += int(s_.shape[0])
            torch.cuda.synchronize()
            dres = time.time() - t1
            out['cloud_resident'] = {'value': nres / dres, 'unit': 'queries/s', 'shapes': len(resident), 'seconds': dres,
                                     'note': 'cloud handle + query grid cached across steps, SDF left in HBM (what BENCH_r02 measured)'}
            for c in resident:
                c.close()
        # (3) secondary passes: configs[3]'s model and the fast exact encoders, complete shapes, full-grid checks
        if (world == 1 and not args.no_secondary and not bf16 and mname == 'p2s_max' and args.points == 0
                and args.res == GRID_RES):
            out['secondary'] = secondary_block(args, engine, parity, synth, sharding, shapes, value, check, bail)
        # (4) the CPU baseline (rank 0, also at N > 1: shapes/hour against the host-CPU baseline), and the device against it
        if args.cpu_seconds > 0:
            c0 = engine.Cloud(shapes[0][1])
            q = c0.query_grid(args.res, EPSILON).cpu().numpy()
            c0.close()
            base, sdf_cpu, what = cpu_baseline(w, cfg, mname, None if args.points else shapes[0][0], shapes[0][1], q,
                                               args.cpu_seconds, args.res)
            out['cpu_baseline'] = base
            n = sdf_cpu.shape[0]
            if what == 'grid32':           # the reference evaluated the 32^3 grid of the first shape: the device does the same
                reseed(rng)
                dev = complete_shape(engine, model, shapes[0][1], rng, 32, args.chunk)[0].numpy()
            elif len(chk_shapes) == len(shapes):
                dev = sdfs[0][:n]
            else:
                reseed(rng)
                c0 = engine.Cloud(shapes[0][1])
                dev = engine.infer_shape(model, c0, rng, args.res, EPSILON, q_end=n, want_queries=False)[0].cpu().numpy()
                c0.close()
            check['vs_cpu_' + base['kind']] = {'queries': int(n), 'max_abs_dsdf': float(np.abs(sdf_cpu - dev).max()),
                                               'sign_flips': int((np.sign(sdf_cpu) != np.sign(dev)).sum())}
            q_per_shape = float(np.mean([q_of_cloud[ci] for ci in cloud_of[args.warmup * world:]]))
            base['shapes_per_hour_cpu'] = base['value'] / q_per_shape * 3600.0
            out['config']['shapes_per_hour_vs_cpu'] = shapes_per_hour / base['shapes_per_hour_cpu']
            if check['vs_cpu_' + base['kind']]['max_abs_dsdf'] > tol or (not bf16 and check['vs_cpu_' + base['kind']]['sign_flips']):
                bail('against the CPU %s' % base['kind'], check)
        else:
            out['cpu_baseline'] = None
        out['self_check'] = check
        print(json.dumps(out), flush=True)
    if sharding.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


class ClockSampler:
    """Shader clock and socket power of the device while a timed region runs, read from the amdgpu hwmon files of the PCI device
    torch reports for it (freq1_input = sclk in Hz, power1_input in microwatt) by a thread every ~20 ms.  Explains box-to-box
    spread: under 16-bit MFMA load the sustained clock differs by ~10 % between boxes (profiles/README.md).  Measurement side
    only; ``region()`` returns None where the files do not exist."""

    def __init__(self, device=0, hwmon=None):
        self.freq, self.power, self.cap_w = None, None, None
        try:
            if hwmon is None:
                import glob
                import torch
                pr = torch.cuda.get_device_properties(device)
                dev = '/sys/bus/pci/devices/%04x:%02x:%02x.0' % (int(getattr(pr, 'pci_domain_id', 0)), int(pr.pci_bus_id),
                                                                 int(getattr(pr, 'pci_device_id', 0)))
                cand = sorted(glob.glob(os.path.join(dev, 'hwmon', 'hwmon*')))
                hwmon = cand[0] if cand else None
            if hwmon and os.path.isfile(os.path.join(hwmon, 'freq1_input')):
                self.freq = os.path.join(hwmon, 'freq1_input')
                if os.path.isfile(os.path.join(hwmon, 'power1_input')):
                    self.power = os.path.join(hwmon, 'power1_input')
                if os.path.isfile(os.path.join(hwmon, 'power1_cap')):
                    self.cap_w = self._read(os.path.join(hwmon, 'power1_cap')) * 1e-6
        except Exception:
            self.freq = None
        self._t, self._stop, self._f, self._p = None, False, [], []

    @staticmethod
    def _read(path):
        with open(path) as f:
            return float(f.read().strip())

    def _loop(self):
        while not self._stop:
            try:
                self._f.append(self._read(self.freq))
                if self.power:
                    self._p.append(self._read(self.power))
            except Exception:
                pass
            time.sleep(0.02)

    def start(self):
        if self.freq is None:
            return
        import threading
        self._stop, self._f, self._p = False, [], []
        self._t = threading.Thread(target=self._loop, daemon=True)
        self._t.start()

    def stop(self):
        """mean / min shader clock (MHz) and mean power (W) since start(); None without samples"""
        if self._t is None:
            return None
        self._stop = True
        self._t.join()
        self._t = None
        if not self._f:
            return None
        out = {'sclk_MHz_mean': sum(self._f) / len(self._f) * 1e-6, 'sclk_MHz_min': min(self._f) * 1e-6, 'samples': len(self._f),
               'source': self.freq}
        if self._p:
            out['power_W_mean'] = sum(self._p) / len(self._p) * 1e-6
        if self.cap_w is not None:
            out['power_cap_W'] = self.cap_w
        return out


def secondary_block(args, engine, parity, synth, sharding, shapes, headline_value, check, bail):
    """N = 1: per entry ONE warm-up and ``--secondary-reps`` timed complete shapes of the test shape (a fresh cloud handle per
    shape, ONE generator handle re-seeded before each: nothing is allocated inside a timed shape), the median with all
    values, the stage times of the median shape and what is not encoder time (``non_chain_ms``)"""
    import torch
    sec = {}
    fixture = np.ascontiguousarray(np.load(cloud_path(FIXTURE_SHAPE))[:, :3], dtype=np.float32)
    reps = max(1, int(args.secondary_reps))
    t_inf = {}
    for key, mname, enc in (('p2s_vanilla_fp32', 'p2s_vanilla', 0), ('p2s_max_bf16x3', 'p2s_max', 3),
                            ('p2s_max_fp16x2', 'p2s_max', 4), ('p2s_vanilla_fp16x2', 'p2s_vanilla', 4)):
        w2, cfg2 = synth.make_weights(mname)
        if enc:
            cfg2 = dict(cfg2, encoder_bf16=enc)
        m2 = engine.Model(w2, cfg2)
        m2.set_profiling(True)
        r2 = engine.Rng(SEED_DATA)
        complete_shape(engine, m2, fixture, r2, args.res, args.chunk)       # warm-up: every buffer of the pass exists now
        torch.cuda.synchronize()
        runs = []
        clocks = ClockSampler()
        clocks.start()
        for _ in range(reps):
            reseed(r2)
            ev2 = {}
            torch.cuda.synchronize()
            t1 = time.time()
            s2 = complete_shape(engine, m2, fixture, r2, args.res, args.chunk, ev2)[0].numpy()
            d2 = time.time() - t1
            cnt = m2.counters()
            st = {k: v for k, v in cnt.items() if k.startswith('ms_')}
            st.update({k: float(v[0]) for k, v in ev2.items()})
            runs.append((d2, st, int(cnt['launches_chain']), s2))
        clk = clocks.stop()
        order = sorted(range(reps), key=lambda i: runs[i][0])
        d2, st, launches, s2 = runs[order[reps // 2]]
        chain_ms = st.get('ms_chain_stn', 0.0) + st.get('ms_chain_main', 0.0) + st.get('ms_chain_qstn', 0.0)
        gfile = os.path.join(GOLDEN, 'ref_rec_%s_testset_grid%d.npz' % (mname, args.res))
        srec = {'value': s2.shape[0] / d2, 'unit': 'queries/s', 'ms_per_step': d2 * 1e3, 'queries': int(s2.shape[0]),
                'values_all': [r[3].shape[0] / r[0] for r in runs], 'ms_per_step_all': [r[0] * 1e3 for r in runs],
                'timed_shapes': reps, 'statistic': 'median',
                'model': mname, 'dtype': DTYPE[enc],
                'workload': 'one complete shape (host to host) of the abc_minimal test shape at %d^3' % args.res,
                'chain_ms': chain_ms, 'chain_launches': launches, 'non_chain_ms': d2 * 1e3 - chain_ms, 'stage_ms': st,
                'device_clock': clk}
        t_inf[key] = d2
        if os.path.isfile(gfile):
            ref2 = np.load(gfile)['rec_0']
            worst = {'max_abs_dsdf': 0.0}
            for r in runs:                                   # every timed shape is checked, not only the median one
                rec2, ok2 = golden_check(engine, parity, m2, w2, cfg2, [(FIXTURE_SHAPE, fixture, ref2)], args.res, [r[3]], 1e-4, enc)
                if not ok2 or rec2['max_abs_dsdf'] >= worst['max_abs_dsdf']:
                    worst = rec2
                if not ok2:
                    break
            worst['file'] = os.path.relpath(gfile, REPO)
            srec['vs_reference_golden'] = {k: worst[k] for k in ('file', 'queries', 'max_abs_dsdf', 'max_abs_diff_unmasked',
                                                                 'sign_flips', 'sign_flips_not_ties', 'flipped', 'tie_logit')}
            if not ok2:
                check['secondary_' + key] = srec
                bail('in the secondary pass ' + key, check)
        sec[key] = srec
        if mname == 'p2s_vanilla' and enc == 0:
            # the cost of keeping the dataset stream exact under sharding: skipping this shape's draws (no inference)
            c2 = engine.Cloud(fixture)
            sharding.skip_shape_stream(c2, r2, cfg2, 32, EPSILON, m2.sub_sample_size)
            ts = []
            for _ in range(reps):
                reseed(r2)
                torch.cuda.synchronize()
                t1 = time.time()
                nskip = sharding.skip_shape_stream(c2, r2, cfg2, args.res, EPSILON, m2.sub_sample_size)
                ts.append(time.time() - t1)
            c2.close()
            sec['skip'] = {'t_s': float(np.median(ts)), 't_s_all': ts, 'unit': 's', 'queries': int(nskip),
                           'ms_per_4096_queries': float(np.median(ts)) * 1e3 * 4096.0 / max(nskip, 1), 'model': 'p2s_vanilla',
                           'what': 'sharding.skip_shape_stream of the test shape at %d^3 (NULL-ids path: tables + offsets pass '
                                   'of the weighted choice, no ids, no inference); stream_mode dataset/handoff needs one such '
                                   'skip per shape hand-over' % args.res}
        m2.close()
        r2.close()
    if 'skip' in sec:
        t_s = sec['skip']['t_s']
        for key in ('p2s_vanilla_fp32', 'p2s_vanilla_fp16x2'):
            sec['skip']['t_i_' + key] = t_inf[key]
            sec['skip']['modelled_efficiency_handoff_' + key] = {str(n): handoff_efficiency(t_inf[key], t_s, n) for n in (2, 4, 8)}
    # the same workload measured through boundary B1 (drop-in API, files in / files out), fp32 and fp16 pair; the SDF files
    # it wrote go through the same golden check as the engine's (a flipped sign passes only as a verified tie)
    for enc_name, enc in (('fp32', 0), ('fp16x2', 4)):
        srec, sdfs = dropin_leg(shapes, args.res, enc_name)
        srec['ratio_to_engine'] = srec['value_shape_loop'] / (headline_value if enc == 0 else max(sec['p2s_max_fp16x2']['value'], 1e-9))
        if any(ref is not None for _, _, ref in shapes):
            w3, cfg3 = synth.make_weights('p2s_max')
            if enc:
                cfg3 = dict(cfg3, encoder_bf16=enc)
            m3 = engine.Model(w3, cfg3)
            rec3, ok3 = golden_check(engine, parity, m3, w3, cfg3, shapes, args.res, sdfs, 1e-4, enc)
            m3.close()
            srec['vs_reference_golden'] = {k: rec3[k] for k in ('queries', 'max_abs_dsdf', 'sign_flips', 'sign_flips_not_ties',
                                                                'flipped', 'tie_logit')}
            if not ok3:
                check['secondary_dropin_' + enc_name] = srec
                bail('in the drop-in pass ' + enc_name, check)
        sec['dropin_' + enc_name] = srec
    return sec


if __name__ == '__main__':
    main()
