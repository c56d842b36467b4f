"""Drop-in ``source.points_to_surf_eval`` (boundary level B1).

``parse_arguments`` accepts exactly the reference's flags (reference source/points_to_surf_eval.py:16-65)
and ``points_to_surf_eval(eval_opt)`` has the same side effects -- files under ``<outdir>/rec/...`` --
so that the reference's ``full_eval.py`` (:17-75) runs unchanged.  Underneath, the whole per-query
path runs on the MI355X engine: the query grid, the kNN patches, the global sub-sample (numpy-legacy
MT19937 stream reproduced on the device, one stream over all shapes in dataset order = the
reference's ``--workers 0`` semantics), the encoders and the decoder.

Deliberate differences (documented in INTEGRATION.md):
  * ``--workers`` / ``--cache_capacity`` / ``--batchSize`` do not influence results (the reference's
    results depend on the worker count through duplicated RNG streams);
  * ``--gpu_idx < 0`` raises: the reference's CPU branch does not run as written either
    (:167,362 call .cuda() unconditionally) and this engine has no CPU fallback;
  * the debug visualisations ``<out>/vis/*.ply`` and ``rec/query_pts_ms_vis/*.ply`` (sdf.visualize_query_points,
    source/sdf.py:269-285) are written by a dependency-free PLY writer (points2surf_amd/ply.py; trimesh is absent,
    byte-identity with its exporter is unpinned);
  * the GT-query pass (``reconstruction=False``: query points from ``05_query_pts``, one random rotation per query
    from the dataset's first RandomState, source/data_loader.py:365-393) runs on the device too (p2s_infer_queries);
    ``full_eval.py`` calls it first whenever ``<indir>/05_query_dist`` exists (:31-33);
  * with torchrun (WORLD_SIZE > 1; ``python -m points2surf_amd.dropin.run`` starts one rank per visible GPU by
    itself) shapes are sharded over ranks (one process per GPU, LPT by query count).  By default the results stay
    identical to the single-process run (dataset-wide stream): the owner of a shape hands the generator state to the
    owner of the next one (sharding.StreamHandoff); ``P2S_RNG_MODE=per_shape`` instead seeds shape i with
    ``seed + i`` (no cross-shape dependency, scales freely, but differs from the reference from the second
    shape on -- a declared deviation).
"""
import argparse
import concurrent.futures
import contextlib
import os
import random
import time

import numpy as np
import torch

from points2surf_amd import engine as _engine
from points2surf_amd import sharding as _sharding
from points2surf_amd.model_spec import strip_module_prefix


_HANDOFF_SEQ = 0
last_run_stats = {}          # timings of the most recent points_to_surf_eval call (bench.py's drop-in leg reads them)


def parse_arguments(args=None):
    parser = argparse.ArgumentParser()

    parser.add_argument('--indir', type=str, default='datasets/abc_minimal', help='input folder (meshes)')
    parser.add_argument('--outdir', type=str, default='results',
                        help='output folder (estimated point cloud properties)')
    parser.add_argument('--dataset', nargs='+', type=str, default=['testset.txt'], help='shape set file name')
    parser.add_argument('--reconstruction', type=bool, default=False, help='do reconstruction instead of evaluation')
    parser.add_argument('--query_grid_resolution', type=int, default=None,
                        help='resolution of sampled volume used for reconstruction')
    parser.add_argument('--epsilon', type=int, default=None, help='neighborhood size for reconstruction')
    parser.add_argument('--certainty_threshold', type=float, default=None, help='')
    parser.add_argument('--sigma', type=int, default=None, help='')
    parser.add_argument('--up_sampling_factor', type=int, default=10, help='unused by the reconstruction path')
    parser.add_argument('--modeldir', type=str, default='models', help='model folder')
    parser.add_argument('--models', type=str, default='p2s_vanilla',
                        help='names of trained models, can evaluate multiple models')
    parser.add_argument('--modelpostfix', type=str, default='_model.pth', help='model file postfix')
    parser.add_argument('--parampostfix', type=str, default='_params.pth', help='parameter file postfix')
    parser.add_argument('--gpu_idx', type=int, default=0, help='GPU index (the engine has no CPU path)')
    parser.add_argument('--sparse_patches', type=int, default=False, help='unused by the reconstruction path')
    parser.add_argument('--sampling', type=str, default='full',
                        help='sampling strategy, any of [full, sequential_shapes_random_patches]')
    parser.add_argument('--patches_per_shape', type=int, default=1000, help='only for random-patch sampling')
    parser.add_argument('--query_points_per_patch', type=int, default=1, help='number of query points per patch')
    parser.add_argument('--sub_sample_size', type=int, default=500, help='overridden by the training parameters')
    parser.add_argument('--seed', type=int, default=40938661, help='manual seed')
    parser.add_argument('--batchSize', type=int, default=0, help='accepted and ignored (the engine chooses its chunk size)')
    parser.add_argument('--workers', type=int, default=0, help='ignored by the device data path')
    parser.add_argument('--cache_capacity', type=int, default=100, help='ignored by the device data path')

    opt = parser.parse_args(args=args)
    if len(opt.dataset) == 1:
        opt.dataset = opt.dataset[0]
    return opt


def get_output_dimensions(train_opt):
    """reference :81-103: the engine implements outputs = magnitude + sign (pred_dim 2) and outputs = imp_surf (pred_dim 1)"""
    pred_dim = 0
    for o in train_opt.outputs:
        if o in ('imp_surf', 'imp_surf_magnitude', 'imp_surf_sign'):
            pred_dim += 1
        elif o in ('p_index', 'patch_pts_ids'):
            pass
        else:
            raise ValueError('Unknown output: %s' % o)
    return pred_dim


def _load_train_opt(param_filename):
    train_opt = torch.load(param_filename, weights_only=False)
    if not hasattr(train_opt, 'single_transformer'):
        train_opt.single_transformer = 0
    if not hasattr(train_opt, 'shared_transformer'):
        train_opt.shared_transformer = False
    return train_opt


def _engine_cfg(train_opt, pred_dim):
    outputs = list(train_opt.outputs)
    pred_cols = [o for o in outputs if o in ('imp_surf', 'imp_surf_magnitude', 'imp_surf_sign')]
    if pred_cols not in (['imp_surf_magnitude', 'imp_surf_sign'], ['imp_surf']):
        # the reference maps prediction columns by the ORDER of train_opt.outputs (output_pred_ind, :81-103); the
        # decoder tail of the engine is column 0 = magnitude, column 1 = sign -- or the single signed-distance logit
        # of the regression model (experiments/train_p2s_regression.sh)
        raise ValueError('the HIP engine supports outputs imp_surf_magnitude, imp_surf_sign (in this order) or imp_surf '
                         '(got %s)' % outputs)
    return dict(
        # > 0: experiments/train_p2s_{small,medium,large}_radius.sh (points2surf_amd/csrc/p2s_ball.hip)
        patch_radius=max(float(getattr(train_opt, 'patch_radius', 0.0) or 0.0), 0.0),
        net_size=getattr(train_opt, 'net_size', 1024), points_per_patch=train_opt.points_per_patch,
        sub_sample_size=train_opt.sub_sample_size, output_dim=pred_dim,
        use_point_stn=bool(train_opt.use_point_stn), use_feat_stn=bool(train_opt.use_feat_stn),
        sym_op=train_opt.sym_op, single_transformer=bool(train_opt.single_transformer),
        shared_transformer=bool(train_opt.shared_transformer),
        uniform_subsample=bool(getattr(train_opt, 'uniform_subsample', 0)),
        fixed_subsample=bool(getattr(train_opt, 'fixed_subsample', 0)),
        # opt-in reduced precision (BASELINE configs[3]): per-point encoder layers on bf16 MFMA, everything else fp32
        encoder_bf16={'fp32': 0, 'bf16': 1, 'bf16x2': 2, 'bf16x3': 3, 'fp16x2': 4}[os.environ.get('P2S_ENCODER', 'fp32')])


def _load_points(indir, shape_name):
    base = os.path.join(indir, '04_pts', shape_name + '.xyz')
    if os.path.isfile(base + '.npy'):
        pts = np.load(base + '.npy')
    elif os.path.isfile(base):
        pts = np.loadtxt(base).astype(np.float32)
    else:
        raise FileNotFoundError(base + '[.npy]')
    if pts.shape[1] > 3:
        pts = pts[:, 0:3]
    if pts.dtype != np.float32:
        print('Warning: pts_np must be converted to float32: {}'.format(base))
        pts = pts.astype(np.float32)
    return np.ascontiguousarray(pts)


def _infer_one_shape(model, cloud, rng_dev, res, eps, chunk, rng_patch=None, want_logits=False):
    """the reference's batch loop for one shape (:358-404).  Both sub-sample modes run on the device:
    randint (p2s_max) and the distance-weighted choice without replacement (p2s_vanilla); fixed-radius models draw
    their patch choice from ``rng_patch``, the data set's first generator."""
    return _engine.infer_shape(model, cloud, rng_dev, res, eps, chunk=chunk, rng_patch=rng_patch, want_logits=want_logits)


def _tie_report(path, shape_name, encoder_bf16, logits, sdf_np, q_np):
    """P2S_TIE_REPORT: the queries of a shape whose sign logit lies within the noise of the reference's sign decision
    ``logit >= 0`` (reference source/sdf_nn.py:16-21) -- the only queries whose sign can differ from a reference run, each
    worth one voxel of the volume.  One JSON line per query, appended."""
    import json
    from points2surf_amd import parity
    sign = logits[:, logits.shape[1] - 1]
    idx = torch.nonzero(sign.abs() < parity.tie_logit(encoder_bf16)).reshape(-1).cpu().numpy()
    vals = sign.cpu().numpy()
    with open(path, 'a') as f:
        for j in idx:
            f.write(json.dumps({'shape': shape_name, 'query': int(j), 'query_point_ms': [float(x) for x in q_np[j]],
                                'sign_logit': float(vals[j]), 'sdf': float(sdf_np[j]),
                                'tie_logit': parity.tie_logit(encoder_bf16)}) + '\n')
    return int(idx.size)


def _visualize_query_points(query_pts_ms, query_dist_ms, file_out):
    """sdf.visualize_query_points (reference source/sdf.py:269-285): red = negative, green = positive distance,
    brightness = |d| / max|d|; written as a coloured point-cloud PLY by the native host writer (p2s_write_query_vis_ply)"""
    from points2surf_amd import writers
    writers.query_vis_ply(file_out, query_pts_ms, query_dist_ms)


def _save_shape(model_out_dir, shape_name, sdf_np, q_np, reconstruction=True):
    """files of save_evaluation (+ save_reconstruction_data in reconstruction mode), reference :199-222, :263-282.  The
    text file is np.savetxt's bytes from the native host writer (p2s_write_txt_f32: the GIL is released, the writer
    threads run beside the inference loop)"""
    from points2surf_amd import writers
    os.makedirs(os.path.join(model_out_dir, 'eval'), exist_ok=True)
    np.save(os.path.join(model_out_dir, 'eval', shape_name + '.xyz.npy'), sdf_np)
    writers.savetxt_f32(os.path.join(model_out_dir, 'eval', shape_name + '.xyz.txt'), sdf_np)
    _visualize_query_points(q_np, sdf_np, os.path.join(model_out_dir, 'vis', shape_name + '.ply'))
    if not reconstruction:
        return
    os.makedirs(os.path.join(model_out_dir, 'query_pts_ms'), exist_ok=True)
    np.save(os.path.join(model_out_dir, 'query_pts_ms', shape_name + '.xyz.npy'), q_np)
    os.makedirs(os.path.join(model_out_dir, 'dist_ms'), exist_ok=True)
    np.save(os.path.join(model_out_dir, 'dist_ms', shape_name + '.xyz.npy'), sdf_np)
    _visualize_query_points(q_np, sdf_np, os.path.join(model_out_dir, 'query_pts_ms_vis', shape_name + '.ply'))


def _save_sampled_shape(model_out_dir, shape_name, sdf_np, q_sel, q_all, picked, reconstruction):
    """the files of save_evaluation for ``--sampling sequential_shapes_random_patches`` (reference :263-294): the values of
    the SAMPLED queries in evaluation order, ``<shape>.idx`` with their indices, and (reconstruction) ALL query points
    beside them, as the reference writes them.  The reference hands all N query points with the M sampled distances to
    sdf.visualize_query_points -- arrays of different length; the visualisations here show the sampled points."""
    from points2surf_amd import writers
    os.makedirs(os.path.join(model_out_dir, 'eval'), exist_ok=True)
    np.save(os.path.join(model_out_dir, 'eval', shape_name + '.xyz.npy'), sdf_np)
    writers.savetxt_f32(os.path.join(model_out_dir, 'eval', shape_name + '.xyz.txt'), sdf_np)
    _visualize_query_points(q_sel, sdf_np, os.path.join(model_out_dir, 'vis', shape_name + '.ply'))
    if reconstruction:
        os.makedirs(os.path.join(model_out_dir, 'query_pts_ms'), exist_ok=True)
        np.save(os.path.join(model_out_dir, 'query_pts_ms', shape_name + '.xyz.npy'), q_all)
        os.makedirs(os.path.join(model_out_dir, 'dist_ms'), exist_ok=True)
        np.save(os.path.join(model_out_dir, 'dist_ms', shape_name + '.xyz.npy'), sdf_np)
        _visualize_query_points(q_sel, sdf_np, os.path.join(model_out_dir, 'query_pts_ms_vis', shape_name + '.ply'))
    np.savetxt(os.path.join(model_out_dir, shape_name + '.idx'), picked, fmt='%d')


def _write_part(model_out_dir, shape_name, rank, sdf_np, q_np):
    """query-range sharding: this rank's ordered piece of a shape (atomic: visible only when complete)"""
    pdir = os.path.join(model_out_dir, '.parts')
    os.makedirs(pdir, exist_ok=True)
    tmp = os.path.join(pdir, '%s.%d.tmp.npz' % (shape_name, rank))
    np.savez(tmp, sdf=sdf_np, q=q_np)
    os.replace(tmp, os.path.join(pdir, '%s.%d.npz' % (shape_name, rank)))


def _try_assemble(model_out_dir, shape_name, world, rank):
    """write the reference's output files of a shape from the ordered pieces of all ranks.  Called after the barrier
    (all pieces on disk); exactly one rank wins the atomic rename of piece 0 and assembles, the pieces are removed.
    Returns True if this rank assembled the shape."""
    pdir = os.path.join(model_out_dir, '.parts')
    parts = [os.path.join(pdir, '%s.%d.npz' % (shape_name, r)) for r in range(world)]
    if not all(os.path.isfile(p) for p in parts):
        return False
    claimed = parts[0] + '.claimed%d' % rank
    try:
        os.rename(parts[0], claimed)
    except OSError:
        return False
    parts[0] = claimed
    loaded = [np.load(p) for p in parts]
    _save_shape(model_out_dir, shape_name, np.concatenate([d['sdf'] for d in loaded]),
                np.concatenate([d['q'] for d in loaded]))
    for p in parts:
        os.remove(p)
    return True


def _load_query_points(indir, shape_name):
    """GT query points of the evaluation pass (reference source/data_loader.py:439-452, :36-49)"""
    q = np.load(os.path.join(indir, '05_query_pts', shape_name + '.ply.npy'))
    if q.dtype != np.float32:
        print('Warning: imp_surf_query_point_ms must be converted to float32')
        q = q.astype(np.float32)
    return np.ascontiguousarray(q)


def points_to_surf_eval(eval_opt):
    models = eval_opt.models.split()
    if eval_opt.seed < 0:
        eval_opt.seed = random.randint(1, 10000)
    if eval_opt.gpu_idx < 0:
        raise RuntimeError('points2surf_amd: --gpu_idx < 0 (CPU) is not available; the HIP engine needs an MI355X')
    reconstruction = bool(eval_opt.reconstruction)
    if eval_opt.sampling not in ('full', 'sequential_shapes_random_patches'):
        raise ValueError('Unknown sampling strategy: %s' % eval_opt.sampling)          # reference :137-138
    random_patches = eval_opt.sampling == 'sequential_shapes_random_patches'
    if reconstruction and (eval_opt.query_grid_resolution is None or eval_opt.epsilon is None):
        raise ValueError('reconstruction needs --query_grid_resolution and --epsilon')

    world, rank, local_rank = _sharding.dist_env()
    # launched by torchrun (tests run "ranks" one after the other without a rendezvous); P2S_DIST_FORCE: the group also
    # at world size 1 (RCCL then executes the barriers of this function on a one-GPU box)
    if (world > 1 or os.environ.get('P2S_DIST_FORCE')) and 'MASTER_PORT' in os.environ:
        _sharding.init_process_group()
    device = _engine.select_device(eval_opt.gpu_idx if world == 1 else _sharding.local_device_index(local_rank))

    for model_name in models:
        print('Random Seed: %d' % eval_opt.seed)
        random.seed(eval_opt.seed)
        torch.manual_seed(eval_opt.seed)
        model_filename = os.path.join(eval_opt.modeldir, model_name + eval_opt.modelpostfix)
        param_filename = os.path.join(eval_opt.modeldir, model_name + eval_opt.parampostfix)
        train_opt = _load_train_opt(param_filename)
        pred_dim = get_output_dimensions(train_opt)
        cfg = _engine_cfg(train_opt, pred_dim)
        state = strip_module_prefix(torch.load(model_filename, map_location='cpu', weights_only=False))
        t_load0 = time.time()
        model = _engine.Model(state, cfg, device=device)
        torch.cuda.synchronize(device)
        t_model = time.time() - t_load0
        # the engine's own chunk size (8192 / 4096 / 2048 queries by model and encoder): results do not depend on it, and the
        # reference's --batchSize (501 in its scripts: a DataLoader batch) as chunk size costs 9 % of the throughput
        # (163 k instead of 179 k queries/s at 256^3)
        chunk = 0

        with open(os.path.join(eval_opt.indir, eval_opt.dataset)) as f:
            shape_names = [x.strip() for x in f.readlines()]
        shape_names = list(filter(None, shape_names))
        model_out_dir = os.path.join(eval_opt.outdir, 'rec' if reconstruction else 'eval')
        os.makedirs(model_out_dir, exist_ok=True)
        print('getting information for {} shapes'.format(len(shape_names)))

        # one RNG stream over all shapes in dataset order (--workers 0 semantics of the reference)
        rng_dev = _engine.Rng(eval_opt.seed, device=device)
        # the dataset's FIRST RandomState (data_loader.py:272): rand(3) per query -> rotation, GT-query pass only
        # -- and, for fixed-radius models, the patch choice of every query in both passes (:336)
        ball = cfg['patch_radius'] > 0.0
        rng_rot = None if (reconstruction and not ball) else _engine.Rng(eval_opt.seed, device=device)
        mine = set(range(len(shape_names)))
        per_shape_rng = os.environ.get('P2S_RNG_MODE', 'dataset') == 'per_shape'
        # P2S_SHARD=queries: every rank takes a contiguous query range of EVERY shape (few, large shapes; 512^3 grids)
        # instead of whole shapes; the RNG stream is advanced past the other ranks' queries, results stay identical
        shard_queries = world > 1 and os.environ.get('P2S_SHARD', 'shapes') == 'queries' and not per_shape_rng \
            and reconstruction
        owner, handoff = None, None
        if world > 1 and reconstruction and not shard_queries:
            if not per_shape_rng and _sharding.stream_handoff_enabled():
                # keys are written once per store: every call of this function (all ranks make the same calls in the
                # same order) gets its own key space.  Created BEFORE the count loop below: a rank that fails there (a
                # missing or corrupt cloud file, out of memory) leaves its failure record, peers raise instead of polling
                global _HANDOFF_SEQ
                _HANDOFF_SEQ += 1
                handoff = _sharding.StreamHandoff('eval%d/%s' % (_HANDOFF_SEQ, model_name), [], rank=rank)
            # ONE policy (sharding.assign_shapes, also bench.py's): LPT over the shapes' query counts -- every rank
            # voxelises every cloud once up front (upload + index + grid: ~1.5 ms per shape) and gets the same list
            counts = []
            with (handoff.guard(-1) if handoff is not None else contextlib.nullcontext()):
                for n in shape_names:
                    c = _engine.Cloud(_load_points(eval_opt.indir, n), device=device)
                    counts.append(c.count_queries(eval_opt.query_grid_resolution, eval_opt.epsilon))
                    c.close()
            parts, owner = _sharding.assign_shapes(counts, world)
            mine = set(parts[rank])
            if handoff is not None:
                handoff.owner = list(owner)
        total_q = 0
        # opt-in list of the queries whose sign is a tie of the reference's own decision
        tie_file = os.environ.get('P2S_TIE_REPORT') if reconstruction else None
        if tie_file and world > 1:
            tie_file += '.rank%d' % rank
        ties_listed = 0
        t0 = time.time()
        # result files are written on background threads while the next shape is on the GPU (np.savetxt alone
        # costs ~0.27 s per 300k values -- a quarter of a shape's inference time)
        writers = concurrent.futures.ThreadPoolExecutor(max_workers=2)
        pending = []
        if shard_queries:
            # stale pieces of an earlier run into the same outdir must not be mistaken for this run's
            if rank == 0:
                import shutil
                shutil.rmtree(os.path.join(model_out_dir, '.parts'), ignore_errors=True)
            _sharding.barrier()
        # --sampling sequential_shapes_random_patches (reference :126-136; source/data_loader.py:88-139): the sampler's
        # OWN RandomState(seed) picks min(patches_per_shape, count) query indices per shape with
        # ``choice(range(start, end), size, replace=False)`` over the data set's global patch indices, and the queries
        # are evaluated in THAT order (the sub-sample stream is consumed in it).  Host logic, numpy's own generator.
        sampler_rng = np.random.RandomState(eval_opt.seed) if random_patches else None
        sampler_start = 0
        for shape_ind, shape_name in enumerate(shape_names):
            if random_patches:
                if rank != 0:
                    continue               # a few hundred queries per shape: rank 0 alone
                if ball and reconstruction:
                    raise ValueError('sampling sequential_shapes_random_patches in reconstruction mode is not built for '
                                     'fixed-radius models')
                cloud = _engine.Cloud(_load_points(eval_opt.indir, shape_name), device=device)
                q_all = cloud.query_grid(eval_opt.query_grid_resolution, eval_opt.epsilon).cpu().numpy() if reconstruction \
                    else _load_query_points(eval_opt.indir, shape_name)
                count = int(q_all.shape[0])
                picked = sampler_rng.choice(range(sampler_start, sampler_start + count),
                                            size=min(int(eval_opt.patches_per_shape), count), replace=False) - sampler_start
                sampler_start += count
                q_sel = np.ascontiguousarray(q_all[picked])
                sdf = _engine.infer_queries(model, cloud, rng_dev, rng_rot, _engine.upload(q_sel, device), chunk=chunk)
                sdf_np = sdf.cpu().numpy()
                total_q += sdf_np.shape[0]
                pending.append(writers.submit(_save_sampled_shape, model_out_dir, shape_name, sdf_np, q_sel, q_all, picked,
                                              reconstruction))
                cloud.close()
                continue
            if not reconstruction:
                # GT-query pass: a few thousand given queries per shape and two dataset-wide streams -- rank 0 runs it
                # alone (sharding it would cost more in stream skipping than the pass itself)
                if rank != 0:
                    continue
                cloud = _engine.Cloud(_load_points(eval_opt.indir, shape_name), device=device)
                q_np = _load_query_points(eval_opt.indir, shape_name)
                sdf = _engine.infer_queries(model, cloud, rng_dev, rng_rot, _engine.upload(q_np, device), chunk=chunk)
                sdf_np = sdf.cpu().numpy()
                total_q += sdf_np.shape[0]
                pending.append(writers.submit(_save_shape, model_out_dir, shape_name, sdf_np, q_np, False))
                cloud.close()
                continue
            if shard_queries:
                cloud = _engine.Cloud(_load_points(eval_opt.indir, shape_name), device=device)
                q_all = cloud.query_grid(eval_opt.query_grid_resolution, eval_opt.epsilon)
                q0, q1 = _sharding.query_range(int(q_all.shape[0]), world, rank)
                _sharding.skip_queries(cloud, rng_dev, cfg, q_all[:q0], model.sub_sample_size, rng_patch=rng_rot)
                sdf, q = _engine.infer_shape(model, cloud, rng_dev, eval_opt.query_grid_resolution, eval_opt.epsilon,
                                             q_begin=q0, q_end=q1, chunk=chunk, rng_patch=rng_rot)
                _sharding.skip_queries(cloud, rng_dev, cfg, q_all[q1:], model.sub_sample_size, rng_patch=rng_rot)
                total_q += int(sdf.shape[0])
                _write_part(model_out_dir, shape_name, rank, sdf.cpu().numpy(), q.cpu().numpy())
                cloud.close()
                continue
            if per_shape_rng:
                if shape_ind not in mine:
                    continue
                rng_dev = _engine.Rng((eval_opt.seed + shape_ind) & 0xffffffff, device=device)
                if ball:
                    rng_rot = _engine.Rng((eval_opt.seed + shape_ind) & 0xffffffff, device=device)
            if shape_ind not in mine and handoff is not None:
                continue                   # exact stream by hand-off: a rank never touches a foreign shape
            if shape_ind not in mine:
                # no process group (ranks run one after the other) / P2S_STREAM_HANDOFF=replicate: keep the
                # dataset-wide stream exact on every rank by consuming this shape's draws without inference
                cloud = _engine.Cloud(_load_points(eval_opt.indir, shape_name), device=device)
                try:
                    _sharding.skip_shape_stream(cloud, rng_dev, cfg, eval_opt.query_grid_resolution,
                                                eval_opt.epsilon, model.sub_sample_size, rng_patch=rng_rot)
                finally:
                    cloud.close()
                continue
            guard = handoff.guard(shape_ind) if handoff is not None else contextlib.nullcontext()
            with guard:                    # an exception here leaves a 'failed' record: waiting ranks raise, not hang
                pts_np = _load_points(eval_opt.indir, shape_name)      # (inside the guard: a missing / corrupt file too)
                if handoff is not None:
                    rngs = [rng_dev] + ([rng_rot] if rng_rot is not None else [])
                    handoff.begin(shape_ind, rngs)         # waits until the owner of the shape before has published
                    if handoff.must_publish(shape_ind):
                        def _advance(k):
                            # on a handle of its own: a cloud smaller than the sub-sample is shuffled in place by its draws
                            c2 = _engine.Cloud(pts_np if k == shape_ind else _load_points(eval_opt.indir, shape_names[k]),
                                               device=device)
                            try:
                                _sharding.skip_shape_stream(c2, rng_dev, cfg, eval_opt.query_grid_resolution,
                                                            eval_opt.epsilon, model.sub_sample_size, rng_patch=rng_rot)
                            finally:
                                c2.close()
                        handoff.publish_after(shape_ind, rngs, _advance)
                cloud = _engine.Cloud(pts_np, device=device)
                res_ = _infer_one_shape(model, cloud, rng_dev, eval_opt.query_grid_resolution, eval_opt.epsilon, chunk,
                                        rng_patch=rng_rot, want_logits=tie_file is not None)
                sdf, q = res_[0], res_[1]
                sdf_np = sdf.cpu().numpy()
                q_np = q.cpu().numpy()
                if tie_file is not None:
                    ties_listed += _tie_report(tie_file, shape_name, cfg['encoder_bf16'], res_[2], sdf_np, q_np)
                total_q += sdf_np.shape[0]
                pending.append(writers.submit(_save_shape, model_out_dir, shape_name, sdf_np, q_np))
                cloud.close()
                if handoff is not None:
                    handoff.done(shape_ind)
        try:
            for f in pending:
                f.result()                 # re-raise writer errors; everything is on disk when we return
            writers.shutdown(wait=True)
        except BaseException as e:
            if handoff is not None:
                handoff.fail(len(shape_names), e)          # a writer error is this rank's failure too: peers raise in finish()
            raise
        dt = time.time() - t0
        if handoff is not None:
            handoff.finish()               # every rank is through, or raises with the record of the one that failed
        _sharding.barrier()                # no-op without a process group
        if shard_queries:
            # every piece is on disk (barrier): rank r takes the shapes i = r mod world first; the claim is atomic,
            # so ranks run one after the other (tests) work too: the last one finds all pieces
            order = sorted(range(len(shape_names)), key=lambda i: ((i - rank) % world, i))
            for shape_ind in order:
                _try_assemble(model_out_dir, shape_names[shape_ind], world, rank)
            _sharding.barrier()
        if rank == 0 and (world == 1 or _sharding.is_initialized()):
            # all writers of all ranks have finished (barrier): every shape must have its files
            want = ['eval'] + (['dist_ms', 'query_pts_ms'] if reconstruction else [])
            missing = [os.path.join(model_out_dir, d, n + '.xyz.npy') for n in shape_names for d in want
                       if not os.path.isfile(os.path.join(model_out_dir, d, n + '.xyz.npy'))]
            if missing:
                raise RuntimeError('points_to_surf_eval: outputs missing after the run: %s' % missing[:4])
        last_run_stats.update(model=model_name, queries=int(total_q), shapes=len(mine), seconds_shapes=dt,
                              seconds_model_create=t_model, rank=rank, world=world,
                              stream_wait_s=None if handoff is None else handoff.waited_s,
                              ties_listed=ties_listed if tie_file else None)
        print('evaluated %d patches of %d shapes in %.2f s (%.0f queries/s on rank %d)'
              % (total_q, len(mine), dt, total_q / max(dt, 1e-9), rank))
        model.close()


if __name__ == '__main__':
    points_to_surf_eval(parse_arguments())
