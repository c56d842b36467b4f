"""TEST INFRASTRUCTURE ONLY -- data-path golden of a whole data set at full size, from the UNMODIFIED reference.

Run in the build container only (needs /root/reference; ~100 min of CPU for p2s_vanilla / abc3 / 256):

    python -m oracle.make_golden_datapath p2s_vanilla abc3 256 [stride=8]

Why: the full reference run of p2s_vanilla over the three abc_minimal clouds at 256^3 (BASELINE.json configs[3], the
workload ``bench.py --model p2s_vanilla`` times) is 7.5 h of CPU.  What makes that workload different from the per-shape
goldens that exist is the DATA PATH: 1,378,242 ``RandomState.choice(N, 1000, replace=False, p)`` draws from ONE
data-set-wide stream (reference source/data_loader.py:274-277), the start of every draw depending on every redraw
before it.  This job pins exactly that, and the network on a strided subset of it:

  * the reference's own ``PointcloudPatchDataset`` (source/data_loader.py:181-318, made by
    ``points_to_surf_eval.make_dataset``, source/points_to_surf_eval.py:105-123) is iterated in the order of the
    reference's ``SequentialPointcloudPatchSampler`` with ``--workers 0``: ``dataset[0] ... dataset[len - 1]``
    (``__getitem__``: source/data_loader.py:322-421).
  * per query the script records what the reference hands to the network: the sub-sample ids (the return value of
    ``rng_global_sample.choice``, observed by a delegating proxy around the dataset's RandomState -- the reference
    code is untouched; the ids are cross-checked against ``pts_sub_sample_ms`` of the item), ``patch_pts_ps`` [300,3]
    and ``patch_radius_ms``.  Stored: per shape and per block of 1024 queries the sha256 of the int32 ids, of the
    float32 patches, of the float32 radii, and of the generator state (624 key words + position) after the block;
    the full generator state after every shape.
  * every ``stride``-th query of every shape (local index % stride == 0) also goes through the reference's network:
    ``make_regressor`` (source/points_to_surf_eval.py:150-171), batches of 500 built by torch's default collate,
    ``post_process`` (:174-196), magnitude * sign (:263-271) and NaN -> 1 (:205-207).  Stored: raw logits + SDF.

Output: tests/golden/ref_datapath_<model>_<dataset>_grid<res>.npz + an entry in tests/golden/meta_sizes.json.
Weights: seeded synthetic (points2surf_amd/synth.py, seed 1234) like every other golden.
"""
import hashlib
import os
import queue
import shutil
import sys
import tempfile
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import ref_shims  # noqa: E402
from oracle.make_golden import train_namespace, sha, SEED_DATA  # noqa: E402
from oracle import make_golden_sizes as sizes  # noqa: E402
from points2surf_amd import synth  # noqa: E402

BLOCK = 1024


class RecordingRandomState:
    """delegates everything to the dataset's own RandomState and remembers what ``choice`` returned"""

    def __init__(self, rs):
        self._rs = rs
        self.last_choice = None

    def choice(self, *a, **k):
        self.last_choice = self._rs.choice(*a, **k)
        return self.last_choice

    def __getattr__(self, name):
        return getattr(self._rs, name)


def state_digest(rs):
    name, key, pos, has_gauss, _ = rs.get_state()
    assert name == 'MT19937' and has_gauss == 0
    return hashlib.sha256(np.ascontiguousarray(key, dtype=np.uint32).tobytes() + np.int32(pos).tobytes()).digest()


def run(model, dataset_name, res, stride=8, batch=500, limit=None):
    import torch
    threads = int(os.environ.get('P2S_GOLDEN_THREADS', 4))
    torch.set_num_threads(threads)
    ref_shims.install()
    from source import points_to_surf_eval as ref_eval
    from torch.utils.data.dataloader import default_collate

    w, cfg = synth.make_weights(model, seed=1234)
    tmp = tempfile.mkdtemp(prefix='p2s_golden_')
    out = {}
    meta = {'model': model, 'dataset': dataset_name, 'grid': res, 'job': 'datapath', 'torch': torch.__version__,
            'numpy': np.__version__, 'threads': threads, 'batchSize': batch, 'seed': SEED_DATA, 'stride': stride,
            'block': BLOCK}
    try:
        modeldir = os.path.join(tmp, 'models')
        os.makedirs(modeldir)
        model_file = os.path.join(modeldir, model + '_model.pth')
        torch.save(synth.to_torch_state_dict(w), model_file)
        train_opt = train_namespace(cfg, batch=batch)
        indir_root = sizes.dataset_dir(tmp)
        sub = 'small' if dataset_name == sizes.SMALL else ('standin' if dataset_name == sizes.STANDIN else 'abc_minimal')
        opt = ref_eval.parse_arguments([
            '--indir', os.path.join(indir_root, sub), '--outdir', os.path.join(tmp, 'out'),
            '--dataset', dataset_name + '.txt', '--modeldir', modeldir, '--models', model,
            '--query_grid_resolution', str(res), '--epsilon', '3', '--certainty_threshold', '13', '--sigma', '5',
            '--gpu_idx', '-1', '--workers', '0', '--batchSize', str(batch), '--cache_capacity', '5'])
        opt.reconstruction = True
        torch.manual_seed(opt.seed)
        output_ids = ref_eval.get_output_ids(train_opt)
        pred_dim, output_pred_ind = ref_eval.get_output_dimensions(train_opt)
        dataset = ref_eval.make_dataset(train_opt=train_opt, eval_opt=opt)
        rec = RecordingRandomState(dataset.rng_global_sample)
        dataset.rng_global_sample = rec
        net = ref_eval.make_regressor(train_opt=train_opt, pred_dim=pred_dim, model_filename=model_file,
                                      device=torch.device('cpu'))
        oid_ism = output_pred_ind[output_ids['ism'][0]]
        oid_iss = output_pred_ind[output_ids['iss'][0]]

        # ---- the network on the strided subset, in a worker thread (ATen releases the GIL) ----
        jobs = queue.Queue(maxsize=8)
        results = {}
        failure = []

        def worker():
            try:
                while True:
                    job = jobs.get()
                    if job is None:
                        return
                    key, items = job
                    batch_data = default_collate(items)
                    with torch.no_grad():
                        pred = net(batch_data)
                    logits = pred.clone().numpy()
                    ref_eval.post_process(pred, train_opt, output_ids, output_pred_ind, batch_data['patch_radius_ms'], False)
                    sdf = (pred[:, oid_ism:oid_ism + 1].squeeze(1) * pred[:, oid_iss:oid_iss + 1].squeeze(1)).numpy()
                    sdf[np.isnan(sdf)] = 1.0
                    results[key] = (logits.astype(np.float32), sdf.astype(np.float32))
            except BaseException as e:     # noqa: B902
                failure.append(e)
                raise

        th = threading.Thread(target=worker, daemon=True)
        th.start()

        names = sizes.shapes_of(dataset_name)
        counts = list(dataset.shape_patch_count)
        assert len(names) == len(counts)
        t0 = time.time()
        index = 0
        meta['shapes'] = []
        for s, (name, nq) in enumerate(zip(names, counts)):
            if limit:
                nq_run = min(nq, limit)
            else:
                nq_run = nq
            shape = dataset.shape_cache.get(s)
            cloud = np.asarray(shape.pts)
            h_ids, h_patch, h_rad, h_state = [], [], [], []
            ids_blk = np.empty((BLOCK, train_opt.sub_sample_size), np.int32)
            patch_blk = np.empty((BLOCK, train_opt.points_per_patch, 3), np.float32)
            rad_blk = np.empty((BLOCK,), np.float32)
            pending, n_batches = [], 0
            for j in range(nq_run):
                item = dataset[index + j]
                ids = rec.last_choice
                b = j % BLOCK
                ids_blk[b] = ids
                patch_blk[b] = item['patch_pts_ps'].numpy()
                rad_blk[b] = item['patch_radius_ms'].numpy()
                if j % 4096 == 0:       # the recorded ids are what the item was gathered with
                    assert np.array_equal(cloud[ids], item['pts_sub_sample_ms'].numpy())
                if j % stride == 0:
                    pending.append(item)
                    if len(pending) == batch:
                        jobs.put(((s, n_batches), pending))
                        pending, n_batches = [], n_batches + 1
                if b == BLOCK - 1 or j == nq_run - 1:
                    n = b + 1
                    h_ids.append(hashlib.sha256(ids_blk[:n].tobytes()).digest())
                    h_patch.append(hashlib.sha256(patch_blk[:n].tobytes()).digest())
                    h_rad.append(hashlib.sha256(rad_blk[:n].tobytes()).digest())
                    h_state.append(state_digest(rec._rs))
                    if failure:
                        raise failure[0]
                    if (j // BLOCK) % 32 == 0:
                        el = time.time() - t0
                        print('shape %d query %d / %d   %.0f s   %.1f queries/s' % (s, j + 1, nq_run, el, (index + j + 1) / el),
                              flush=True)
            if pending:
                jobs.put(((s, n_batches), pending))
                n_batches += 1
            index += nq      # the stream continues where THIS run stopped (limit: debugging only)
            _, key, pos, _, _ = rec._rs.get_state()
            out['ids_sha_%d' % s] = np.frombuffer(b''.join(h_ids), np.uint8).reshape(-1, 32)
            out['patch_sha_%d' % s] = np.frombuffer(b''.join(h_patch), np.uint8).reshape(-1, 32)
            out['radius_sha_%d' % s] = np.frombuffer(b''.join(h_rad), np.uint8).reshape(-1, 32)
            out['state_sha_%d' % s] = np.frombuffer(b''.join(h_state), np.uint8).reshape(-1, 32)
            out['state_key_%d' % s] = np.asarray(key, np.uint32)
            out['state_pos_%d' % s] = np.int32(pos)
            meta['shapes'].append({'name': name, 'queries': int(nq), 'queries_run': int(nq_run), 'n_points': int(cloud.shape[0]),
                                   'query_sha256': sha(shape.imp_surf_query_point_ms), 'batches': n_batches})
        meta['datapath_seconds'] = time.time() - t0
        jobs.put(None)
        th.join()
        if failure:
            raise failure[0]
        meta['reference_seconds'] = time.time() - t0
        for s, sh in enumerate(meta['shapes']):
            out['logits_sub_%d' % s] = np.concatenate([results[(s, k)][0] for k in range(sh['batches'])])
            out['sdf_sub_%d' % s] = np.concatenate([results[(s, k)][1] for k in range(sh['batches'])])
            assert out['sdf_sub_%d' % s].shape[0] == (sh['queries_run'] + stride - 1) // stride
            sh['pos_frac_sub'] = float((out['sdf_sub_%d' % s] > 0).mean())
        meta['queries_total'] = int(sum(sh['queries_run'] for sh in meta['shapes']))
        meta['queries_network'] = int(sum(out['sdf_sub_%d' % s].shape[0] for s in range(len(names))))
        meta['reference_queries_per_s'] = meta['queries_total'] / meta['datapath_seconds']
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    key = 'ref_datapath_%s_%s_grid%d' % (model, dataset_name, res) + ('_limit%d' % limit if limit else '')
    np.savez_compressed(os.path.join(sizes.GOLDEN, key + '.npz'), **out)
    sizes.update_meta(key, meta)
    print(key, meta, flush=True)


if __name__ == '__main__':
    run(sys.argv[1], sys.argv[2], int(sys.argv[3]), stride=int(sys.argv[4]) if len(sys.argv) > 4 else 8,
        limit=int(sys.argv[5]) if len(sys.argv) > 5 else None)
