"""CPU: the numpy oracle's data path (a4-a6 of SURVEY 8: kNN patch, radius, distance-weighted sub-sample from ONE
dataset-wide legacy MT19937 stream) against the DATA-PATH golden the unmodified reference's ``PointcloudPatchDataset``
wrote for the three abc_minimal clouds at grid 32 (oracle/make_golden_datapath.py; 13,550 queries, p2s_vanilla): the
sha256 of the sub-sample ids of every block of 1024 queries, of the kNN patches / radii of the first and last block of
every shape, and the generator after every shape.  (The 256^3 twin of this golden, 1,378,242 queries, is the GPU test
tests/test_gpu_datapath.py and the self-check of ``bench.py --model p2s_vanilla``.)"""
import hashlib
import json
import os

import numpy as np

from conftest import missing_golden

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FIX = os.path.join(GOLDEN, 'abc_minimal')


def test_oracle_data_path_matches_the_reference_dataset_grid32():
    from oracle import p2s_oracle as O
    key = 'ref_datapath_p2s_vanilla_abc3_grid32'
    path = os.path.join(GOLDEN, key + '.npz')
    if not os.path.isfile(path):
        missing_golden(key, cpu_test=True)
    g = np.load(path)
    with open(os.path.join(GOLDEN, 'meta_sizes.json')) as f:
        meta = json.load(f)[key]
    blk = meta['block']
    rng = O.LegacyMT19937(meta['seed'])
    for si, ms in enumerate(meta['shapes']):
        pts = np.load(os.path.join(FIX, '04_pts', ms['name'] + '.xyz.npy'))
        q = O.query_grid(pts, meta['grid'], 3)[0]
        assert q.shape[0] == ms['queries'] and hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest() == ms['query_sha256']
        ids = np.empty((q.shape[0], 1000), np.int32)
        for j in range(q.shape[0]):
            ids[j] = O.subsample_ids(rng, pts, q[j], 1000, uniform=False)
        nb = (q.shape[0] + blk - 1) // blk
        for b in range(nb):
            assert hashlib.sha256(ids[b * blk:(b + 1) * blk].tobytes()).digest() == bytes(g['ids_sha_%d' % si][b]), (si, b)
        for b in (0, nb - 1):
            qq = q[b * blk:(b + 1) * blk]
            knn = O.knn_ids(pts, qq, 300)
            rad, ps = O.patch_radius_and_ps(pts, knn, qq)
            assert hashlib.sha256(np.ascontiguousarray(ps, np.float32).tobytes()).digest() == bytes(g['patch_sha_%d' % si][b]), (si, b)
            assert hashlib.sha256(np.ascontiguousarray(rad, np.float32).tobytes()).digest() == bytes(g['radius_sha_%d' % si][b]), (si, b)
        # the generator after the shape (the oracle twists as lazily as numpy does: same array, same position)
        assert np.array_equal(rng.mt, g['state_key_%d' % si]) and rng.pos == int(g['state_pos_%d' % si]), si
