"""Point cloud -> SDF samples -> sign volume -> iso-surface -> PLY on one MI355X, timed per stage (what the
reconstruction half of full_eval.py does for one shape: reference source/points_to_surf_eval.py:358-404 +
source/sdf.py:181-230).    python tools/shape_to_mesh.py [--model p2s_max] [--res 256] [--out /tmp/mesh.ply]"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='p2s_max')
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--out', default='/tmp/p2s_mesh.ply')
    args = ap.parse_args()
    import torch
    from points2surf_amd import engine, synth, ply
    pts = np.load(os.path.join(REPO, 'tests', 'golden', 'abc_minimal', '04_pts',
                               '00994122_57d9d4755722f9d2d7436f0a_trimesh_000.xyz.npy'))
    w, cfg = synth.make_weights(args.model)
    model = engine.Model(w, cfg)
    cloud = engine.Cloud(pts)
    out = {'model': args.model, 'res': args.res, 'points': int(pts.shape[0])}
    for rep in range(2):                       # first pass warms allocations up
        rng = engine.Rng(40938661)
        torch.cuda.synchronize()
        t0 = time.time()
        sdf, q = engine.infer_shape(model, cloud, rng, args.res, 3)
        torch.cuda.synchronize()
        t1 = time.time()
        vol, sweeps = engine.sdf_volume(q, sdf, args.res, 5, 13.0, clamp=True)
        torch.cuda.synchronize()
        t2 = time.time()
        v, f, inverted = engine.marching_cubes(vol, model_space=True, fix_inversion=True)
        torch.cuda.synchronize()
        t3 = time.time()
        vh, fh = v.cpu().numpy(), f.cpu().numpy()
        ply.write_ply(args.out, vh, fh)
        t4 = time.time()
    e = set()
    for a, b in ((0, 1), (1, 2), (2, 0)):
        e.update(map(tuple, np.sort(fh[:, [a, b]], axis=1)))
    out.update({'queries': int(sdf.shape[0]), 'inference_ms': (t1 - t0) * 1e3, 'sign_propagation_ms': (t2 - t1) * 1e3,
                'sweeps': int(sweeps), 'iso_surface_ms': (t3 - t2) * 1e3, 'download_and_ply_ms': (t4 - t3) * 1e3,
                'total_ms': (t4 - t0) * 1e3, 'vertices': int(vh.shape[0]), 'faces': int(fh.shape[0]),
                'euler_characteristic': int(vh.shape[0] - len(e) + fh.shape[0]), 'ply_bytes': os.path.getsize(args.out)})
    print(json.dumps(out))


if __name__ == '__main__':
    main()
