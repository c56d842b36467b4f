"""Cost of keeping the dataset-wide RNG stream exact under shape sharding: time to SKIP one shape's draws
(sharding.skip_shape_stream, NULL-ids path) vs time to INFER it, for p2s_max (randint stream) and p2s_vanilla
(distance-weighted choice), on the abc_minimal fixture shape at 256^3.  Feeds the efficiency model in DESIGN.md 5.

    python tools/skip_bench.py [--res 256]
"""
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
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--reps', type=int, default=3)
    args = ap.parse_args()
    import torch
    from points2surf_amd import engine, synth, sharding
    pts = np.load(os.path.join(REPO, 'tests', 'golden', 'abc_minimal', '04_pts',
                               '00994122_57d9d4755722f9d2d7436f0a_trimesh_000.xyz.npy'))
    out = {'res': args.res}
    for name in ('p2s_max', 'p2s_vanilla'):
        w, cfg = synth.make_weights(name)
        model = engine.Model(w, cfg)
        cloud = engine.Cloud(pts)
        rng = engine.Rng(40938661)
        engine.infer_shape(model, cloud, rng, args.res, 3, want_queries=False, q_end=8192)     # warm-up
        sharding.skip_shape_stream(cloud, rng, cfg, 32, 3, 1000)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.reps):
            n = sharding.skip_shape_stream(cloud, rng, cfg, args.res, 3, 1000)
        torch.cuda.synchronize()
        t_skip = (time.time() - t0) / args.reps
        t0 = time.time()
        sdf, _ = engine.infer_shape(model, cloud, rng, args.res, 3, want_queries=False)
        torch.cuda.synchronize()
        t_inf = time.time() - t0
        out[name] = {'queries': int(n), 'skip_ms': t_skip * 1e3, 'infer_ms': t_inf * 1e3, 'skip_over_infer': t_skip / t_inf,
                     'modelled_efficiency': {str(g): 1.0 / (1.0 + (g - 1) * t_skip / t_inf) for g in (1, 2, 4, 8)}}
        model.close()
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
