"""Fixed-radius models on the abc_minimal test shape: stand-alone cost of the patch stages (no encoders running) and
whole-shape throughput at 128^3.    python tools/ball_bench.py [--no-models | --only-models]"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from points2surf_amd import engine, synth   # noqa: E402

pts = np.load(os.path.join(REPO, 'tests', 'golden', 'abc_minimal', '04_pts',
                           '00994122_57d9d4755722f9d2d7436f0a_trimesh_000.xyz.npy')).astype(np.float32)


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e3


cloud = engine.Cloud(pts)
q = cloud.query_grid(128, 3)
n = int(q.shape[0])
for radius in (() if '--only-models' in sys.argv else (0.05, 0.1, 0.2)):
    rng = engine.Rng(1)
    c = engine.ball_count(cloud, q, radius)
    t_count = timed(lambda: engine.ball_count(cloud, q, radius))
    t_skip = timed(lambda: engine.ball_skip(cloud, rng, q, radius, 300))
    t_full = timed(lambda: engine.ball_patch(cloud, rng, q, radius, 300, want_ids=False))
    print('r=%.2f  %d queries, mean ball %.0f, %.0f %% above 300: count %.1f ms, count+chain %.1f ms, all %.1f ms (%.2f us/query)'
          % (radius, n, float(c.float().mean()), 100 * float((c > 300).float().mean()), t_count, t_skip, t_full, t_full / n * 1e3), flush=True)
if '--no-models' not in sys.argv:
    for model in ('p2s_small_radius', 'p2s_medium_radius', 'p2s_large_radius'):
        w, cfg = synth.make_weights(model)
        m = engine.Model(w, cfg)
        r1, r2 = engine.Rng(1), engine.Rng(1)
        engine.infer_shape(m, cloud, r1, 64, 3, rng_patch=r2)
        torch.cuda.synchronize()
        t0 = time.time()
        sdf, _ = engine.infer_shape(m, cloud, r1, 128, 3, rng_patch=r2)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(model, int(sdf.shape[0]), 'queries', '%.1f ms' % (dt * 1e3), '%.0f q/s' % (sdf.shape[0] / dt), flush=True)
        m.close()
