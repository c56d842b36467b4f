"""Cost of keeping the dataset-wide RNG stream exact under shape sharding: time to SKIP one shape's draws
(sharding.skip_shape_stream, NULL-ids path) vs time to INFER it, for p2s_max (randint stream) and p2s_vanilla
(distance-weighted choice), on the abc_minimal fixture shape at 256^3.  Feeds the efficiency model in DESIGN.md.

    python tools/skip_bench.py [--res 256] [--models p2s_vanilla] [--skip-only] [--shape 0|1|2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
ABC3 = ['00011084_fddd53ce45f640f3ab922328_trimesh_019', '00016513_3d6966cd42eb44ab8f4224f2_trimesh_053',
        '00994122_57d9d4755722f9d2d7436f0a_trimesh_000']


def efficiency(t_i, t_s, n, mode):
    """modelled weak-scaling efficiency of the exact dataset stream on n GPUs (multi-GPU)"""
    if mode == 'handoff':
        return t_i / max(t_i + t_s, n * t_s) if n > 1 else 1.0
    return 1.0 / (1.0 + (n - 1) * t_s / t_i)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--models', default='p2s_max,p2s_vanilla')
    ap.add_argument('--shape', type=int, default=2, help='index into abc3 (2 = the abc_minimal test shape)')
    ap.add_argument('--skip-only', action='store_true', help='no inference leg (kernel traces of the skip alone)')
    ap.add_argument('--encoder', type=int, default=0, help='cfg.encoder_bf16 of the inference leg (4 = fp16 pair)')
    args = ap.parse_args()
    import torch
    from points2surf_amd import engine, synth, sharding
    pts = np.load(os.path.join(REPO, 'tests', 'golden', 'abc_minimal', '04_pts', ABC3[args.shape] + '.xyz.npy'))
    out = {'res': args.res, 'shape': ABC3[args.shape][:8], 'points': int(pts.shape[0])}
    for name in args.models.split(','):
        w, cfg = synth.make_weights(name)
        if args.encoder:
            cfg = dict(cfg, encoder_bf16=args.encoder)
        model = engine.Model(w, cfg)
        cloud = engine.Cloud(pts)
        rng = engine.Rng(40938661)
        if not args.skip_only:
            engine.infer_shape(model, cloud, rng, args.res, 3, want_queries=False, q_end=8192)     # warm-up
        sharding.skip_shape_stream(cloud, rng, cfg, 32, 3, 1000)
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            t0 = time.time()
            n = sharding.skip_shape_stream(cloud, rng, cfg, args.res, 3, 1000)
            torch.cuda.synchronize()
            ts.append(time.time() - t0)
        t_skip = float(np.median(ts))
        rec = {'queries': int(n), 'skip_ms': t_skip * 1e3, 'skip_ms_all': [t * 1e3 for t in ts],
               'skip_ms_per_4096': t_skip * 1e3 * 4096.0 / max(n, 1)}
        if not args.skip_only:
            t0 = time.time()
            engine.infer_shape(model, cloud, rng, args.res, 3, want_queries=False)
            torch.cuda.synchronize()
            t_inf = time.time() - t0
            rec.update({'infer_ms': t_inf * 1e3, 'skip_over_infer': t_skip / t_inf,
                        'modelled_efficiency_handoff': {str(g): efficiency(t_inf, t_skip, g, 'handoff') for g in (1, 2, 4, 8)},
                        'modelled_efficiency_replicate': {str(g): efficiency(t_inf, t_skip, g, 'replicate') for g in (1, 2, 4, 8)}})
        out[name] = rec
        model.close()
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
