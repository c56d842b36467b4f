"""Complete-shape throughput of a model on the abc_minimal test shape at 256^3 for several pipeline chunk sizes
(development aid: the default chunk per model / encoder in p2s_pipeline.hip comes from such a sweep).

    python tools/chunk_sweep.py --model p2s_vanilla --encoder 0 --chunks 2048,4096,8192
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
    ap.add_argument('--model', default='p2s_vanilla')
    ap.add_argument('--encoder', type=int, default=0)
    ap.add_argument('--chunks', default='2048,4096,8192')
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--reps', type=int, default=2)
    args = ap.parse_args()
    import torch
    from points2surf_amd import engine, synth
    pts = np.load(os.path.join(REPO, 'tests', 'golden', 'abc_minimal', '04_pts',
                               '00994122_57d9d4755722f9d2d7436f0a_trimesh_000.xyz.npy'))
    w, cfg = synth.make_weights(args.model)
    if args.encoder:
        cfg = dict(cfg, encoder_bf16=args.encoder)
    model = engine.Model(w, cfg)
    cloud = engine.Cloud(pts)
    rng = engine.Rng(40938661)
    out = {'model': args.model, 'encoder': args.encoder}
    engine.infer_shape(model, cloud, rng, args.res, 3, want_queries=False, q_end=16384, chunk=8192)
    for c in [int(x) for x in args.chunks.split(',')]:
        best = 0.0
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.time()
            sdf, _ = engine.infer_shape(model, cloud, rng, args.res, 3, want_queries=False, chunk=c)
            torch.cuda.synchronize()
            best = max(best, sdf.shape[0] / (time.time() - t0))
        out[str(c)] = best
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
