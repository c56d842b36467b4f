"""Throughput over the stand-in data sets of SURVEY 8d configs 3-5 (Famous / ABC / Thingi10k are not downloadable offline):
N clouds = the three abc_minimal clouds under seeded random rotations, re-normalised to the unit cube
(points2surf_amd/synth.py:standin_cloud), every one a COMPLETE shape -- upload, device index build, query grid, inference,
optionally sign propagation + iso-surface, download -- with one dataset-wide sub-sample stream.

    python tools/standin_bench.py [--shapes 22] [--res 256] [--model p2s_max] [--encoder 0|3|4] [--mesh]
    python tools/standin_bench.py --dropin [--shapes 22] [--encoder 0|4] [--workers 7]

--dropin: the same data set THROUGH THE BOUNDARY (bench.py's drop-in leg): the clouds as .npy files, the drop-in's
``points_to_surf_eval(opt)`` (all result files written) then ``implicit_surface_to_mesh_directory`` -- the timed region
of the reference's full_eval.py:44-64.
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
    ap.add_argument('--shapes', type=int, default=22)
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--model', default='p2s_max')
    ap.add_argument('--encoder', type=int, default=0, help='cfg encoder_bf16: 0 fp32, 3 bf16x3, 4 fp16 pair')
    ap.add_argument('--mesh', action='store_true', help='also run sign propagation + iso-surface per shape')
    ap.add_argument('--dropin', action='store_true', help='through the drop-in API (files in / files out) instead of the engine')
    args = ap.parse_args()
    import torch
    from points2surf_amd import engine, synth
    base_dir = os.path.join(REPO, 'tests', 'golden', 'abc_minimal', '04_pts')
    bases = [np.load(os.path.join(base_dir, f)) for f in sorted(os.listdir(base_dir)) if f.endswith('.xyz.npy')]
    clouds = [synth.standin_cloud(bases[i % 3], i) for i in range(args.shapes)]
    if args.dropin:
        import importlib.util
        from points2surf_amd import parity
        spec = importlib.util.spec_from_file_location('p2s_bench', os.path.join(REPO, 'bench.py'))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        bench.GRID_RES = args.res
        shapes = [('standin_%03d' % i, c, None) for i, c in enumerate(clouds)]
        rec = bench.dropin_leg(shapes, args.res, {0: 'fp32', 3: 'bf16x3', 4: 'fp16x2'}[args.encoder], None, parity, model=args.model)
        rec.update(model=args.model, res=args.res, dataset='%d stand-in clouds (synth.standin_cloud, seeds 0..%d)' % (args.shapes, args.shapes - 1))
        print(json.dumps(rec))
        return
    w, cfg = synth.make_weights(args.model)
    model = engine.Model(w, dict(cfg, encoder_bf16=args.encoder))
    rng = engine.Rng(40938661)

    def shape(pts):
        cloud = engine.Cloud(pts)
        sdf, q = engine.infer_shape(model, cloud, rng, args.res, 3, want_queries=args.mesh)
        nv = nf = 0
        if args.mesh:
            vol, _ = engine.sdf_volume(q, sdf, args.res, 5, 13.0)
            v, f, _ = engine.marching_cubes(vol)
            nv, nf = int(v.shape[0]), int(f.shape[0])
            v.cpu(), f.cpu()
        out = sdf.cpu()
        cloud.close()
        return int(out.shape[0]), nv, nf

    shape(clouds[0])                                           # warm-up (allocations)
    shape(clouds[1 % len(clouds)])
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.time()
    res = [shape(c) for c in clouds]
    torch.cuda.synchronize()
    dt = time.time() - t0
    free1 = torch.cuda.mem_get_info()[0]
    nq = sum(r[0] for r in res)
    print(json.dumps({'model': args.model, 'encoder_bf16': args.encoder, 'res': args.res, 'shapes': args.shapes,
                      'mesh_stage': bool(args.mesh), 'queries': nq, 'seconds': dt, 'queries_per_s': nq / dt,
                      'shapes_per_hour': args.shapes / dt * 3600.0,
                      'queries_per_shape_min_max': [min(r[0] for r in res), max(r[0] for r in res)],
                      'vertices_faces_total': [sum(r[1] for r in res), sum(r[2] for r in res)],
                      'device_memory_growth_MB_over_the_run': (free0 - free1) / 1e6}))


if __name__ == '__main__':
    main()
