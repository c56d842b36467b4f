"""p2s_vanilla (QSTN + distance-weighted sub-sample) throughput at 256^3 on a synthetic cloud: whole pipeline and the
sub-sample stage alone.  Secondary measurement (bench.py reports the headline p2s_max metric).

    python tools/vanilla_bench.py [--points 50000] [--steps 1] [--res 256]
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--points', type=int, default=50000)
    ap.add_argument('--steps', type=int, default=1)
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--sub-queries', type=int, default=4096)
    ap.add_argument('--skip-pipeline', action='store_true')
    ap.add_argument('--bf16', nargs='?', const=1, default=0, type=int, help='bf16 encoder + fp32 decoder (BASELINE configs[3]); 2 / 3 = split bf16')
    args = ap.parse_args()
    import torch
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights('p2s_vanilla')
    if args.bf16:
        cfg = dict(cfg, encoder_bf16=int(args.bf16))
    model = engine.Model(w, cfg)
    model.set_profiling(True)
    pts = synth.make_cloud(args.points, seed=1000)
    cloud = engine.Cloud(pts)
    rng = engine.Rng(40938661)
    q = cloud.query_grid(args.res, 3)
    out = {'queries_per_shape': int(q.shape[0]), 'points': args.points}

    # sub-sample stage alone
    nq = min(args.sub_queries, int(q.shape[0]))
    rng.subsample_weighted(cloud, q[:64], 1000, want_pts=False)
    torch.cuda.synchronize()
    t0 = time.time()
    rng.subsample_weighted(cloud, q[:nq], 1000, want_pts=False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    rng.check()
    out['subsample_alone'] = {'queries': nq, 'ms': dt * 1e3, 'us_per_query': dt / nq * 1e6}

    if not args.skip_pipeline:
        engine.infer_shape(model, cloud, rng, args.res, 3, want_queries=False, q_end=8192)
        torch.cuda.synchronize()
        t0 = time.time()
        n = 0
        acc = {}
        for _ in range(args.steps):
            sdf, _ = engine.infer_shape(model, cloud, rng, args.res, 3, want_queries=False)
            n += int(sdf.shape[0])
            for k, v in model.counters().items():
                acc[k] = acc.get(k, 0) + v
        torch.cuda.synchronize()
        dt = time.time() - t0
        rng.check()
        out['pipeline'] = {'queries_per_s': n / dt, 'ms_per_shape': dt / args.steps * 1e3,
                           'stage_ms': {k: acc[k] for k in sorted(acc) if k.startswith('ms_')}}
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
