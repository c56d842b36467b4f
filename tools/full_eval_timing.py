"""The statement sequence of the reference's ``full_eval.py`` (:17-75) through the drop-in, with a timer per stage, on the
committed abc_minimal data (three clouds with GT query points and reference meshes) -- the reference file itself is not on
the GPU box, so its statements are restated here one for one (tests/test_full_eval_reference_file.py drives the real file on
CPU).  Stages: GT-query pass -> eval_predictions -> reconstruction pass -> implicit_surface_to_mesh_directory ->
mesh_comparison.

    python tools/full_eval_timing.py [--res 256] [--model p2s_max] [--encoder fp32|fp16x2] [--workers 7]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'points2surf_amd', 'dropin'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--model', default='p2s_max')
    ap.add_argument('--encoder', default='fp32')
    ap.add_argument('--workers', type=int, default=7)
    ap.add_argument('--cold', action='store_true', help='no warm-up run: the first call of the process (allocations, tables)')
    args = ap.parse_args()
    os.environ['P2S_ENCODER'] = args.encoder
    import torch
    from points2surf_amd import synth
    from source import points_to_surf_eval
    from source.base import evaluation
    from source import sdf
    tmp = tempfile.mkdtemp(prefix='p2s_full_eval_')
    t = {}
    try:
        shutil.copytree(os.path.join(REPO, 'tests', 'golden', 'abc_minimal'), os.path.join(tmp, 'datasets', 'abc_minimal'))
        synth.write_model_files(os.path.join(tmp, 'models'), args.model)
        argv = ['--indir', os.path.join(tmp, 'datasets'), '--outdir', os.path.join(tmp, 'results'), '--dataset',
                'abc_minimal/abc3.txt', '--modeldir', os.path.join(tmp, 'models'), '--models', args.model,
                '--query_grid_resolution', str(args.res), '--epsilon', '3', '--certainty_threshold', '13', '--sigma', '5',
                '--workers', str(args.workers), '--batchSize', '501', '--cache_capacity', '5']

        def full_eval(opt, timers):
            # ---- full_eval.py:17-75 ----
            indir_root = opt.indir
            outdir_root = os.path.join(opt.outdir, opt.models + os.path.splitext(opt.modelpostfix)[0])
            datasets = opt.dataset
            if not isinstance(datasets, list):
                datasets = [datasets]
            for dataset in datasets:
                opt.indir = os.path.join(indir_root, os.path.dirname(dataset))
                opt.outdir = os.path.join(outdir_root, os.path.dirname(dataset))
                opt.dataset = os.path.basename(dataset)
                if os.path.exists(os.path.join(opt.indir, '05_query_dist')):
                    t0 = time.time()
                    opt.reconstruction = False
                    points_to_surf_eval.points_to_surf_eval(opt)
                    timers['gt_query_pass'] = time.time() - t0
                    res_dir_eval = os.path.join(opt.outdir, 'eval')
                    t0 = time.time()
                    evaluation.eval_predictions(os.path.join(res_dir_eval, 'eval'), os.path.join(opt.indir, '05_query_dist'),
                                                os.path.join(res_dir_eval, 'rme_comp_res.csv'), unsigned=False)
                    timers['eval_predictions'] = time.time() - t0
                t0 = time.time()
                opt.reconstruction = True
                points_to_surf_eval.points_to_surf_eval(opt)
                res_dir_rec = os.path.join(opt.outdir, 'rec')
                timers['reconstruction_pass'] = time.time() - t0
                timers['queries'] = points_to_surf_eval.last_run_stats.get('queries')
                t0 = time.time()
                sdf.implicit_surface_to_mesh_directory(
                    os.path.join(res_dir_rec, 'dist_ms'), os.path.join(res_dir_rec, 'query_pts_ms'),
                    os.path.join(res_dir_rec, 'vol'), os.path.join(res_dir_rec, 'mesh'),
                    opt.query_grid_resolution, opt.sigma, opt.certainty_threshold, opt.workers)
                timers['mesh_directory'] = time.time() - t0
                t0 = time.time()
                evaluation.mesh_comparison(
                    new_meshes_dir_abs=os.path.join(res_dir_rec, 'mesh'), ref_meshes_dir_abs=os.path.join(opt.indir, '03_meshes'),
                    num_processes=opt.workers, report_name=os.path.join(res_dir_rec, 'hausdorff_dist_pred_rec.csv'),
                    samples_per_model=10000, dataset_file_abs=os.path.join(opt.indir, opt.dataset))
                timers['mesh_comparison'] = time.time() - t0
                return res_dir_rec

        if not args.cold:
            warm = points_to_surf_eval.parse_arguments(argv)
            warm.query_grid_resolution = 64
            warm.outdir = os.path.join(tmp, 'warm')
            full_eval(warm, {})                                 # warm-up: allocations, generator tables, page cache
            torch.cuda.synchronize()
        t0 = time.time()
        rec = full_eval(points_to_surf_eval.parse_arguments(argv), t)
        t['total'] = time.time() - t0
        with open(os.path.join(rec, 'hausdorff_dist_pred_rec.csv')) as f:
            csv = f.read().strip().split('\n')
        print(json.dumps({'model': args.model, 'encoder': args.encoder, 'res': args.res, 'shapes': 3, 'cold': bool(args.cold), 'seconds': t,
                          'queries_per_s_reconstruction': t['queries'] / t['reconstruction_pass'],
                          'shapes_per_hour_whole_sequence': 3 / t['total'] * 3600.0,
                          'hausdorff_csv_head': csv[:4]}))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
