"""Drop-in ``source.base.evaluation``: the two functions ``full_eval.py`` calls.

* ``eval_predictions``  (reference source/base/evaluation.py:84-127, full_eval.py:37-41): per-shape MSE / means /
  variances of the GT-query pass against ``05_query_dist``, same CSV -- host-side report arithmetic on 2000 values
  per shape, the device is not involved.
* ``mesh_comparison``   (reference :307-392, full_eval.py:66-75): Hausdorff and Chamfer distances between the
  reconstructed and the ground-truth meshes; surface sampling and nearest-neighbour searches run on the MI355X
  (points2surf_amd/metrics.py).  The reference needs trimesh for this and samples with an unseeded generator.

Everything else of the reference module (confusion-matrix helpers, ``visualize_patch``, ...) is re-exported from the
reference checkout when one is on ``sys.path``.
"""
import importlib.util
import os
import sys

import numpy as np

import source.base as _pkg

_HERE = os.path.dirname(os.path.abspath(__file__))


def _load_reference():
    for p in _pkg.__path__:
        cand = os.path.join(p, 'evaluation.py')
        if os.path.abspath(p) != _HERE and os.path.isfile(cand):
            spec = importlib.util.spec_from_file_location('source.base._reference_evaluation', cand)
            mod = importlib.util.module_from_spec(spec)
            try:
                spec.loader.exec_module(mod)
            except ImportError:
                return None
            return mod
    return None


_ref = _load_reference()
if _ref is not None:
    globals().update({k: v for k, v in vars(_ref).items() if not k.startswith('__')})


def _report_lines(rows, keys):
    """the 'csv' layout of print_list_of_dicts (reference :129-180): right-aligned 10-character columns, names cut to 10
    characters with '_' shown as ' ', numbers with 5 decimals, rows sorted, header first"""
    lines = []
    for d in rows:
        line = ''
        for key in keys:
            cell = d[key][:10].replace('_', ' ') if isinstance(d[key], str) else '{0:.5f}'.format(d[key])
            line += cell.rjust(max(10, len(key))) + ','
        lines.append(line)
    lines.sort()
    lines.insert(0, ''.join(key.replace('_', ' ').rjust(10) + ',' for key in keys))
    return lines


def _rank0_only():
    """under torchrun every rank executes the driver script; reports are written once, by rank 0 (the others wait)"""
    from points2surf_amd import sharding
    world, rank, _ = sharding.dist_env()
    return world, rank


def eval_predictions(pred_path, gt_path, report_file=None, unsigned=False):
    world, rank = _rank0_only()
    if rank != 0:
        return
    files = [f for f in os.listdir(pred_path) if os.path.isfile(os.path.join(pred_path, f)) and f[-4:] == '.npy']
    results = []
    for f in files:
        mat_gt = np.load(os.path.join(gt_path, f[:-8] + '.ply.npy'))
        mat_rec = np.load(os.path.join(pred_path, f))
        if unsigned:
            mat_gt, mat_rec = np.abs(mat_gt), np.abs(mat_rec)
        nz = ((mat_rec != 0.0) + (mat_gt != 0.0)) > 0
        l2 = mat_rec - mat_gt
        mean_gt, mean_rec = mat_gt.mean(), mat_rec.mean()
        results.append({'file': f, 'mse': (l2 * l2)[nz].mean(), 'mean_gt': mean_gt, 'mean_pred': mean_rec,
                        'var_gt': (mat_gt * mat_gt).mean() - mean_gt * mean_gt,
                        'var_pred': (mat_rec * mat_rec).mean() - mean_rec * mean_rec})
    print('compare_prediction: {} vs {}\n'.format(gt_path, pred_path))
    if not results:
        return
    lines = _report_lines(results, ['file', 'mse', 'mean_gt', 'mean_pred', 'var_gt', 'var_pred'])
    for line in lines:
        print(line)
    if report_file is not None:
        if os.path.dirname(report_file):
            os.makedirs(os.path.dirname(report_file), exist_ok=True)
        with open(report_file, 'w') as fh:
            for line in lines:
                fh.write(line + '\n')


def mesh_comparison(new_meshes_dir_abs, ref_meshes_dir_abs, num_processes, report_name, samples_per_model=10000,
                    dataset_file_abs=None):
    from points2surf_amd import metrics, sharding
    world, rank = _rank0_only()
    if rank != 0:
        sharding.barrier()
        return None
    try:
        return _mesh_comparison_rank0(metrics, new_meshes_dir_abs, ref_meshes_dir_abs, num_processes, report_name,
                                      samples_per_model, dataset_file_abs)
    finally:
        if world > 1:
            sharding.barrier()


def _mesh_comparison_rank0(metrics, new_meshes_dir_abs, ref_meshes_dir_abs, num_processes, report_name, samples_per_model,
                           dataset_file_abs):
    return metrics.mesh_comparison(new_meshes_dir_abs, ref_meshes_dir_abs, num_processes, report_name,
                                   samples_per_model=samples_per_model, dataset_file_abs=dataset_file_abs,
                                   seed=int(os.environ.get('P2S_METRIC_SEED', '0')))
