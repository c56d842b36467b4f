"""Mesh metrics on the device: what reference source/base/evaluation.py:222-392 computes with trimesh +
scipy on the host -- even surface samples of the reconstructed and the reference mesh, Hausdorff and Chamfer distances
between the sample sets -- through libp2s_hip.so (p2s_mesh_sample_surface, p2s_points_remove_close,
p2s_nn_distance_stats).  Torch tensors are containers only; no CPU fallback.

The reference samples with numpy's unseeded global generator (trimesh.sample.sample_surface -> np.random.random), so
its numbers change from run to run; here the deviates come from a seeded device twin of ``np.random.RandomState``
(``seed`` argument), in the order trimesh draws them.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from . import engine as _engine
from . import ply as _ply


def _dev(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError('points2surf_amd needs a ROCm GPU (gfx950); no CPU fallback exists')
    return torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)


def sample_surface(verts, faces, count, rng, want_faces=False):
    """trimesh.sample.sample_surface(mesh, count): (points [count,3] float32 device tensor, area)"""
    lib = _lib.load()
    dev = verts.device
    v = verts.to(dev, torch.float32).contiguous()
    f = faces.to(dev, torch.int32).contiguous()
    u = torch.empty((3 * count,), dtype=torch.float64, device=dev)
    pts = torch.empty((count, 3), dtype=torch.float32, device=dev)
    fid = torch.empty((count,), dtype=torch.int32, device=dev) if want_faces else None
    area = ctypes.c_double(0.0)
    with torch.cuda.device(dev):
        s = _engine._stream_ptr(dev)
        _lib.check(lib.p2s_rng_random_sample(rng.handle, 3 * count, _engine._ptr(u), s))
        _lib.check(lib.p2s_mesh_sample_surface(_engine._ptr(v), _engine._ptr(f), int(f.shape[0]), _engine._ptr(u), int(count),
                                               _engine._ptr(pts), _engine._ptr(fid), ctypes.byref(area), dev.index, s))
    return (pts, float(area.value), fid) if want_faces else (pts, float(area.value))


def sample_surface_even(verts, faces, count, rng):
    """trimesh.sample.sample_surface_even(mesh, count) (reference source/base/evaluation.py:235): 3 * count area-weighted
    samples, minus the points with a neighbour within sqrt(area / (3 count)), cut to count.  Returns [<= count, 3]."""
    lib = _lib.load()
    dev = verts.device
    cand, area = sample_surface(verts, faces, 3 * count, rng)
    radius = float(np.sqrt(area / (3 * count)))
    out = torch.empty((count, 3), dtype=torch.float32, device=dev)
    n = ctypes.c_int64(0)
    with torch.cuda.device(dev):
        _lib.check(lib.p2s_points_remove_close(_engine._ptr(cand), int(cand.shape[0]), ctypes.c_double(radius), int(count),
                                               _engine._ptr(out), ctypes.byref(n), dev.index, _engine._stream_ptr(dev)))
    return out[:n.value]


def directed_stats(samples_from, samples_to):
    """(max, sum) of the nearest-neighbour distances from every point of ``samples_from`` to the set ``samples_to``:
    scipy.spatial.distance.directed_hausdorff(from, to)[0] and the Chamfer term np.sum(kdtree_to.query(from)[0])"""
    lib = _lib.load()
    target = _engine.Cloud(samples_to)
    q = samples_from.to(target.device, torch.float32).contiguous()
    mx, sm = ctypes.c_double(0.0), ctypes.c_double(0.0)
    with torch.cuda.device(target.device):
        _lib.check(lib.p2s_nn_distance_stats(target.handle, _engine._ptr(q), int(q.shape[0]), None, ctypes.byref(mx),
                                             ctypes.byref(sm), _engine._stream_ptr(target.device)))
    target.close()
    return float(mx.value), float(sm.value)


def mesh_distances(file_in, file_ref, samples_per_model=10000, seed=0, device=None):
    """(hausdorff new->ref, hausdorff ref->new, hausdorff, chamfer) of two mesh files; (-1, -1, -1, -1) if a mesh is
    missing or empty (reference :244-245, :278-279, :298-299)"""
    dev = _dev(device)
    sets = []
    rng = _engine.Rng(seed, device=dev)
    for path in (file_in, file_ref):
        try:
            v, f = _ply.read_ply(path)
        except Exception:
            v, f = np.zeros((0, 3)), np.zeros((0, 3), np.int64)
        if v.shape[0] == 0 or f.shape[0] == 0:
            return -1.0, -1.0, -1.0, -1.0
        f = np.asarray(f)
        if f.ndim != 2 or f.shape[1] != 3 or f.min() < 0 or f.max() >= v.shape[0] or not np.isfinite(v).all():
            # malformed / truncated mesh: the reference's try/except around trimesh ends in (-1, -1, -1, -1)
            # (source/base/evaluation.py:244-245); never hand out-of-range indices to the device kernels
            return -1.0, -1.0, -1.0, -1.0
        vt = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev)
        ft = torch.from_numpy(np.ascontiguousarray(f, dtype=np.int32)).to(dev)
        s = sample_surface_even(vt, ft, samples_per_model, rng)
        if s.shape[0] == 0:
            return -1.0, -1.0, -1.0, -1.0
        sets.append(s)
    new, ref = sets
    h_nr, s_nr = directed_stats(new, ref)
    h_rn, s_rn = directed_stats(ref, new)
    return h_nr, h_rn, max(h_nr, h_rn), s_nr + s_rn


def mesh_comparison(new_meshes_dir_abs, ref_meshes_dir_abs, num_processes, report_name, samples_per_model=10000,
                    dataset_file_abs=None, seed=0):
    """reference source/base/evaluation.py:307-392: the same pairing rules and the same CSV; the distances come from the
    device.  ``num_processes`` is accepted and ignored (one mesh pair takes milliseconds; HIP contexts do not fork)."""
    if not os.path.isdir(new_meshes_dir_abs):
        print('Warning: dir to check doesn\'t exist'.format(new_meshes_dir_abs))
        return
    new_mesh_files = sorted(f for f in os.listdir(new_meshes_dir_abs) if os.path.isfile(os.path.join(new_meshes_dir_abs, f)))
    ref_mesh_files = sorted(f for f in os.listdir(ref_meshes_dir_abs) if os.path.isfile(os.path.join(ref_meshes_dir_abs, f)))
    if dataset_file_abs is None:
        to_compare = set(ref_mesh_files)
    else:
        if not os.path.isfile(dataset_file_abs):
            raise ValueError('File does not exist: {}'.format(dataset_file_abs))
        with open(dataset_file_abs) as f:
            to_compare = set((line.replace('\n', '') + '.ply').split('.')[0] for line in f.readlines())

    def ref_for(new_mesh_file):
        stem = new_mesh_file.split('.')[0]
        return sorted(set(f for f in ref_mesh_files if f.split('.')[0] == stem))

    results = []
    for new_mesh_file in new_mesh_files:
        if new_mesh_file.split('.')[0] in to_compare:
            match = ref_for(new_mesh_file)
            if match:
                a, b = os.path.join(new_meshes_dir_abs, new_mesh_file), os.path.join(ref_meshes_dir_abs, match[0])
                h_nr, h_rn, h, ch = mesh_distances(a, b, samples_per_model, seed=seed)
                results.append((a, b, str(h_nr), str(h_rn), str(h), str(ch)))
    if len(results) == 0:
        raise ValueError('Results are empty!')
    for new_mesh_file in new_mesh_files:          # no reference but reconstruction
        stem = new_mesh_file.split('.')[0]
        if stem not in to_compare:
            if dataset_file_abs is None:
                match = ref_for(new_mesh_file)
                if match:
                    results.append((os.path.join(new_meshes_dir_abs, new_mesh_file),
                                    os.path.join(ref_meshes_dir_abs, match[0]), str(-2), str(-2), str(-2), str(-2)))
        else:
            to_compare.remove(stem)
    for missing in sorted(to_compare):            # no reconstruction but reference
        results.append((os.path.join(new_meshes_dir_abs, missing), os.path.join(ref_meshes_dir_abs, missing),
                        str(-1), str(-1), str(-1), str(-1)))
    results = sorted(results, key=lambda x: x[0])
    if os.path.dirname(report_name):
        os.makedirs(os.path.dirname(report_name), exist_ok=True)
    csv_lines = ['in mesh,ref mesh,Hausdorff dist new-ref,Hausdorff dist ref-new,Hausdorff dist,'
                 'Chamfer dist(-1: no input; -2: no reference)']
    csv_lines += [','.join(item) for item in results]
    with open(report_name, 'w') as text_file:
        text_file.write('\n'.join(csv_lines))
    return results
