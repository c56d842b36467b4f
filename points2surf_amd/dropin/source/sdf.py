"""Drop-in ``source.sdf``: the reference module with the whole consumer stage of the SDF samples on
the MI355X -- ``add_samples_to_volume`` + ``propagate_sign`` (reference source/sdf.py:82-178), the iso-surface the
reference obtains from scikit-image's ``marching_cubes_lewiner`` (:211-215), the vertex transform (:223),
``trimesh.repair.fix_inversion`` (:224-225) and the mesh export (:226-227, dependency-free PLY writer).

Everything else -- the query-grid helpers, ``get_query_pts_for_mesh`` (dataset generation), ... -- is the reference's
own code: this module loads the reference's ``source/sdf.py`` from the other ``source`` directory on ``sys.path`` (see
``source/__init__.py``) when one is there and re-exports it.  The functions ``full_eval.py`` calls
(``implicit_surface_to_mesh_directory`` -> ``implicit_surface_to_mesh_file`` -> ``implicit_surface_to_mesh``, and
``visualize_query_points``) are this module's own and need neither the reference checkout nor scikit-image / trimesh.
No CPU fallback: without a GPU they raise.

The iso-surface is scikit-image's ``marching_cubes_lewiner`` itself, re-implemented for the device and pinned to
scikit-image 0.18.3 (identical vertices and oriented triangles on the reference's volumes 32^3 ... 512^3,
tests/test_gpu_mesh.py).  Differences from the reference, all declared in INTEGRATION.md:
``implicit_surface_to_mesh_directory`` runs the shapes serially in this process whatever ``num_processes`` says (a HIP
context cannot be used from the forked ``multiprocessing.Pool`` workers of source/base/utils_mp.py:33-35).
"""
import importlib.util
import os
import sys
import threading
import time

import numpy as np

import source as _pkg

_HERE = os.path.dirname(os.path.abspath(__file__))
_IMPORT_PID = os.getpid()
_DEVICE_STAGE = threading.Lock()


def _load_reference_sdf():
    for p in _pkg.__path__:
        cand = os.path.join(p, 'sdf.py')
        if os.path.abspath(p) != _HERE and os.path.isfile(cand):
            spec = importlib.util.spec_from_file_location('source._reference_sdf', cand)
            mod = importlib.util.module_from_spec(spec)
            sys.modules['source._reference_sdf'] = mod
            try:
                spec.loader.exec_module(mod)           # needs the reference's own imports (trimesh at module level)
            except ImportError:
                del sys.modules['source._reference_sdf']
                return None
            return mod
    return None


_ref = _load_reference_sdf()
if _ref is not None:
    globals().update({k: v for k, v in vars(_ref).items() if not k.startswith('__')})


def _check_process():
    if os.getpid() != _IMPORT_PID:
        # HIP contexts do not survive fork(): the reference's multiprocessing.Pool meshing workers
        # (source/base/utils_mp.py:33-35) must not touch the device
        raise RuntimeError('points2surf_amd: device stage called in a forked worker process; it runs in the parent '
                           '(use source.sdf.implicit_surface_to_mesh_directory of the drop-in)')


class _VolumeWithSamples(np.ndarray):
    """the zero volume of implicit_surface_to_mesh, carrying the samples to the propagation call"""
    _p2s_samples = None


def add_samples_to_volume(vol, pos_ms, val):
    out = np.asarray(vol).view(_VolumeWithSamples)
    out._p2s_samples = (np.ascontiguousarray(pos_ms, dtype=np.float32), np.ascontiguousarray(val, dtype=np.float32))
    return out


def propagate_sign(vol, sigma=5, certainty_threshold=13):
    _check_process()
    samples = getattr(vol, '_p2s_samples', None)
    if samples is None:
        # called on an already populated volume (not the implicit_surface_to_mesh sequence): the samples are the
        # non-zero voxels
        idx = np.nonzero(np.asarray(vol))
        res = vol.shape[0]
        pos = ((np.stack(idx, axis=1) + 0.5) / res * 2.0 - 1.0).astype(np.float32)
        samples = (pos, np.asarray(vol)[idx].astype(np.float32))
    from points2surf_amd import engine
    dev_vol, _ = engine.sdf_volume(samples[0], samples[1], vol.shape[0], sigma, certainty_threshold, clamp=False)
    return dev_vol.cpu().numpy().astype(np.float64)


def visualize_query_points(query_pts_ms, query_dist_ms, file_out_off):
    """reference :269-285: red = negative, green = positive distance, brightness = |d| / max|d|; coloured point-cloud PLY
    (native host writer, p2s_write_query_vis_ply)"""
    from points2surf_amd import writers
    writers.query_vis_ply(file_out_off, query_pts_ms, query_dist_ms)


def implicit_surface_to_mesh(query_dist_ms, query_pts_ms, volume_out_file, mc_out_file, grid_res, sigma,
                             certainty_threshold=26):
    """reference :181-230 with the volume, the iso-surface and the inversion fix on the device"""
    _check_process()
    import torch
    from points2surf_amd import engine, ply, writers
    query_dist_ms = np.asarray(query_dist_ms)
    if query_dist_ms.max() == 0.0 and query_dist_ms.min() == 0.0:
        print('WARNING: implicit surface for {} contains only zeros'.format(volume_out_file))
        return
    # the device stages of one shape at a time (the library's volume / iso-surface scratch is per device); the host work
    # around them runs in parallel when implicit_surface_to_mesh_directory uses several threads
    with _DEVICE_STAGE:
        start = time.time()
        volume, _ = engine.sdf_volume(np.asarray(query_pts_ms), query_dist_ms, grid_res, sigma, certainty_threshold, clamp=True)
        torch.cuda.synchronize()
        print('Sign propagation took: {}'.format(time.time() - start))
        # reference :211-229: mesh only if the volume holds both signs; an iso-surface without a 0-level set is empty, so
        # the extraction itself answers that (no separate min / max pass over the volume)
        start = time.time()
        v, f, _ = engine.marching_cubes(volume, model_space=True, fix_inversion=True)
        torch.cuda.synchronize()
        print('Marching Cubes took: {}'.format(time.time() - start))
        v_np, f_np = v.cpu().numpy(), f.cpu().numpy()
        del volume, v, f

    # green = inside; red = outside (the reference's debug output of the samples, :203-209 -> mesh_io.write_off,
    # source/base/mesh_io.py:75-140): the same bytes from the native host writer (p2s_write_coff_samples) instead of a
    # Python loop with six str() calls per sample (seconds per 256^3 shape)
    writers.coff_samples(volume_out_file, np.asarray(query_pts_ms), query_dist_ms)

    if v_np.shape[0] == 0 and f_np.shape[0] == 0:
        print('Warning: volume for marching cubes contains no 0-level set!')
    else:
        if os.path.dirname(mc_out_file):
            os.makedirs(os.path.dirname(mc_out_file), exist_ok=True)
        # trimesh.Trimesh(vertices=v, faces=f) of the reference (:224) merges coincident vertices before the export
        mv, mf = ply.merge_vertices(v_np, f_np)
        ply.write_ply(mc_out_file, mv, mf)


def implicit_surface_to_mesh_file(query_dist_ms_file, query_pts_ms_file, volume_out_file, mc_out_file, grid_res, sigma,
                                  certainty_threshold):
    implicit_surface_to_mesh(np.load(query_dist_ms_file), np.load(query_pts_ms_file), volume_out_file, mc_out_file,
                             grid_res, sigma, certainty_threshold)


def _call_necessary(files_in, files_out):
    """file_utils.call_necessary (source/base/file_utils.py:194-240): inputs exist and an output is missing or older"""
    if any(not os.path.isfile(f) for f in files_in):
        print('WARNING: Input file are missing: {}'.format([f for f in files_in if not os.path.isfile(f)]))
        return False
    if any(not os.path.isfile(f) or os.path.getsize(f) == 0 for f in files_out):
        return True
    return max(os.path.getmtime(f) for f in files_in) >= min(os.path.getmtime(f) for f in files_out)


def implicit_surface_to_mesh_directory(imp_surf_dist_ms_dir, query_pts_ms_dir, vol_out_dir, mesh_out_dir,
                                       grid_res, sigma, certainty_threshold, num_processes=1):
    """reference :240-266.  The reference forks ``num_processes`` workers (source/base/utils_mp.py:33-35), each running
    149 s of sign propagation per 256^3 shape on a CPU core; here a shape's propagation + iso-surface takes milliseconds
    on the device and what is left is host work -- loading the arrays, the coloured-samples ``.off``, the vertex merge and
    the PLY export -- which ``num_processes`` THREADS of this process share (a HIP context does not survive a fork; the
    native writers and numpy release the GIL)."""
    os.makedirs(vol_out_dir, exist_ok=True)
    os.makedirs(mesh_out_dir, exist_ok=True)
    dist_files = sorted(f for f in os.listdir(imp_surf_dist_ms_dir)
                        if os.path.isfile(os.path.join(imp_surf_dist_ms_dir, f)) and f[-8:] == '.xyz.npy')
    # under torchrun every rank executes the driver script: the shapes are dealt round-robin (same sorted list on
    # every rank), nobody writes a file another rank writes
    from points2surf_amd import sharding
    world, rank, _ = sharding.dist_env()
    calls = []
    for f in dist_files[rank::world] if world > 1 else dist_files:
        f_dist, f_query = os.path.join(imp_surf_dist_ms_dir, f), os.path.join(query_pts_ms_dir, f)
        f_vol, f_mesh = os.path.join(vol_out_dir, f[:-8] + '.off'), os.path.join(mesh_out_dir, f[:-8] + '.ply')
        if _call_necessary([f_dist, f_query], [f_vol, f_mesh]):
            calls.append((f_dist, f_query, f_vol, f_mesh, grid_res, sigma, certainty_threshold))
    n_threads = max(1, min(int(num_processes or 1), len(calls), 16))
    if n_threads <= 1:
        for c in calls:
            implicit_surface_to_mesh_file(*c)
    else:
        import concurrent.futures
        import torch
        dev = torch.cuda.current_device()

        def work(c):
            torch.cuda.set_device(dev)             # the current device is per thread
            implicit_surface_to_mesh_file(*c)
        with concurrent.futures.ThreadPoolExecutor(max_workers=n_threads) as ex:
            for fut in [ex.submit(work, c) for c in calls]:
                fut.result()                       # re-raise a worker's error
    if world > 1:
        sharding.barrier()               # the metrics stage that follows reads every rank's meshes


if _ref is not None:
    # code of the reference module that resolves these names in ITS globals gets the device versions too
    _ref.add_samples_to_volume = add_samples_to_volume
    _ref.propagate_sign = propagate_sign
    _ref.visualize_query_points = visualize_query_points
