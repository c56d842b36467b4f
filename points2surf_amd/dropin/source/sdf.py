"""Drop-in ``source.sdf``: the reference module with ``add_samples_to_volume`` +
``propagate_sign`` (reference source/sdf.py:82-178) executed on the MI355X.

Everything else -- ``implicit_surface_to_mesh*`` (:181-266), marching cubes via scikit-image, mesh export via
trimesh, the query-grid helpers -- is the reference's own code: this module loads the reference's
``source/sdf.py`` from the other ``source`` directory on ``sys.path`` (see ``source/__init__.py``) and
re-exports it, swapping only the two functions.  ``implicit_surface_to_mesh`` calls them back to back
(:192-198), so ``add_samples_to_volume`` just remembers its arguments on the returned array and
``propagate_sign`` runs scatter + propagation on the device in one go (p2s_sdf_volume, C ABI).
No CPU fallback: without a GPU ``propagate_sign`` raises.
"""
import importlib.util
import os
import sys

import numpy as np

import source as _pkg

_HERE = os.path.dirname(os.path.abspath(__file__))


def _load_reference_sdf():
    for p in _pkg.__path__:
        cand = os.path.join(p, 'sdf.py')
        if os.path.abspath(p) != _HERE and os.path.isfile(cand):
            spec = importlib.util.spec_from_file_location('source._reference_sdf', cand)
            mod = importlib.util.module_from_spec(spec)
            sys.modules['source._reference_sdf'] = mod
            spec.loader.exec_module(mod)
            return mod
    raise ImportError('points2surf_amd drop-in: the reference checkout (its source/sdf.py) must also be on sys.path')


_ref = _load_reference_sdf()
globals().update({k: v for k, v in vars(_ref).items() if not k.startswith('__')})


class _VolumeWithSamples(np.ndarray):
    """the zero volume of implicit_surface_to_mesh, carrying the samples to the propagation call"""
    _p2s_samples = None


def add_samples_to_volume(vol, pos_ms, val):
    out = np.asarray(vol).view(_VolumeWithSamples)
    out._p2s_samples = (np.ascontiguousarray(pos_ms, dtype=np.float32), np.ascontiguousarray(val, dtype=np.float32))
    return out


_IMPORT_PID = os.getpid()


def propagate_sign(vol, sigma=5, certainty_threshold=13):
    if os.getpid() != _IMPORT_PID:
        # HIP contexts do not survive fork(): the reference's multiprocessing.Pool meshing workers
        # (source/base/utils_mp.py:33-35) must not touch the device
        raise RuntimeError('points2surf_amd: propagate_sign called in a forked worker process; the device stage runs '
                           'in the parent (use source.sdf.implicit_surface_to_mesh_directory of the drop-in)')
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


def implicit_surface_to_mesh_directory(imp_surf_dist_ms_dir, query_pts_ms_dir, vol_out_dir, mesh_out_dir,
                                       grid_res, sigma, certainty_threshold, num_processes=1):
    """reference source/sdf.py:240-266 with the per-shape calls made serially in THIS process: the volume stage
    runs on the GPU, and a HIP context cannot be used from the forked ``multiprocessing.Pool`` workers the reference
    starts for ``num_processes > 1`` (full_eval.py passes ``--workers``).  ``num_processes`` is accepted and ignored:
    one shape's propagation takes milliseconds on the device (149 s on a CPU core at 256^3)."""
    from source.base import file_utils
    os.makedirs(vol_out_dir, exist_ok=True)
    os.makedirs(mesh_out_dir, exist_ok=True)
    dist_files = [f for f in os.listdir(imp_surf_dist_ms_dir)
                  if os.path.isfile(os.path.join(imp_surf_dist_ms_dir, f)) and f[-8:] == '.xyz.npy']
    for f in dist_files:
        f_dist, f_query = os.path.join(imp_surf_dist_ms_dir, f), os.path.join(query_pts_ms_dir, f)
        f_vol, f_mesh = os.path.join(vol_out_dir, f[:-8] + '.off'), os.path.join(mesh_out_dir, f[:-8] + '.ply')
        if file_utils.call_necessary([f_dist, f_query], [f_vol, f_mesh]):
            _ref.implicit_surface_to_mesh_file(f_dist, f_query, f_vol, f_mesh, grid_res, sigma, certainty_threshold)


# the reference's implicit_surface_to_mesh resolves both names in ITS module globals
_ref.add_samples_to_volume = add_samples_to_volume
_ref.propagate_sign = propagate_sign
