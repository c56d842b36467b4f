"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the mesh metrics.

Reference: source/base/evaluation.py:222-305 -- ``trimesh.sample.sample_surface_even`` (trimesh >= 3.5,
requirements.txt:13; ABSENT from this image: restated from the published implementation, **parity with trimesh itself
is unpinned**), ``scipy.spatial.cKDTree.query`` / ``scipy.spatial.distance.directed_hausdorff`` (installed, used as
they are).  The reference draws its samples from numpy's unseeded global generator; here the generator is passed in.
"""
import numpy as np
import scipy.spatial as spatial


def sample_surface(verts, faces, count, rng):
    """trimesh.sample.sample_surface: area-weighted faces (searchsorted on the cumulative areas), folded barycentric
    lengths.  Draw order: ``count`` face picks, then ``(count, 2, 1)`` lengths."""
    v = np.asarray(verts, dtype=np.float64)
    tri = v[np.asarray(faces)]
    area = np.sqrt((np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]) ** 2).sum(axis=1)) / 2.0
    cum = np.cumsum(area)
    pick = rng.random_sample(count) * cum[-1]
    idx = np.minimum(np.searchsorted(cum, pick), len(cum) - 1)
    origin = tri[idx, 0]
    vec = tri[idx, 1:] - origin[:, None, :]
    lengths = rng.random_sample((count, 2, 1))
    fold = lengths.sum(axis=1).reshape(-1) > 1.0
    lengths[fold] -= 1.0
    lengths = np.abs(lengths)
    return ((vec * lengths).sum(axis=1) + origin).astype(np.float32), float(cum[-1]), idx


def remove_close(points, radius):
    """trimesh.points.remove_close: of every pair within ``radius`` drop the point with the higher pair count (the
    first one on ties)"""
    p = np.asarray(points, dtype=np.float64)
    pairs = spatial.cKDTree(p).query_pairs(radius, output_type='ndarray')
    mask = np.ones(len(p), dtype=bool)
    if len(pairs):
        count = np.bincount(pairs.ravel(), minlength=len(p))
        column = count[pairs].argmax(axis=1)
        highest = pairs.ravel()[column + 2 * np.arange(len(column))]
        mask[highest] = False
    return points[mask], mask


def sample_surface_even(verts, faces, count, rng):
    cand, area, _ = sample_surface(verts, faces, count * 3, rng)
    pts, _ = remove_close(cand, np.sqrt(area / (3 * count)))
    return pts[:count]


def mesh_distances(new_samples, ref_samples):
    """(hausdorff new->ref, ref->new, symmetric, chamfer) as reference :249-254 and :301-304"""
    a, b = np.asarray(new_samples, np.float64), np.asarray(ref_samples, np.float64)
    h_nr = spatial.distance.directed_hausdorff(a, b)[0]
    h_rn = spatial.distance.directed_hausdorff(b, a)[0]
    d_rn = spatial.cKDTree(a, 100).query(b, 1)[0]
    d_nr = spatial.cKDTree(b, 100).query(a, 1)[0]
    return h_nr, h_rn, max(h_nr, h_rn), float(np.sum(d_rn) + np.sum(d_nr))
