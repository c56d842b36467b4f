"""TEST INFRASTRUCTURE ONLY -- restatement of the two trimesh functions the reference's GT-query
evaluation pass calls (reference source/data_loader.py:381-393):

    trimesh.transformations.random_rotation_matrix(rand)      (trimesh >= 3.5, requirements.txt:13)
    trimesh.transformations.transform_points(points, matrix)

trimesh is NOT installed in this image and cannot be fetched (no network), so these follow the
*published* formulas (trimesh/transformations.py is Christoph Gohlke's ``transformations.py``:
``random_quaternion`` -> ``quaternion_matrix``; ``transform_points`` is a homogeneous float64
``np.dot``).  **Parity of the rotation pass is pinned to this restatement, not to trimesh itself.**
Self-checks that do not need trimesh: the matrix is orthonormal with det +1, and the documented
special values (rand = (0,0,0) -> quaternion (0, 0, 1, 0)) hold -- tests/test_oracle_rotation.py.
"""
import math

import numpy as np

_EPS = np.finfo(float).eps * 4.0


def random_quaternion(rand=None):
    """Uniform random unit quaternion from three uniform deviates in [0, 1) (Shoemake)."""
    if rand is None:
        rand = np.random.rand(3)
    else:
        assert len(rand) == 3
    r1 = np.sqrt(1.0 - rand[0])
    r2 = np.sqrt(rand[0])
    pi2 = math.pi * 2.0
    t1 = pi2 * rand[1]
    t2 = pi2 * rand[2]
    return np.array([np.cos(t2) * r2, np.sin(t1) * r1, np.cos(t1) * r1, np.sin(t2) * r2])


def quaternion_matrix(quaternion):
    """Homogeneous 4x4 rotation matrix of a quaternion (w, x, y, z)."""
    q = np.array(quaternion, dtype=np.float64, copy=True)
    n = np.dot(q, q)
    if n < _EPS:
        return np.identity(4)
    q *= math.sqrt(2.0 / n)
    q = np.outer(q, q)
    return np.array([
        [1.0 - q[2, 2] - q[3, 3], q[1, 2] - q[3, 0], q[1, 3] + q[2, 0], 0.0],
        [q[1, 2] + q[3, 0], 1.0 - q[1, 1] - q[3, 3], q[2, 3] - q[1, 0], 0.0],
        [q[1, 3] - q[2, 0], q[2, 3] + q[1, 0], 1.0 - q[1, 1] - q[2, 2], 0.0],
        [0.0, 0.0, 0.0, 1.0]])


def random_rotation_matrix(rand=None):
    return quaternion_matrix(random_quaternion(rand))


def identity_matrix():
    return np.identity(4)


def transform_points(points, matrix, translate=True):
    """float64 homogeneous transform of [n, 3] points (returns float64; the reference casts to float32)."""
    points = np.asanyarray(points, dtype=np.float64)
    if len(points) == 0 or matrix is None:
        return points.copy()
    matrix = np.asanyarray(matrix, dtype=np.float64)
    count, dim = points.shape
    if np.abs(matrix - np.eye(dim + 1)).max() < 1e-8:
        return np.ascontiguousarray(points.copy())
    if translate:
        stack = np.column_stack((points, np.ones(count)))
        return np.dot(matrix, stack.T).T[:, :dim]
    return np.dot(matrix[:dim, :dim], points.T).T
