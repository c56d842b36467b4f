"""CPU: the native host writers of the C ABI (p2s_hostio.hip; no device is touched) write the bytes the reference's
numpy / Python calls write: np.savetxt (source/points_to_surf_eval.py:210), mesh_io.write_off of the coloured samples
(source/sdf.py:203-209, source/base/mesh_io.py:75-140), and the drop-in's visualisation PLY (points2surf_amd/ply.py)."""
import os

import numpy as np
import pytest

from points2surf_amd import ply, writers


def _values(n, seed):
    rs = np.random.RandomState(seed)
    v = (rs.standard_normal(n) * np.exp(rs.uniform(-12, 3, n))).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, 1e-4, 9.9999e-5, 1e-5, 1.5e-5, 123456789.0, 1e15, 1e16, 3e16, 1e-38, 1e-45,
                        3.4e38, 0.1, 0.5, 0.25, 1 / 3, 16777216.0, 1e7, 99999.99, 100000.0, 0.001, 0.00099999],
                       dtype=np.float32)
    v[:special.size] = special
    v[special.size:2 * special.size] = -special
    return v


def test_savetxt_bytes(tmp_path):
    v = _values(50000, 1)
    a, b = str(tmp_path / 'a.txt'), str(tmp_path / 'b.txt')
    np.savetxt(a, v)
    writers.savetxt_f32(b, v)
    assert open(a, 'rb').read() == open(b, 'rb').read()
    w = v.copy()
    w[7], w[9], w[11] = np.nan, np.inf, -np.inf
    np.savetxt(a, w)
    writers.savetxt_f32(b, w)
    assert open(a, 'rb').read() == open(b, 'rb').read()
    writers.savetxt_f32(b, np.zeros(0, np.float32))
    assert open(b, 'rb').read() == b''


def _reference_coff(path, q, d):
    """source/sdf.py:203-209 + source/base/mesh_io.py:75-140, restated line for line (test infrastructure)"""
    norm = d / np.max(np.abs(d))
    col = np.zeros((norm.shape[0], 3))
    col[norm < 0.0, 0] = np.abs(norm[norm < 0.0]) + 1.0 / 2.0
    col[norm > 0.0, 1] = norm[norm > 0.0] + 1.0 / 2.0
    with open(path, 'w') as fp:
        fp.write('COFF\n')
        fp.write(str(len(q)) + ' ' + str(0) + ' 0\n')
        for vi, v in enumerate(q):
            line = str(v[0]) + ' ' + str(v[1]) + ' ' + str(v[2]) + ' '
            for c in range(3):
                line += str(col[vi][c]) + ' '
            fp.write(line + '\n')


@pytest.mark.parametrize('seed', [2, 3])
def test_coff_samples_bytes(tmp_path, seed):
    n = 30000
    q = _values(3 * n, seed).reshape(n, 3)
    d = (np.random.RandomState(seed + 10).standard_normal(n) * 0.02).astype(np.float32)
    d[::97] = 0.0
    a, b = str(tmp_path / 'a.off'), str(tmp_path / 'b.off')
    with np.errstate(all='ignore'):
        _reference_coff(a, q, d)
    writers.coff_samples(b, q, d)
    assert open(a, 'rb').read() == open(b, 'rb').read()


def test_repr_of_many_float32_values(tmp_path):
    """str(np.float32) for 400k values across all magnitudes (the coordinates column of the COFF file)"""
    n = 133334
    rs = np.random.RandomState(5)
    bits = rs.randint(0, 2 ** 32, size=3 * n, dtype=np.uint64).astype(np.uint32)
    v = bits.view(np.float32)
    v = v[np.isfinite(v)][:3 * (v.size // 3)]
    v = v[:3 * (v.size // 3)].reshape(-1, 3)
    d = np.ones(v.shape[0], np.float32)
    b = str(tmp_path / 'b.off')
    writers.coff_samples(b, v, d)
    lines = open(b).read().split('\n')[2:-1]
    got = [t for ln in lines for t in ln.split(' ')[:3]]
    want = [str(x) for x in v.reshape(-1)]
    assert got == want


def test_query_vis_ply_bytes(tmp_path):
    n = 20000
    rs = np.random.RandomState(7)
    q = rs.uniform(-1, 1, (n, 3)).astype(np.float32)
    d = (rs.standard_normal(n) * 0.03).astype(np.float32)
    d[::50] = 0.0
    # the drop-in's Python writer (sdf.visualize_query_points restated over points2surf_amd/ply.py)
    d_abs = np.abs(d)
    d_norm = d_abs / d_abs.max()
    col = np.zeros((n, 3))
    col[d < 0.0, 0] = 0.5 + 0.5 * d_norm[d < 0.0]
    col[d > 0.0, 1] = 0.5 + 0.5 * d_norm[d > 0.0]
    a, b = str(tmp_path / 'a.ply'), str(tmp_path / 'b.ply')
    ply.write_ply(a, q, vertex_colors=col)
    writers.query_vis_ply(b, q, d)
    assert open(a, 'rb').read() == open(b, 'rb').read()
    v, f = ply.read_ply(b)
    assert np.array_equal(v.astype(np.float32), q) and f.shape[0] == 0


def test_bad_arguments(tmp_path):
    from points2surf_amd import _lib
    with pytest.raises(_lib.P2SError):
        writers.savetxt_f32(str(tmp_path / 'no_such_dir_x' / 'y' / '..' / '..' / 'nope' / 'f.txt') + '/', np.zeros(3, np.float32))
    with pytest.raises(ValueError):
        writers.query_vis_ply(str(tmp_path / 'x.ply'), np.zeros((4, 3), np.float32), np.zeros(5, np.float32))


def test_write_errors_are_reported_with_the_io_error_code(tmp_path):
    """a failed open / write / flush / close is P2S_EIO with errno's text, never a silent truncated file.
    /dev/full accepts the open and fails the flush: the deferred error a plain fclose() in a destructor would swallow"""
    from points2surf_amd import _lib
    vals = np.linspace(-1, 1, 5000).astype(np.float32)
    with pytest.raises(_lib.P2SError) as e:
        writers.savetxt_f32(str(tmp_path), vals)                      # a directory: fopen fails
    assert e.value.code == -6 and 'cannot open' in str(e.value)
    if os.path.exists('/dev/full'):
        with pytest.raises(_lib.P2SError) as e:
            writers.savetxt_f32('/dev/full', vals)
        assert e.value.code == -6 and ('No space' in str(e.value) or 'write to' in str(e.value) or 'flush' in str(e.value))
