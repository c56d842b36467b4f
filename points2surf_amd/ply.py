"""Dependency-free binary PLY writers for the files the reference exports through trimesh
(``trimesh.Trimesh(...).export('*.ply')``: reference source/sdf.py:226-227 meshes, :284-285 coloured query points).

trimesh is not installed in this image, so byte-identity with trimesh's exporter is **unpinned**; the layout follows
the PLY 1.0 specification the way trimesh writes it (binary_little_endian, float32 x/y/z, uchar rgba per vertex when
colours are given, ``list uchar int vertex_indices`` faces) and the files load in trimesh / MeshLab / Open3D.
File formatting only -- no part of the SDF computation happens here.
"""
import numpy as np


def _header(n_vertices, n_faces, colors):
    lines = ['ply', 'format binary_little_endian 1.0', 'comment points2surf_amd',
             'element vertex %d' % n_vertices, 'property float x', 'property float y', 'property float z']
    if colors:
        lines += ['property uchar red', 'property uchar green', 'property uchar blue', 'property uchar alpha']
    lines += ['element face %d' % n_faces, 'property list uchar int vertex_indices', 'end_header']
    return ('\n'.join(lines) + '\n').encode('ascii')


def float_colors_to_rgba(colors):
    """float colours in [0, 1] ([n,3] or [n,4]) -> uint8 rgba, opaque"""
    c = np.asarray(colors, dtype=np.float64)
    c = np.where(np.isfinite(c), c, 0.0)
    c = np.clip(np.round(c * 255.0), 0, 255).astype(np.uint8)
    if c.shape[1] == 3:
        c = np.concatenate([c, np.full((c.shape[0], 1), 255, np.uint8)], axis=1)
    return c


def write_ply(path, vertices, faces=None, vertex_colors=None):
    v = np.ascontiguousarray(vertices, dtype='<f4').reshape(-1, 3)
    f = np.zeros((0, 3), '<i4') if faces is None else np.ascontiguousarray(faces, dtype='<i4').reshape(-1, 3)
    with open(path, 'wb') as fh:
        fh.write(_header(v.shape[0], f.shape[0], vertex_colors is not None))
        if vertex_colors is not None:
            rec = np.empty(v.shape[0], dtype=[('p', '<f4', 3), ('c', 'u1', 4)])
            rec['p'] = v
            rec['c'] = float_colors_to_rgba(vertex_colors) if np.asarray(vertex_colors).dtype.kind == 'f' \
                else np.asarray(vertex_colors, np.uint8)
            fh.write(rec.tobytes())
        else:
            fh.write(v.tobytes())
        if f.shape[0]:
            rec = np.empty(f.shape[0], dtype=[('n', 'u1'), ('i', '<i4', 3)])
            rec['n'] = 3
            rec['i'] = f
            fh.write(rec.tobytes())


def read_ply(path):
    """minimal reader for the two layouts above and for ASCII / binary triangle meshes with float x/y/z first
    (the abc_minimal 03_meshes files): returns (vertices [n,3] float64, faces [m,3] int64)"""
    with open(path, 'rb') as fh:
        data = fh.read()
    end = data.index(b'end_header')
    end = data.index(b'\n', end) + 1
    header = data[:end].decode('ascii', 'replace').split('\n')
    fmt = 'ascii'
    elements = []
    for line in header:
        t = line.split()
        if not t:
            continue
        if t[0] == 'format':
            fmt = t[1]
        elif t[0] == 'element':
            elements.append([t[1], int(t[2]), []])
        elif t[0] == 'property':
            elements[-1][2].append(t[1:])
    types = {'char': 'i1', 'uchar': 'u1', 'int8': 'i1', 'uint8': 'u1', 'short': 'i2', 'ushort': 'u2', 'int16': 'i2',
             'uint16': 'u2', 'int': 'i4', 'uint': 'u4', 'int32': 'i4', 'uint32': 'u4', 'float': 'f4', 'float32': 'f4',
             'double': 'f8', 'float64': 'f8'}
    verts, faces = np.zeros((0, 3)), np.zeros((0, 3), np.int64)
    if fmt == 'ascii':
        tokens = data[end:].split()
        pos = 0
        for name, count, props in elements:
            if name == 'vertex':
                k = len(props)
                arr = np.array(tokens[pos:pos + k * count], dtype=np.float64).reshape(count, k)
                pos += k * count
                names = [p[-1] for p in props]
                verts = arr[:, [names.index('x'), names.index('y'), names.index('z')]]
            elif name == 'face':
                out = []
                for _ in range(count):
                    n = int(tokens[pos])
                    idx = [int(x) for x in tokens[pos + 1:pos + 1 + n]]
                    pos += 1 + n
                    for j in range(1, n - 1):
                        out.append((idx[0], idx[j], idx[j + 1]))
                faces = np.array(out, dtype=np.int64).reshape(-1, 3)
            else:
                raise ValueError('read_ply: unsupported ascii element %s' % name)
        return verts, faces
    bo = '<' if fmt == 'binary_little_endian' else '>'
    off = end
    for name, count, props in elements:
        if name == 'vertex':
            dt = np.dtype([(p[-1], bo + types[p[0]]) for p in props])
            arr = np.frombuffer(data, dtype=dt, count=count, offset=off)
            off += dt.itemsize * count
            verts = np.stack([arr['x'], arr['y'], arr['z']], axis=1).astype(np.float64)
        elif name == 'face':
            if len(props) != 1 or props[0][0] != 'list':
                raise ValueError('read_ply: unsupported face layout')
            ct, it = np.dtype(bo + types[props[0][1]]), np.dtype(bo + types[props[0][2]])
            out = []
            uniform = np.dtype([('n', ct), ('i', it, 3)])
            if count and off + uniform.itemsize * count == len(data):
                arr = np.frombuffer(data, dtype=uniform, count=count, offset=off)
                if (arr['n'] == 3).all():
                    faces = arr['i'].astype(np.int64)
                    off += uniform.itemsize * count
                    continue
            for _ in range(count):
                n = int(np.frombuffer(data, ct, 1, off)[0])
                off += ct.itemsize
                idx = np.frombuffer(data, it, n, off)
                off += it.itemsize * n
                for j in range(1, n - 1):
                    out.append((idx[0], idx[j], idx[j + 1]))
            faces = np.array(out, dtype=np.int64).reshape(-1, 3)
        else:
            dt = np.dtype([(p[-1], bo + types[p[0]]) for p in props])
            off += dt.itemsize * count
    return verts, faces


def merge_vertices(vertices, faces, digits=8):
    """What ``trimesh.Trimesh(vertices=v, faces=f)`` (``process=True``, the constructor default the reference uses,
    source/sdf.py:224) does to the mesh before it is exported: ``merge_vertices`` -- referenced vertices whose
    coordinates agree after rounding to ``tol.merge`` = 1e-8 become one vertex, in order of first occurrence; faces are
    re-indexed, none is removed (``validate=False``).  The iso-surface of a volume with exact zeros (and of scikit-image's
    eps-offset vertices next to them) has such coincident vertices.  Restated from trimesh 3.x
    (``trimesh/grouping.py:merge_vertices``, ``trimesh/base.py:process``); trimesh is not installed here: UNPINNED.
    Host-side post-processing of the export, like the PLY writer itself."""
    v = np.asarray(vertices)
    f = np.asarray(faces)
    if len(v) == 0 or len(f) == 0:
        return v, f
    referenced = np.zeros(len(v), dtype=bool)
    referenced[f.reshape(-1)] = True
    rows = np.round(v[referenced].astype(np.float64) * 10 ** digits).astype(np.int64)
    # group equal rows: through a 64-bit hash of the row (np.unique of a 1-D array is ~10x faster than of rows -- 2 M vertices
    # per 512^3 shape), verified exactly; a hash collision between different rows falls back to the row-wise unique
    with np.errstate(over='ignore'):
        u = rows.astype(np.uint64)
        h = u[:, 0] * np.uint64(0x9E3779B97F4A7C15) + u[:, 1] * np.uint64(0xC2B2AE3D27D4EB4F) + u[:, 2] * np.uint64(0x165667B19E3779F9)
    _, first, inv = np.unique(h, return_index=True, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    if not np.array_equal(rows[first[inv]], rows):
        _, first, inv = np.unique(rows, axis=0, return_index=True, return_inverse=True)
        inv = np.asarray(inv).reshape(-1)
    order = np.argsort(first, kind='stable')                 # unique rows in order of first occurrence
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    inverse = np.zeros(len(v), dtype=np.int64)
    inverse[referenced] = rank[inv]
    mask = np.nonzero(referenced)[0][first[order]]
    return v[mask], inverse[f].astype(f.dtype)
