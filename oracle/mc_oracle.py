"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the iso-surface step.

The reference calls ``skimage.measure.marching_cubes_lewiner(volume, 0)`` (source/sdf.py:211-215, scikit-image >= 0.16,
requirements.txt:3), then ``v = ((v + 0.5) / res - 0.5) * 2`` (:223) and ``trimesh.repair.fix_inversion`` (:224-225).
Neither scikit-image nor trimesh is installed here and their sources cannot be fetched: **parity with skimage's
Lewiner tables (exact vertex / face counts) is UNPINNED.**  What this module restates is the published algorithm
family those tables implement -- marching cubes whose ambiguous faces are resolved by the asymptotic decider
(Nielson & Hamann 1991; Lewiner et al. 2003 "face test") -- with interior ambiguities resolved as separated
sheets, loops fanned from their smallest edge.  It is written independently of the device code (no shared table):
  * ``cell_triangles(case, face_bits)``  loops + fan for one configuration (compared with the library's generated
    table for all 256 x 64 configurations in tests/test_mc_oracle.py),
  * ``marching_cubes(volume)``           vertices / faces in the deterministic order the device emits,
  * ``mesh_checks(verts, faces)``        watertightness, orientation, Euler characteristic.
Checkable without skimage: closed 2-manifold with consistent orientation for volumes whose border is outside;
vertices on the level set of the trilinear interpolant's edges; V - E + F = 2 per genus-0 component.
"""
import numpy as np

# corner i = (dx, dy, dz) = (i & 1, i >> 1 & 1, i >> 2 & 1); edge id = axis * 4 + (u + 2 v)


def _others(axis):
    return [a for a in (0, 1, 2) if a != axis]


def edge_ends(e):
    a, u, v = e >> 2, e & 1, (e >> 1) & 1
    b, c = _others(a)
    c0 = (u << b) | (v << c)
    return c0, c0 | (1 << a)


def _edge_id(axis, coords):
    """edge along ``axis`` whose other two coordinates are given as {axis: value}"""
    b, c = _others(axis)
    return axis * 4 + coords[b] + 2 * coords[c]


def face_layout(f):
    a, s = f >> 1, f & 1
    b, c = _others(a)
    uv = [(0, 0), (1, 0), (1, 1), (0, 1)]
    corners = [(s << a) | (u << b) | (v << c) for u, v in uv]
    edges = [_edge_id(b, {a: s, c: 0}), _edge_id(c, {a: s, b: 1}), _edge_id(b, {a: s, c: 1}), _edge_id(c, {a: s, b: 0})]
    return corners, edges


_CPOS = np.array([[i & 1, (i >> 1) & 1, (i >> 2) & 1] for i in range(8)], dtype=np.float64)
_CACHE = {}


def _edge_faces(e):
    a, u, v = e >> 2, e & 1, (e >> 1) & 1
    b, c = _others(a)
    return {b * 2 + u, c * 2 + v}


def _triangulate(loop):
    """Triangles of an oriented loop such that no diagonal joins two vertices lying on edges of one cube face (the
    neighbouring cell could produce the same segment: four triangles on one mesh edge).  Interval DP, smallest apex on
    ties; returns (triangles, centre_loop): if every triangulation needs such a diagonal the loop is fanned around a
    centre vertex (edge id 12) and returned as centre_loop."""
    n = len(loop)

    def bad(i, j):
        return 1 if _edge_faces(loop[i]) & _edge_faces(loop[j]) else 0
    cost = [[0] * n for _ in range(n)]
    choice = [[0] * n for _ in range(n)]
    for length in range(2, n):
        for i in range(0, n - length):
            j = i + length
            best, bk = 1 << 20, i + 1
            for k in range(i + 1, j):
                c = cost[i][k] + cost[k][j] + (bad(i, k) if k > i + 1 else 0) + (bad(k, j) if j > k + 1 else 0)
                if c < best:
                    best, bk = c, k
            cost[i][j], choice[i][j] = best, bk
    if n >= 3 and cost[0][n - 1] > 0:
        return [(loop[k], loop[(k + 1) % n], 12) for k in range(n)], list(loop)
    tris = []

    def emit(i, j):
        if j - i < 2:
            return
        k = choice[i][j]
        tris.append((loop[i], loop[k], loop[j]))
        emit(i, k)
        emit(k, j)
    emit(0, n - 1)
    return tris, None


def cell_triangles(case, face_bits, with_center=False):
    """list of triangles (3 cube-edge ids each; 12 = the cell's centre vertex) of one configuration"""
    key = (case, face_bits)
    if key in _CACHE:
        return _CACHE[key] if with_center else _CACHE[key][0]
    r = _cell_triangles(case, face_bits)
    _CACHE[key] = r
    return r if with_center else r[0]


def _cell_triangles(case, face_bits):
    inside = [(case >> i) & 1 for i in range(8)]
    cut = [inside[edge_ends(e)[0]] != inside[edge_ends(e)[1]] for e in range(12)]
    succ = {}

    def link(f, e0, e1):
        """contour segment on face f between the crossing points of e0 and e1, directed so that the inside lies on
        its left when the face is seen from outside the cube: the loops then bound the inside region on the cube
        surface and the neighbouring cell (which sees the face from the other side) runs the segment backwards"""
        a, sgn = f >> 1, (1.0 if f & 1 else -1.0)
        m0 = 0.5 * (_CPOS[edge_ends(e0)[0]] + _CPOS[edge_ends(e0)[1]])
        m1 = 0.5 * (_CPOS[edge_ends(e1)[0]] + _CPOS[edge_ends(e1)[1]])
        c0, c1 = edge_ends(e0)
        cin = _CPOS[c0 if inside[c0] else c1]
        left = np.cross(m1 - m0, cin - m0)[a] * sgn
        if left > 0:
            succ[e0] = e1
        else:
            succ[e1] = e0
    for f in range(6):
        corners, edges = face_layout(f)
        ce = [k for k in range(4) if cut[edges[k]]]
        if len(ce) == 2:
            link(f, edges[ce[0]], edges[ce[1]])
        elif len(ce) == 4:
            inside_connected = (face_bits >> f) & 1
            for k in range(4):
                if inside[corners[k]] != inside_connected:       # this corner is cut off
                    link(f, edges[(k + 3) % 4], edges[k])
    tris, used, center = [], set(), None
    for start in range(12):
        if not cut[start] or start in used:
            continue
        loop, cur = [], start
        while True:
            loop.append(cur)
            used.add(cur)
            cur = succ[cur]
            if cur == start or len(loop) >= 12:
                break
        t, c = _triangulate(loop)
        tris += t
        if c is not None:
            center = c
    return tris, center


def marching_cubes(volume, model_space=True, fix_inversion=True):
    """returns (verts [V,3] float32, faces [F,3] int32, inverted)"""
    vol = np.ascontiguousarray(volume, dtype=np.float32)
    res = vol.shape[0]
    ins = vol > 0
    # ---- configuration of every cell
    case = np.zeros((res - 1, res - 1, res - 1), dtype=np.int32)
    cv = {}
    for i in range(8):
        dx, dy, dz = i & 1, (i >> 1) & 1, (i >> 2) & 1
        cv[i] = vol[dx:res - 1 + dx, dy:res - 1 + dy, dz:res - 1 + dz]
        case |= (cv[i] > 0).astype(np.int32) << i
    fb = np.zeros_like(case)
    for f in range(6):
        corners, _ = face_layout(f)
        A, B, C, D = (cv[c].astype(np.float64) for c in corners)
        iA, iB, iC, iD = (((case >> c) & 1).astype(bool) for c in corners)
        amb = (iA == iC) & (iB == iD) & (iA != iB)
        ac, bd = A * C, B * D
        conn = np.where(iA, ac - bd > 0, bd - ac > 0)
        fb |= (amb & conn).astype(np.int32) << f
    cx, cy, cz = np.nonzero((case != 0) & (case != 255))
    cells = [(int(x), int(y), int(z)) + cell_triangles(int(case[x, y, z]), int(fb[x, y, z]), with_center=True)
             for x, y, z in zip(cx, cy, cz)]

    def edge_point(x, y, z, e):
        a, u, v = e >> 2, e & 1, (e >> 1) & 1
        b, c = _others(a)
        q = [x, y, z]
        q[b] += u
        q[c] += v
        a0 = float(vol[q[0], q[1], q[2]])
        q1 = list(q)
        q1[a] += 1
        a1 = float(vol[q1[0], q1[1], q1[2]])
        p = np.array(q, dtype=np.float64)
        p[a] += a0 / (a0 - a1)
        return p
    # ---- vertices: per point (C order) its crossing edges towards +x, +y, +z, then the centre vertex of its cell
    own = np.zeros((res, res, res, 4), dtype=bool)
    own[:-1, :, :, 0] = ins[1:] != ins[:-1]
    own[:, :-1, :, 1] = ins[:, 1:] != ins[:, :-1]
    own[:, :, :-1, 2] = ins[:, :, 1:] != ins[:, :, :-1]
    centers = {}
    for x, y, z, tris, cloop in cells:
        if cloop is not None:
            own[x, y, z, 3] = True
            centers[(x, y, z)] = np.mean([edge_point(x, y, z, e) for e in cloop], axis=0) if False else \
                sum(edge_point(x, y, z, e) for e in cloop) / float(len(cloop))
    vid = np.cumsum(own.reshape(-1)).reshape(own.shape) - 1            # vertex index of (point, slot)
    px, py, pz, pa = np.nonzero(own)
    verts = np.zeros((len(pa), 3), dtype=np.float64)
    em = pa < 3
    ex, ey, ez, ea = px[em], py[em], pz[em], pa[em]
    a0 = vol[ex, ey, ez].astype(np.float64)
    nxt = np.stack([ex, ey, ez], axis=1)
    nxt[np.arange(len(ea)), ea] += 1
    a1 = vol[nxt[:, 0], nxt[:, 1], nxt[:, 2]].astype(np.float64)
    ev = np.stack([ex, ey, ez], axis=1).astype(np.float64)
    ev[np.arange(len(ea)), ea] += a0 / (a0 - a1)
    verts[em] = ev
    for k in np.nonzero(~em)[0]:
        verts[k] = centers[(int(px[k]), int(py[k]), int(pz[k]))]
    verts = verts.astype(np.float32).astype(np.float64)
    if model_space:
        verts = (((verts + 0.5) / float(res)) - 0.5) * 2.0
    verts = verts.astype(np.float32)
    # ---- faces: cells in C order
    faces = []
    for x, y, z, tris, cloop in cells:
        for tri in tris:
            row = []
            for e in tri:
                if e == 12:
                    row.append(vid[x, y, z, 3])
                    continue
                a, u, v = e >> 2, e & 1, (e >> 1) & 1
                b, c = _others(a)
                p = [x, y, z]
                p[b] += u
                p[c] += v
                row.append(vid[p[0], p[1], p[2], a])
            faces.append(row)
    faces = np.array(faces, dtype=np.int32).reshape(-1, 3)
    inverted = False
    if fix_inversion and len(faces):
        v = verts.astype(np.float64)
        vol6 = np.einsum('ij,ij->i', v[faces[:, 0]], np.cross(v[faces[:, 1]], v[faces[:, 2]])).sum()
        if vol6 < 0:
            faces = faces[:, [0, 2, 1]]
            inverted = True
    return verts, faces, inverted


def mesh_checks(verts, faces):
    """dict: closed (every undirected edge in exactly two faces), oriented (every directed edge once), euler, components"""
    f = np.asarray(faces, dtype=np.int64)
    if len(f) == 0:
        return {'closed': True, 'oriented': True, 'euler': 0, 'components': 0, 'V': len(verts), 'E': 0, 'F': 0}
    d = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    nv = int(max(len(verts), d.max() + 1))
    key_d = d[:, 0] * nv + d[:, 1]
    und = np.sort(d, axis=1)
    key_u = und[:, 0] * nv + und[:, 1]
    _, cnt_u = np.unique(key_u, return_counts=True)
    _, cnt_d = np.unique(key_d, return_counts=True)
    # connected components over the vertices that faces use (union-find on edges)
    parent = np.arange(nv)

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i
    for a, b in und[np.unique(key_u, return_index=True)[1]]:
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[ra] = rb
    usedv = np.unique(f)
    comps = len({find(i) for i in usedv})
    E = len(cnt_u)
    return {'closed': bool((cnt_u == 2).all()), 'oriented': bool((cnt_d == 1).all()), 'euler': int(len(usedv) - E + len(f)),
            'components': comps, 'V': int(len(verts)), 'E': int(E), 'F': int(len(f)), 'unused_vertices': int(len(verts) - len(usedv))}
