// Selection of the tiling of one cube -- the decision procedure of scikit-image's marching_cubes_lewiner (Lewiner,
// Lopes, Vieira, Tavares 2003: process_cube / test_face / test_interior), which the reference calls in
// source/sdf.py:213-215.  Included twice by p2s_mesh.hip: once as device code over __device__ copies of the look-up
// tables, once as host code (p2s_mc_cell, the single-cube diagnostic of the C ABI).  LW_FN = function qualifiers.
//
// Cube values c[0..7] are (value - level) in Lewiner's corner order, float64 (the grid holds float32: every product of
// two values is exact).  scikit-image specifics (pinned by tests/golden/mesh_cells_skimage.npz, see oracle/lewiner_mc.c):
// inside = value > 0; eps = 2^-52 everywhere; the interior test divides by (denominator + eps) and answers 0 where
// Lewiner's C++ falls through to `return s < 0`.

LW_FN bool lw_face(const double *c, int face) {
    // corners A, B, C, D of face |face| (A and C diagonal); sign of `face` and of A flip the answer
    const int f = face < 0 ? -face : face;
    const int ia = f == 1 ? 0 : f == 2 ? 1 : f == 3 ? 2 : f == 4 ? 3 : f == 5 ? 0 : 4;
    const int ib = f == 1 ? 4 : f == 2 ? 5 : f == 3 ? 6 : f == 4 ? 7 : f == 5 ? 3 : 7;
    const int ic = f == 1 ? 5 : f == 2 ? 6 : f == 3 ? 7 : f == 4 ? 4 : f == 5 ? 2 : 6;
    const int id = f == 1 ? 1 : f == 2 ? 2 : f == 3 ? 3 : f == 4 ? 0 : f == 5 ? 1 : 5;
    const double d = c[ia] * c[ic] - c[ib] * c[id];
    if (d > -LW_EPS && d < LW_EPS) return face >= 0;
    return ((double)face * c[ia]) * d >= 0.0;
}

// value at parameter t on the cube edge p -> q
LW_FN double lw_lerp(const double *c, int p, int q, double t) { return c[p] + (c[q] - c[p]) * t; }

// `edge` < 0: the slice of cases 4 / 10 (along Lewiner's z at the extremum of A C - B D); else the slice through the
// level crossing of that cube edge
LW_FN bool lw_interior(const double *c, int edge, int s) {
    double t, A, B, C, D;
    if (edge < 0) {
        const double a = (c[4] - c[0]) * (c[6] - c[2]) - (c[7] - c[3]) * (c[5] - c[1]);
        const double b = c[2] * (c[4] - c[0]) + c[0] * (c[6] - c[2]) - c[1] * (c[7] - c[3]) - c[3] * (c[5] - c[1]);
        t = -b / ((a + a) + LW_EPS);
        if (0.0 > t || t > 1.0) return s > 0;
        A = lw_lerp(c, 0, 4, t);
        B = lw_lerp(c, 3, 7, t);
        C = lw_lerp(c, 2, 6, t);
        D = lw_lerp(c, 1, 5, t);
    } else {
        // corners of edge e: ring edges 0-3 (bottom), 4-7 (top): p = e, q = next in the ring; 8-11 vertical: p = e - 8
        const int p = edge < 8 ? edge : edge - 8;
        const int q = edge < 4 ? ((edge + 1) & 3) : edge < 8 ? 4 + ((edge + 1) & 3) : edge - 4;
        t = c[p] / ((c[p] - c[q]) + LW_EPS);
        // the three parallel edges, oriented like p -> q: neighbour (B), opposite (C), other neighbour (D)
        int bp, bq, cp, cq, dp, dq;
        if (edge >= 8) {            // vertical: the other three corners of the bottom ring
            bp = (p + 3) & 3; cp = (p + 2) & 3; dp = (p + 1) & 3;
            bq = bp + 4; cq = cp + 4; dq = dp + 4;
        } else {                    // ring edge: same edge of the other ring (D), the two opposite ones (B below/above, C)
            const int base = edge & 4, k = edge & 3;
            const int op = (k + 3) & 3, oq = (k + 2) & 3;          // opposite edge of the same ring, same direction
            bp = base + op; bq = base + oq;
            cp = (base ^ 4) + op; cq = (base ^ 4) + oq;
            dp = (base ^ 4) + k; dq = (base ^ 4) + ((k + 1) & 3);
        }
        A = 0.0;
        B = lw_lerp(c, bp, bq, t);
        C = lw_lerp(c, cp, cq, t);
        D = lw_lerp(c, dp, dq, t);
    }
    const int pattern = (A >= 0.0 ? 1 : 0) | (B >= 0.0 ? 2 : 0) | (C >= 0.0 ? 4 : 0) | (D >= 0.0 ? 8 : 0);
    // bit p of the masks: what the procedure answers for that pattern
    if ((0x135fu >> pattern) & 1u) return s > 0;                     // 0,1,2,3,4,6,8,9,12
    if (pattern == 5) return (A * C - B * D < LW_EPS) ? (s > 0) : false;
    if (pattern == 10) return (A * C - B * D >= LW_EPS) ? (s > 0) : false;
    return s < 0;                                                   // 7, 11, 13, 14, 15
}

// -> row of LW_TRI (or -1: no triangles); *n_tri, *use_c (the tiling refers to the centre vertex, id 12)
LW_FN int lw_select(const double *c, int *n_tri, int *use_c) {
    int idx = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) idx |= (c[p] > 0.0 ? 1 : 0) << p;
    const int kase = LW_CASES[idx][0], cfg = LW_CASES[idx][1];
    int base = -1, sub = 0, subs = 1, nt = 0, uc = 0;
#define LW_PICK(T, S) (base = LW_##T##_BASE, subs = LW_##T##_SUB, nt = LW_##T##_NTRI, uc = LW_##T##_USEC, sub = (S))
    switch (kase) {
    case 1: LW_PICK(TILING1, 0); break;
    case 2: LW_PICK(TILING2, 0); break;
    case 3:
        if (lw_face(c, LW_TEST3[cfg])) LW_PICK(TILING3_2, 0);
        else LW_PICK(TILING3_1, 0);
        break;
    case 4:
        if (lw_interior(c, -1, LW_TEST4[cfg])) LW_PICK(TILING4_1, 0);
        else LW_PICK(TILING4_2, 0);
        break;
    case 5: LW_PICK(TILING5, 0); break;
    case 6:
        if (lw_face(c, LW_TEST6[cfg][0])) LW_PICK(TILING6_2, 0);
        else if (lw_interior(c, LW_TEST6[cfg][2], LW_TEST6[cfg][1])) LW_PICK(TILING6_1_1, 0);
        else LW_PICK(TILING6_1_2, 0);
        break;
    case 7: {
        const int m = (lw_face(c, LW_TEST7[cfg][0]) ? 1 : 0) | (lw_face(c, LW_TEST7[cfg][1]) ? 2 : 0) |
                      (lw_face(c, LW_TEST7[cfg][2]) ? 4 : 0);
        if (m == 0) LW_PICK(TILING7_1, 0);
        else if (m == 1 || m == 2 || m == 4) LW_PICK(TILING7_2, m >> 1);            // 1, 2, 4 -> sub 0, 1, 2
        else if (m == 3 || m == 5 || m == 6) LW_PICK(TILING7_3, m == 3 ? 0 : (m == 5 ? 1 : 2));
        else if (lw_interior(c, LW_TEST7[cfg][4], LW_TEST7[cfg][3])) LW_PICK(TILING7_4_2, 0);
        else LW_PICK(TILING7_4_1, 0);
        break;
    }
    case 8: LW_PICK(TILING8, 0); break;
    case 9: LW_PICK(TILING9, 0); break;
    case 10: {
        const bool f0 = lw_face(c, LW_TEST10[cfg][0]), f1 = lw_face(c, LW_TEST10[cfg][1]);
        if (f0 && f1) LW_PICK(TILING10_1_1_, 0);
        else if (f0) LW_PICK(TILING10_2, 0);
        else if (f1) LW_PICK(TILING10_2_, 0);
        else if (lw_interior(c, -1, LW_TEST10[cfg][2])) LW_PICK(TILING10_1_1, 0);
        else LW_PICK(TILING10_1_2, 0);
        break;
    }
    case 11: LW_PICK(TILING11, 0); break;
    case 12: {
        const bool f0 = lw_face(c, LW_TEST12[cfg][0]), f1 = lw_face(c, LW_TEST12[cfg][1]);
        if (f0 && f1) LW_PICK(TILING12_1_1_, 0);
        else if (f0) LW_PICK(TILING12_2, 0);
        else if (f1) LW_PICK(TILING12_2_, 0);
        else if (lw_interior(c, LW_TEST12[cfg][3], LW_TEST12[cfg][2])) LW_PICK(TILING12_1_1, 0);
        else LW_PICK(TILING12_1_2, 0);
        break;
    }
    case 13: {
        int m = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) m |= (lw_face(c, LW_TEST13[cfg][k]) ? 1 : 0) << k;
        const int sc = LW_SUBCONFIG13[m];
        if (sc == 0) LW_PICK(TILING13_1, 0);
        else if (sc >= 1 && sc <= 6) LW_PICK(TILING13_2, sc - 1);
        else if (sc >= 7 && sc <= 18) LW_PICK(TILING13_3, sc - 7);
        else if (sc >= 19 && sc <= 22) LW_PICK(TILING13_4, sc - 19);
        else if (sc >= 23 && sc <= 26) {
            // reference edge of the interior test = first edge id of the 13.5.1 tiling of this sub-configuration
            const int e = LW_TRI[LW_TILING13_5_1_BASE + cfg * LW_TILING13_5_1_SUB + (sc - 23)][0];
            if (lw_interior(c, e, LW_TEST13[cfg][6])) LW_PICK(TILING13_5_1, sc - 23);
            else LW_PICK(TILING13_5_2, sc - 23);
        } else if (sc >= 27 && sc <= 38) LW_PICK(TILING13_3_, sc - 27);
        else if (sc >= 39 && sc <= 44) LW_PICK(TILING13_2_, sc - 39);
        else if (sc == 45) LW_PICK(TILING13_1_, 0);
        break;                                   // else: Lewiner's "impossible case 13" -- nothing is emitted
    }
    case 14: LW_PICK(TILING14, 0); break;
    default: break;
    }
#undef LW_PICK
    *n_tri = nt;
    *use_c = uc;
    return base < 0 ? -1 : base + cfg * subs + sub;
}
