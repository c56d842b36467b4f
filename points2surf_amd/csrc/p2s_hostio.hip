// Host-side result writers: the text and debug files the reference writes per shape, formatted by
// native code instead of Python loops -- np.savetxt (source/points_to_surf_eval.py:210), sdf.visualize_query_points
// (source/sdf.py:269-285) and mesh_io.write_off of the coloured samples (source/sdf.py:203-209,
// source/base/mesh_io.py:75-140).  Pure host code (no device is touched): at 0.65 s of GPU time per 256^3 shape
// (fp16-pair encoder) the reference's formatting -- 0.3-0.8 s of savetxt, seconds of str() per vertex -- would be the
// bottleneck of points_to_surf_eval.  Byte-identical to what numpy / Python write (tests/test_hostio.py).
#include <charconv>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/p2s_hip.h"
#include "p2s_common.h"

namespace {

struct File {
    FILE *f = nullptr;
    explicit File(const char *path) { f = path ? fopen(path, "wb") : nullptr; }
    ~File() { if (f) fclose(f); }                     // error paths only: a successful writer has called close()
    bool put(const std::string &s) { return fwrite(s.data(), 1, s.size(), f) == s.size(); }
    // flush + close with their results checked: a full disk / quota / NFS error often surfaces only here
    bool close() {
        if (!f) return false;
        const bool ok = fflush(f) == 0 && !ferror(f);
        const bool closed = fclose(f) == 0;
        f = nullptr;
        return ok && closed;
    }
};
int io_fail(const char *who, const char *what, const char *path) {
    p2s_set_error("%s: %s %s: %s", who, what, path, strerror(errno));
    return P2S_EIO;
}

// shortest round-trip digits of a finite, non-zero |x| -> (digits, decimal exponent of the first digit)
template <typename T>
int shortest_digits(T ax, char *digits, int *exp10) {
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), ax, std::chars_format::scientific);
    int nd = 0;
    const char *p = buf;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') digits[nd++] = *p;
    int e = 0;
    std::from_chars(p + 1 + (p[1] == '+' ? 1 : 0), r.ptr, e);
    *exp10 = e;
    return nd;
}

// str(np.float32(x)) / str(np.float64(x)) = repr of a Python float: shortest round-trip digits; positional for
// 1e-4 <= |x| < 1e16 (and 0) with at least one digit behind the point, else scientific with a two-digit exponent
template <typename T>
void append_repr(std::string &out, T x) {
    if (std::isnan(x)) { out += "nan"; return; }
    if (std::signbit(x)) out += '-';
    const T ax = std::fabs(x);
    if (std::isinf(ax)) { out += "inf"; return; }
    if (ax == 0) { out += "0.0"; return; }
    char d[40];
    int e;
    const int nd = shortest_digits(ax, d, &e);
    // numpy compares the VALUE with 1e-4L / 1e16L in long double: float32(1e-4) = 9.99999975e-05 is scientific
    if ((long double)ax >= 1.e-4L && (long double)ax < 1.e16L) {
        if (e < 0) {
            out += "0.";
            out.append((size_t)(-e - 1), '0');
            out.append(d, nd);
        } else if (nd <= e + 1) {
            out.append(d, nd);
            out.append((size_t)(e + 1 - nd), '0');
            out += ".0";
        } else {
            out.append(d, e + 1);
            out += '.';
            out.append(d + e + 1, nd - e - 1);
        }
    } else {
        out += d[0];
        if (nd > 1) { out += '.'; out.append(d + 1, nd - 1); }
        char eb[8];
        snprintf(eb, sizeof(eb), "e%c%02d", e < 0 ? '-' : '+', e < 0 ? -e : e);
        out += eb;
    }
}

inline unsigned char to_u8(float c) {           // float_colors_to_rgba: round-half-even of c * 255 in float64, clipped
    if (!std::isfinite(c)) return 0;
    double v = std::nearbyint((double)c * 255.0);
    return (unsigned char)(v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v));
}

}  // namespace

extern "C" {

// np.savetxt(path, sdf) of a 1-D float32 array: one '%.18e' of the float64 value per line
int p2s_write_txt_f32(const char *path, const float *values_host, int64_t n) {
    if (!path || (n > 0 && !values_host) || n < 0) {
        p2s_set_error("p2s_write_txt_f32: bad argument");
        return P2S_EINVAL;
    }
    File fh(path);
    if (!fh.f) {
        return io_fail("p2s_write_txt_f32", "cannot open", path);
    }
    std::string out;
    out.reserve(1 << 20);
    char buf[48];
    for (int64_t i = 0; i < n; ++i) {
        const double v = (double)values_host[i];
        int len;
        if (std::isnan(v)) len = snprintf(buf, sizeof(buf), "nan\n");                    // Python: 'nan', never '-nan'
        else len = snprintf(buf, sizeof(buf), "%.18e\n", v);
        out.append(buf, (size_t)len);
        if (out.size() > (1 << 20) - 64) {
            if (!fh.put(out)) return io_fail("p2s_write_txt_f32", "write to", path);
            out.clear();
        }
    }
    if (!fh.put(out)) return io_fail("p2s_write_txt_f32", "write to", path);
    if (!fh.close()) return io_fail("p2s_write_txt_f32", "flush / close of", path);
    return P2S_OK;
}

// sdf.visualize_query_points(query_pts_ms, query_dist_ms, path) as the drop-in writes it (points2surf_amd/ply.py layout):
// binary PLY, float32 xyz + uchar rgba; red = negative, green = positive distance, brightness 0.5 + 0.5 |d| / max|d|
int p2s_write_query_vis_ply(const char *path, const float *query_host, const float *dist_host, int64_t n) {
    if (!path || (n > 0 && (!query_host || !dist_host)) || n < 0) {
        p2s_set_error("p2s_write_query_vis_ply: bad argument");
        return P2S_EINVAL;
    }
    File fh(path);
    if (!fh.f) {
        return io_fail("p2s_write_query_vis_ply", "cannot open", path);
    }
    char head[512];
    const int hl = snprintf(head, sizeof(head),
                            "ply\nformat binary_little_endian 1.0\ncomment points2surf_amd\nelement vertex %lld\n"
                            "property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\n"
                            "property uchar blue\nproperty uchar alpha\nelement face 0\nproperty list uchar int vertex_indices\n"
                            "end_header\n", (long long)n);
    if (fwrite(head, 1, (size_t)hl, fh.f) != (size_t)hl) return io_fail("p2s_write_query_vis_ply", "write to", path);
    float dmax = -INFINITY;                    // np.abs(d).max(): NaN propagates
    bool has_nan = false;
    for (int64_t i = 0; i < n; ++i) {
        const float a = std::fabs(dist_host[i]);
        if (std::isnan(a)) has_nan = true;
        else if (a > dmax) dmax = a;
    }
    if (has_nan) dmax = NAN;
    std::vector<unsigned char> rec((size_t)n * 16);
    for (int64_t i = 0; i < n; ++i) {
        unsigned char *r = rec.data() + (size_t)i * 16;
        memcpy(r, query_host + 3 * i, 12);
        const float d = dist_host[i];
        const float c = 0.5f + 0.5f * (std::fabs(d) / dmax);        // float32 arithmetic like numpy
        r[12] = d < 0.0f ? to_u8(c) : 0;
        r[13] = d > 0.0f ? to_u8(c) : 0;
        r[14] = 0;
        r[15] = 255;
    }
    if (n > 0 && fwrite(rec.data(), 16, (size_t)n, fh.f) != (size_t)n) return io_fail("p2s_write_query_vis_ply", "write to", path);
    if (!fh.close()) return io_fail("p2s_write_query_vis_ply", "flush / close of", path);
    return P2S_OK;
}

// mesh_io.write_off(path, query_pts_ms, [], colors_vertex=col) with the colours of source/sdf.py:203-208
// (norm = d / max|d|; red = |norm| + 0.5 for negative, green = norm + 0.5 for positive distances): 'COFF', one line
// 'x y z r g b ' per sample with the coordinates as str(np.float32) and the colours as str(np.float64)
int p2s_write_coff_samples(const char *path, const float *query_host, const float *dist_host, int64_t n) {
    if (!path || (n > 0 && (!query_host || !dist_host)) || n < 0) {
        p2s_set_error("p2s_write_coff_samples: bad argument");
        return P2S_EINVAL;
    }
    if (n == 0) return P2S_OK;                 // write_off returns before opening the file
    File fh(path);
    if (!fh.f) {
        return io_fail("p2s_write_coff_samples", "cannot open", path);
    }
    float dmax = -INFINITY;
    bool has_nan = false;
    for (int64_t i = 0; i < n; ++i) {
        const float a = std::fabs(dist_host[i]);
        if (std::isnan(a)) has_nan = true;
        else if (a > dmax) dmax = a;
    }
    if (has_nan) dmax = NAN;
    std::string out = "COFF\n" + std::to_string((long long)n) + " 0 0\n";
    out.reserve(1 << 20);
    for (int64_t i = 0; i < n; ++i) {
        const float norm = dist_host[i] / dmax;
        double col[3] = {0.0, 0.0, 0.0};
        if (norm < 0.0f) col[0] = (double)(std::fabs(norm) + 0.5f);
        if (norm > 0.0f) col[1] = (double)(norm + 0.5f);
        for (int k = 0; k < 3; ++k) {
            append_repr<float>(out, query_host[3 * i + k]);
            out += ' ';
        }
        for (int k = 0; k < 3; ++k) {
            append_repr<double>(out, col[k]);
            out += ' ';
        }
        out += '\n';
        if (out.size() > (1 << 20) - 256) {
            if (!fh.put(out)) return io_fail("p2s_write_coff_samples", "write to", path);
            out.clear();
        }
    }
    if (!fh.put(out)) return io_fail("p2s_write_coff_samples", "write to", path);
    if (!fh.close()) return io_fail("p2s_write_coff_samples", "flush / close of", path);
    return P2S_OK;
}

}  // extern "C"
