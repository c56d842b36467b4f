// "next" row f-4: mesh metrics on the device -- what reference source/base/evaluation.py:222-305 computes with trimesh
// + scipy on the host for every reconstructed mesh:
//     samples = trimesh.sample.sample_surface_even(mesh, 10000)         (:235, :269, :292)
//     Hausdorff: scipy.spatial.distance.directed_hausdorff both ways    (:301-303)
//     Chamfer  : sum of the nearest-neighbour distances both ways       (:249-254)
// trimesh is absent here; its sampling is restated from the published implementation (area-weighted face pick by
// searchsorted on the cumulative areas, folded barycentric coordinates, then rejection of points that have a
// neighbour within sqrt(area / (3 count))).  The reference draws from numpy's unseeded GLOBAL generator, so its numbers
// differ from run to run; here the uniform deviates come from the caller (a seeded device RandomState twin).
//   mesh_area / mesh_sample     face areas (float64) -> inclusive scan -> per sample binary search + barycentric
//   close_degree / close_remove remove_close: O(M^2) tiles over the 3 * count candidates (M = 30,000: 9e8 distance tests)
//   nn_dist                     exact nearest-neighbour distance in float64 through the cloud's cell index (k = 1)
//   reduce                      max and sum of a float64 array
#include "p2s_common.h"
#include "p2s_internal.h"
#include <algorithm>
#include <vector>

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ double rs_double(uint32_t w0, uint32_t w1) {      // numpy legacy random_sample
    return ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) / 9007199254740992.0;
}

__global__ __launch_bounds__(256) void rs_sample_kernel(const uint32_t *__restrict__ words, const long long *__restrict__ meta,
                                                        long long cap_words, long long n, double *__restrict__ out,
                                                        long long *__restrict__ err) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long c0 = meta[0];
    if (c0 + 2 * n > cap_words) {
        if (i == 0) err[0] = 2;
        return;
    }
    out[i] = rs_double(words[c0 + 2 * i], words[c0 + 2 * i + 1]);
}
__global__ void rs_advance_kernel(long long *meta, long long words) { meta[0] += words; }

// trimesh: area_faces = |cross(v1 - v0, v2 - v0)| / 2 (float64 from float32 vertices)
__global__ __launch_bounds__(256) void mesh_area_kernel(const float *__restrict__ verts, const int *__restrict__ faces, long long nf,
                                                        double *__restrict__ area) {
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    if (f >= nf) return;
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    const double a[3] = {(double)verts[3 * i1] - verts[3 * i0], (double)verts[3 * i1 + 1] - verts[3 * i0 + 1], (double)verts[3 * i1 + 2] - verts[3 * i0 + 2]};
    const double b[3] = {(double)verts[3 * i2] - verts[3 * i0], (double)verts[3 * i2 + 1] - verts[3 * i0 + 1], (double)verts[3 * i2 + 2] - verts[3 * i0 + 2]};
    const double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
    area[f] = sqrt((cx * cx + cy * cy) + cz * cz) * 0.5;
}

// inclusive scan in place (one workgroup, chunks of 1024): np.cumsum up to the association of the additions
__global__ __launch_bounds__(1024) void scan_f64_kernel(double *__restrict__ x, long long n) {
    __shared__ double ws[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double carry = 0.0;
    for (long long b0 = 0; b0 < n; b0 += 1024) {
        const long long i = b0 + tid;
        double v = i < n ? x[i] : 0.0;
        for (int d = 1; d < 64; d <<= 1) {
            const double u = __shfl_up(v, d);
            if (lane >= d) v += u;
        }
        if (lane == 63) ws[wave] = v;
        __syncthreads();
        double base = carry, tot = 0.0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) base += ws[w];
            tot += ws[w];
        }
        if (i < n) x[i] = base + v;
        carry += tot;
        __syncthreads();
    }
}

// sample i: face = searchsorted(cum, u0 * cum[-1]) ('left'); lengths (u1, u2) folded into the triangle:
// if u1 + u2 > 1: (u1, u2) -> |u - 1|; point = origin + u1 (v1 - v0) + u2 (v2 - v0)   (float64, stored as float32)
__global__ __launch_bounds__(256) void mesh_sample_kernel(const float *__restrict__ verts, const int *__restrict__ faces, long long nf,
                                                          const double *__restrict__ cum, const double *__restrict__ u, long long n,
                                                          float *__restrict__ out, int *__restrict__ face_out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // the reference draws all face picks first, then all (count, 2) lengths: u = [picks | lengths row-major]
    const double pick = u[i] * cum[nf - 1];
    long long lo = 0, hi = nf;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (cum[mid] < pick) lo = mid + 1;
        else hi = mid;
    }
    const long long f = lo < nf ? lo : nf - 1;
    double l1 = u[n + 2 * i], l2 = u[n + 2 * i + 1];
    if (l1 + l2 > 1.0) {
        l1 = fabs(l1 - 1.0);
        l2 = fabs(l2 - 1.0);
    }
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double o = verts[3 * i0 + d];
        const double e1 = (double)verts[3 * i1 + d] - o, e2 = (double)verts[3 * i2 + d] - o;
        out[3 * i + d] = (float)((e1 * l1 + e2 * l2) + o);
    }
    if (face_out) face_out[i] = (int)f;
}

// remove_close, pass 1: degree[i] = number of other points within radius (float64 distances of the float32 points)
__global__ __launch_bounds__(256) void close_degree_kernel(const float *__restrict__ pts, int m, double r2, int *__restrict__ degree) {
    __shared__ float tile[256 * 3];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double px = i < m ? pts[3 * i] : 0.0, py = i < m ? pts[3 * i + 1] : 0.0, pz = i < m ? pts[3 * i + 2] : 0.0;
    int deg = 0;
    for (int j0 = 0; j0 < m; j0 += 256) {
        const int j = j0 + threadIdx.x;
        if (j < m) {
            tile[3 * threadIdx.x] = pts[3 * j];
            tile[3 * threadIdx.x + 1] = pts[3 * j + 1];
            tile[3 * threadIdx.x + 2] = pts[3 * j + 2];
        }
        __syncthreads();
        const int lim = min(256, m - j0);
        for (int k = 0; k < lim; ++k) {
            const double dx = px - tile[3 * k], dy = py - tile[3 * k + 1], dz = pz - tile[3 * k + 2];
            deg += ((dx * dx + dy * dy) + dz * dz <= r2) && (j0 + k != i);
        }
        __syncthreads();
    }
    if (i < m) degree[i] = deg;
}
// pass 2: of every close pair (i < j) the point with the higher degree goes (ties: i), as trimesh's remove_close
__global__ __launch_bounds__(256) void close_remove_kernel(const float *__restrict__ pts, int m, double r2, const int *__restrict__ degree,
                                                           unsigned char *__restrict__ keep) {
    __shared__ float tile[256 * 3];
    __shared__ int tdeg[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double px = i < m ? pts[3 * i] : 0.0, py = i < m ? pts[3 * i + 1] : 0.0, pz = i < m ? pts[3 * i + 2] : 0.0;
    const int di = i < m ? degree[i] : 0;
    for (int j0 = (blockIdx.x * 256 / 256) * 256; j0 < m; j0 += 256) {      // only tiles that can hold j > i
        const int j = j0 + threadIdx.x;
        if (j < m) {
            tile[3 * threadIdx.x] = pts[3 * j];
            tile[3 * threadIdx.x + 1] = pts[3 * j + 1];
            tile[3 * threadIdx.x + 2] = pts[3 * j + 2];
            tdeg[threadIdx.x] = degree[j];
        }
        __syncthreads();
        const int lim = min(256, m - j0);
        if (i < m && di > 0) {
            for (int k = 0; k < lim; ++k) {
                const int jj = j0 + k;
                if (jj <= i) continue;
                const double dx = px - tile[3 * k], dy = py - tile[3 * k + 1], dz = pz - tile[3 * k + 2];
                if ((dx * dx + dy * dy) + dz * dz <= r2) keep[di >= tdeg[k] ? i : jj] = 0;
            }
        }
        __syncthreads();
    }
}

// ordered compaction of the kept points, at most `limit` of them (one workgroup)
__global__ __launch_bounds__(1024) void compact_points_kernel(const float *__restrict__ pts, const unsigned char *__restrict__ keep, int m,
                                                              int limit, float *__restrict__ out, int *__restrict__ n_out) {
    __shared__ int ws[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int carry = 0;
    for (int b0 = 0; b0 < m; b0 += 1024) {
        const int i = b0 + tid;
        const int k = i < m ? keep[i] : 0;
        int v = k;
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_up(v, d);
            if (lane >= d) v += u;
        }
        if (lane == 63) ws[wave] = v;
        __syncthreads();
        int base = carry, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) base += ws[w];
            tot += ws[w];
        }
        const int dst = base + v - k;
        if (k && dst < limit) {
            out[3 * dst] = pts[3 * i];
            out[3 * dst + 1] = pts[3 * i + 1];
            out[3 * dst + 2] = pts[3 * i + 2];
        }
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) *n_out = carry < limit ? carry : limit;
}

__global__ __launch_bounds__(256) void pair_distance_kernel(const float *__restrict__ a, const float *__restrict__ b, const int *__restrict__ ids,
                                                            long long n, double *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = ids[i];
    const double dx = (double)a[3 * i] - b[3 * j], dy = (double)a[3 * i + 1] - b[3 * j + 1], dz = (double)a[3 * i + 2] - b[3 * j + 2];
    out[i] = sqrt((dx * dx + dy * dy) + dz * dz);
}

// out[0] = max, out[1] = sum (sequential over 1024-element chunks inside one workgroup: deterministic)
__global__ __launch_bounds__(1024) void reduce_f64_kernel(const double *__restrict__ x, long long n, double *__restrict__ out) {
    __shared__ double wmax[16], wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double mx = 0.0, sm = 0.0;
    for (long long i = tid; i < n; i += 1024) {
        mx = fmax(mx, x[i]);
        sm += x[i];
    }
    for (int d = 32; d > 0; d >>= 1) {
        mx = fmax(mx, __shfl_xor(mx, d));
        sm += __shfl_xor(sm, d);
    }
    if (lane == 0) {
        wmax[wave] = mx;
        wsum[wave] = sm;
    }
    __syncthreads();
    if (tid == 0) {
        double m2 = 0.0, s2 = 0.0;
        for (int w = 0; w < 16; ++w) {
            m2 = fmax(m2, wmax[w]);
            s2 += wsum[w];
        }
        out[0] = m2;
        out[1] = s2;
    }
}

}  // namespace

extern "C" int p2s_rng_random_sample(p2s_rng_t r, int64_t n, double *out_dev, void *stream) {
    if (!r || n < 0 || (n > 0 && !out_dev)) {
        p2s_set_error("p2s_rng_random_sample: bad argument");
        return P2S_EINVAL;
    }
    if (r->levels_max == 0) {
        p2s_set_error("p2s_rng_random_sample: needs the jump-ahead tables (p2s_rng_set_jump_tables)");
        return P2S_EINVAL;
    }
    if (n == 0) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(r->device));
    hipStream_t s = (hipStream_t)stream;
    const long long cap = p2s_rng_session_words(r);
    const long long per = std::max<long long>(1, std::min<long long>(n, (cap - 1024) / 2));
    for (int64_t done = 0; done < n;) {
        const long long cur = std::min<long long>(per, n - done);
        const int rc = p2s_rng_session_raw(r, 2 * cur, s);
        if (rc) return rc;
        long long *meta = p2s_rng_raw_meta(r);
        hipLaunchKernelGGL(rs_sample_kernel, dim3((unsigned)((cur + 255) / 256)), dim3(256), 0, s, r->tmp, meta, cap, cur, out_dev + done,
                           meta + 1);
        hipLaunchKernelGGL(rs_advance_kernel, dim3(1), dim3(1), 0, s, meta, 2 * cur);
        P2S_LAUNCH_CHECK("rs_sample_kernel");
        done += cur;
    }
    return P2S_OK;
}

extern "C" int p2s_mesh_sample_surface(const float *verts_dev, const int32_t *faces_dev, int64_t n_faces, const double *u_dev,
                                       int64_t n_samples, float *pts_out_dev, int32_t *face_out_dev, double *area_host,
                                       int device, void *stream) {
    if (!verts_dev || !faces_dev || n_faces < 1 || n_samples < 0 || (n_samples > 0 && (!u_dev || !pts_out_dev))) {
        p2s_set_error("p2s_mesh_sample_surface: bad argument");
        return P2S_EINVAL;
    }
    if (p2s_device_count() <= device || device < 0) return P2S_ENODEVICE;
    P2S_HIP_CHECK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    double *cum = nullptr;
    if (hipMalloc(&cum, (size_t)n_faces * 8) != hipSuccess) {
        (void)hipGetLastError();
        p2s_set_error("p2s_mesh_sample_surface: hipMalloc failed");
        return P2S_ENOMEM;
    }
    hipLaunchKernelGGL(mesh_area_kernel, dim3((unsigned)((n_faces + 255) / 256)), dim3(256), 0, s, verts_dev, faces_dev, (long long)n_faces, cum);
    hipLaunchKernelGGL(scan_f64_kernel, dim3(1), dim3(1024), 0, s, cum, (long long)n_faces);
    if (n_samples > 0)
        hipLaunchKernelGGL(mesh_sample_kernel, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0, s, verts_dev, faces_dev,
                           (long long)n_faces, cum, u_dev, (long long)n_samples, pts_out_dev, face_out_dev);
    double total = 0.0;
    hipError_t e = hipMemcpyAsync(&total, cum + n_faces - 1, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(cum);
    if (e != hipSuccess) {
        p2s_set_error("p2s_mesh_sample_surface: %s", hipGetErrorString(e));
        return P2S_EHIP;
    }
    if (area_host) *area_host = total;
    return P2S_OK;
}

extern "C" int p2s_points_remove_close(const float *pts_dev, int64_t m, double radius, int64_t limit, float *pts_out_dev,
                                       int64_t *n_out, int device, void *stream) {
    if (!pts_dev || m < 0 || m > 2000000 || !pts_out_dev || !n_out || limit < 0) {
        p2s_set_error("p2s_points_remove_close: bad argument");
        return P2S_EINVAL;
    }
    if (p2s_device_count() <= device || device < 0) return P2S_ENODEVICE;
    *n_out = 0;
    if (m == 0) return P2S_OK;
    P2S_HIP_CHECK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    char *scratch = nullptr;
    if (hipMalloc(&scratch, (size_t)m * 5 + 64) != hipSuccess) {
        (void)hipGetLastError();
        p2s_set_error("p2s_points_remove_close: hipMalloc failed");
        return P2S_ENOMEM;
    }
    int *degree = (int *)scratch;
    unsigned char *keep = (unsigned char *)(degree + m);
    int *n_dev = nullptr;
    if (hipMalloc(&n_dev, 16) != hipSuccess) {
        (void)hipFree(scratch);
        return P2S_ENOMEM;
    }
    (void)hipMemsetAsync(keep, 1, (size_t)m, s);
    const unsigned g = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL(close_degree_kernel, dim3(g), dim3(256), 0, s, pts_dev, (int)m, radius * radius, degree);
    hipLaunchKernelGGL(close_remove_kernel, dim3(g), dim3(256), 0, s, pts_dev, (int)m, radius * radius, degree, keep);
    hipLaunchKernelGGL(compact_points_kernel, dim3(1), dim3(1024), 0, s, pts_dev, keep, (int)m, (int)std::min<int64_t>(limit, m),
                       pts_out_dev, n_dev);
    int h = 0;
    hipError_t e = hipMemcpyAsync(&h, n_dev, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    (void)hipFree(n_dev);
    if (e != hipSuccess) {
        p2s_set_error("p2s_points_remove_close: %s", hipGetErrorString(e));
        return P2S_EHIP;
    }
    *n_out = h;
    return P2S_OK;
}

extern "C" int p2s_nn_distance_stats(p2s_cloud_t target, const float *query_dev, int64_t n, double *dist_out_dev, double *max_host,
                                     double *sum_host, void *stream) {
    if (!target || !query_dev || n < 1) {
        p2s_set_error("p2s_nn_distance_stats: bad argument");
        return P2S_EINVAL;
    }
    P2S_HIP_CHECK(hipSetDevice(target->device));
    hipStream_t s = (hipStream_t)stream;
    char *scratch = nullptr;
    if (hipMalloc(&scratch, (size_t)n * 12 + 64) != hipSuccess) {
        (void)hipGetLastError();
        p2s_set_error("p2s_nn_distance_stats: hipMalloc failed");
        return P2S_ENOMEM;
    }
    double *dist = (double *)scratch;
    int *ids = (int *)(dist + n);
    double *red = (double *)(((uintptr_t)(ids + n) + 15) & ~(uintptr_t)15);
    int rc = p2s_knn_patch(target, query_dev, n, 1, ids, nullptr, nullptr, stream);      // exact (float64 ranking)
    if (rc) {
        (void)hipFree(scratch);
        return rc;
    }
    hipLaunchKernelGGL(pair_distance_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, query_dev, target->d.pts, ids, (long long)n,
                       dist);
    hipLaunchKernelGGL(reduce_f64_kernel, dim3(1), dim3(1024), 0, s, dist, (long long)n, red);
    double h[2] = {0.0, 0.0};
    hipError_t e = hipMemcpyAsync(h, red, 16, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && dist_out_dev) e = hipMemcpyAsync(dist_out_dev, dist, (size_t)n * 8, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    if (e != hipSuccess) {
        p2s_set_error("p2s_nn_distance_stats: %s", hipGetErrorString(e));
        return P2S_EHIP;
    }
    if (max_host) *max_host = h[0];
    if (sum_host) *sum_host = h[1];
    return P2S_OK;
}
