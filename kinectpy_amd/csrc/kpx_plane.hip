// kpx_plane.hip -- a21: PointCloud.segment_plane(distance_threshold, ransac_n, num_iterations)
// (floor_removal.py:70).  All hypotheses are generated and scored in one batch:
//   1. one thread per hypothesis draws its sample (Philox4x32-10 counter RNG, duplicates rejected)
//      and fits the plane (triangle normal for n=3, determinant least-squares fit otherwise);
//   2. scoring: thread <-> hypothesis, loop over a chunk of points (wave-uniform point index, so the
//      point is a scalar operand and no cross-lane reduction is needed); per-chunk (count, sum|d|);
//   3. chunk partials are added in chunk order, the sequential better-than scan of Open3D
//      (with its probabilistic early exit) is replayed by one thread;
//   4. inlier compaction for the winning plane, re-fit to the inliers.
// Distance (contract): |fma(a,x, fma(b,y, fma(c,z, d)))| < thr.
#include "kpx_internal.h"

namespace kpx {

__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// plane through the sample: identical operation order to the oracle (sequential sums)
__device__ void fit_plane_ids(const float *__restrict__ pts, const int32_t *__restrict__ ids, int m, double pl[4])
{
    pl[0] = pl[1] = pl[2] = pl[3] = 0.0;
    if (m == 3) {
        const float *p0 = pts + 3 * (int64_t)ids[0], *p1 = pts + 3 * (int64_t)ids[1], *p2 = pts + 3 * (int64_t)ids[2];
        double e0[3], e1[3];
        for (int a = 0; a < 3; ++a) { e0[a] = (double)p1[a] - (double)p0[a]; e1[a] = (double)p2[a] - (double)p0[a]; }
        double a = e0[1] * e1[2] - e0[2] * e1[1], b = e0[2] * e1[0] - e0[0] * e1[2], c = e0[0] * e1[1] - e0[1] * e1[0];
        double nn = sqrt(a * a + b * b + c * c);
        if (nn == 0.0) return;
        a /= nn; b /= nn; c /= nn;
        pl[0] = a; pl[1] = b; pl[2] = c; pl[3] = -(a * (double)p0[0] + b * (double)p0[1] + c * (double)p0[2]);
        return;
    }
    double cx = 0, cy = 0, cz = 0;
    for (int t = 0; t < m; ++t) { int64_t j = ids[t]; cx += (double)pts[3 * j]; cy += (double)pts[3 * j + 1]; cz += (double)pts[3 * j + 2]; }
    cx /= (double)m; cy /= (double)m; cz /= (double)m;
    double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
    for (int t = 0; t < m; ++t) {
        int64_t j = ids[t];
        double rx = (double)pts[3 * j] - cx, ry = (double)pts[3 * j + 1] - cy, rz = (double)pts[3 * j + 2] - cz;
        xx += rx * rx; xy += rx * ry; xz += rx * rz; yy += ry * ry; yz += ry * rz; zz += rz * rz;
    }
    double det_x = yy * zz - yz * yz, det_y = xx * zz - xz * xz, det_z = xx * yy - xy * xy;
    double a, b, c;
    if (det_x > det_y && det_x > det_z) { a = det_x; b = xz * yz - xy * zz; c = xy * yz - xz * yy; }
    else if (det_y > det_z)             { a = xz * yz - xy * zz; b = det_y; c = xy * xz - yz * xx; }
    else                                { a = xy * yz - xz * yy; b = xy * xz - yz * xx; c = det_z; }
    double nn = sqrt(a * a + b * b + c * c);
    if (nn == 0.0) return;
    a /= nn; b /= nn; c /= nn;
    pl[0] = a; pl[1] = b; pl[2] = c; pl[3] = -(a * cx + b * cy + c * cz);
}

__global__ __launch_bounds__(64) void plane_hyp_kernel(const float *__restrict__ pts, int64_t n, int ransac_n, int H, uint32_t seed_lo,
                                                       uint32_t seed_hi, int32_t *__restrict__ ids_ws, double *__restrict__ hyp)
{
    int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    int32_t *ids = ids_ws + (int64_t)h * ransac_n;
    int got = 0;
    for (uint32_t blk = 0; got < ransac_n; ++blk) {
        uint32_t out[4];
        philox4x32(blk, (uint32_t)h, 0u, 0u, seed_lo, seed_hi, out);
        for (int w = 0; w < 4 && got < ransac_n; ++w) {
            int32_t id = (int32_t)(((uint64_t)out[w] * (uint64_t)n) >> 32);
            bool dup = false;
            for (int t = 0; t < got; ++t) dup |= (ids[t] == id);
            if (!dup) ids[got++] = id;
        }
    }
    double pl[4];
    fit_plane_ids(pts, ids, ransac_n, pl);
    hyp[4 * h] = pl[0]; hyp[4 * h + 1] = pl[1]; hyp[4 * h + 2] = pl[2]; hyp[4 * h + 3] = pl[3];
}

constexpr int kScoreThreads = 256;

__global__ __launch_bounds__(kScoreThreads) void plane_score_kernel(const float *__restrict__ pts, int64_t n, int64_t chunk,
                                                                    const double *__restrict__ hyp, int H, double thr,
                                                                    uint32_t *__restrict__ part_cnt, double *__restrict__ part_err)
{
    const int h = blockIdx.x * kScoreThreads + threadIdx.x;
    const int hh = h < H ? h : H - 1;
    const double a = hyp[4 * hh], b = hyp[4 * hh + 1], c = hyp[4 * hh + 2], d = hyp[4 * hh + 3];
    const int64_t i0 = (int64_t)blockIdx.y * chunk;
    const int64_t i1 = i0 + chunk < n ? i0 + chunk : n;
    uint32_t cnt = 0;
    double err = 0.0;
    for (int64_t i = i0; i < i1; ++i) {            // i is wave-uniform: the point is a scalar operand
        double x = (double)pts[3 * i], y = (double)pts[3 * i + 1], z = (double)pts[3 * i + 2];
        double dist = fabs(fma(a, x, fma(b, y, fma(c, z, d))));
        bool in = dist < thr;
        cnt += in ? 1u : 0u;
        err += in ? dist : 0.0;
    }
    if (h < H) {
        part_cnt[(int64_t)blockIdx.y * H + h] = cnt;
        part_err[(int64_t)blockIdx.y * H + h] = err;
    }
}

__global__ __launch_bounds__(256) void plane_reduce_kernel(const uint32_t *__restrict__ part_cnt, const double *__restrict__ part_err,
                                                           int chunks, int H, int64_t *__restrict__ cnt, double *__restrict__ err)
{
    int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    int64_t c = 0; double e = 0.0;
    for (int k = 0; k < chunks; ++k) { c += part_cnt[(int64_t)k * H + h]; e += part_err[(int64_t)k * H + h]; }
    cnt[h] = c; err[h] = e;
}

// replay of the sequential loop of [O3D] SegmentPlane (better-than test + probabilistic early exit)
__global__ void plane_select_kernel(const double *__restrict__ hyp, const int64_t *__restrict__ cnt, const double *__restrict__ err,
                                    int H, int64_t n, int ransac_n, double probability, double *__restrict__ best)
{
    if (threadIdx.x || blockIdx.x) return;
    double best_fit = 0.0, best_rmse = 0.0, bp[4] = { 0, 0, 0, 0 };
    double break_it = INFINITY;
    for (int it = 0; it < H; ++it) {
        if ((double)it > break_it) break;
        double a = hyp[4 * it], b = hyp[4 * it + 1], c = hyp[4 * it + 2], d = hyp[4 * it + 3];
        if (a == 0.0 && b == 0.0 && c == 0.0 && d == 0.0) continue;
        int64_t k = cnt[it];
        double fit = k ? (double)k / (double)n : 0.0;
        double rmse = k ? err[it] / sqrt((double)k) : 0.0;
        if (fit > best_fit || (fit == best_fit && rmse < best_rmse)) {
            best_fit = fit; best_rmse = rmse; bp[0] = a; bp[1] = b; bp[2] = c; bp[3] = d;
            if (fit < 1.0) {
                double bi = log(1.0 - probability) / log(1.0 - pow(fit, (double)ransac_n));
                bi = bi < (double)H ? bi : (double)H;
                break_it = floor(bi);
            } else break_it = 0.0;
        }
    }
    best[0] = bp[0]; best[1] = bp[1]; best[2] = bp[2]; best[3] = bp[3];
}

struct PlaneInlierPred {
    const float *pts; const double *pl; double thr;
    __device__ bool operator()(int64_t i, int) const
    {
        double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        return fabs(fma(pl[0], x, fma(pl[1], y, fma(pl[2], z, pl[3])))) < thr;
    }
};
struct PlaneIdxEmit {
    int32_t *idx;
    __device__ void operator()(int64_t i, int, int32_t dst) const { idx[dst] = (int32_t)i; }
};

// re-fit to the inliers: pass 0 sums (x,y,z,1), pass 1 the centred second moments
__global__ __launch_bounds__(256) void plane_refit_sum_kernel(const float *__restrict__ pts, int64_t n, const double *__restrict__ pl,
                                                              double thr, const double *__restrict__ cen, int pass,
                                                              double *__restrict__ part /* [blocks][6] */)
{
    __shared__ double sh[4];
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    const double a = pl[0], b = pl[1], c = pl[2], d = pl[3];
    double cx = 0, cy = 0, cz = 0;
    if (pass) { cx = cen[0]; cy = cen[1]; cz = cen[2]; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        if (!(fabs(fma(a, x, fma(b, y, fma(c, z, d)))) < thr)) continue;
        if (pass == 0) { acc[0] += x; acc[1] += y; acc[2] += z; acc[3] += 1.0; }
        else {
            double rx = x - cx, ry = y - cy, rz = z - cz;
            acc[0] += rx * rx; acc[1] += rx * ry; acc[2] += rx * rz; acc[3] += ry * ry; acc[4] += ry * rz; acc[5] += rz * rz;
        }
    }
    for (int q = 0; q < 6; ++q) {
        double v = block_sum(acc[q], sh);
        if (threadIdx.x == 0) part[(int64_t)blockIdx.x * 6 + q] = v;
    }
}
__global__ void plane_refit_final_kernel(const double *__restrict__ part, int nb, int pass, double *__restrict__ cen, double *__restrict__ plane)
{
    if (threadIdx.x || blockIdx.x) return;
    double s[6] = { 0, 0, 0, 0, 0, 0 };
    for (int b = 0; b < nb; ++b) for (int q = 0; q < 6; ++q) s[q] += part[(int64_t)b * 6 + q];
    if (pass == 0) {
        double m = s[3];
        cen[3] = m;
        cen[0] = m > 0 ? s[0] / m : 0.0; cen[1] = m > 0 ? s[1] / m : 0.0; cen[2] = m > 0 ? s[2] / m : 0.0;
        return;
    }
    plane[0] = plane[1] = plane[2] = plane[3] = 0.0;
    if (cen[3] < 3.0) return;
    double xx = s[0], xy = s[1], xz = s[2], yy = s[3], yz = s[4], zz = s[5];
    double det_x = yy * zz - yz * yz, det_y = xx * zz - xz * xz, det_z = xx * yy - xy * xy;
    double a, b, c;
    if (det_x > det_y && det_x > det_z) { a = det_x; b = xz * yz - xy * zz; c = xy * yz - xz * yy; }
    else if (det_y > det_z)             { a = xz * yz - xy * zz; b = det_y; c = xy * xz - yz * xx; }
    else                                { a = xy * yz - xz * yy; b = xy * xz - yz * xx; c = det_z; }
    double nn = sqrt(a * a + b * b + c * c);
    if (nn == 0.0) return;
    a /= nn; b /= nn; c /= nn;
    plane[0] = a; plane[1] = b; plane[2] = c; plane[3] = -(a * cen[0] + b * cen[1] + c * cen[2]);
}

static int plane_impl(const float *pts, int64_t n, double thr, int ransac_n, int H, double probability, uint64_t seed,
                      double *d_plane, int32_t *inl_idx, int32_t *d_count, Arena &a, hipStream_t st)
{
    int64_t chunk = 2048;
    if (cdiv(n > 0 ? n : 1, chunk) > 1024) chunk = cdiv(n, 1024);
    const int chunks = (int)cdiv(n > 0 ? n : 1, chunk);
    const size_t HH = (size_t)(H > 0 ? H : 1);
    int32_t *ids = a.get<int32_t>(HH * (size_t)ransac_n);
    double *hyp = a.get<double>(HH * 4);
    uint32_t *part_cnt = a.get<uint32_t>((size_t)chunks * HH);
    double *part_err = a.get<double>((size_t)chunks * HH);
    int64_t *cnt = a.get<int64_t>(HH);
    double *err = a.get<double>(HH);
    double *best = a.get<double>(8);
    double *rpart = a.get<double>(1024 * 6);
    int32_t *counts = a.get<int32_t>((size_t)compact_ws_ints(n));
    if (a.dry) return KPX_OK;
    KPX_ARENA_CHECK(a);
    if (H > 0) {
        hipLaunchKernelGGL(plane_hyp_kernel, dim3((unsigned)cdiv(H, 64)), dim3(64), 0, st, pts, n, ransac_n, H, (uint32_t)seed,
                           (uint32_t)(seed >> 32), ids, hyp);
        {
            ProfScope prof(KPX_PROF_PLANE_SCORE, 12.0 * (double)n, st);      // one algorithmic sweep of the points for all H
            hipLaunchKernelGGL(plane_score_kernel, dim3((unsigned)cdiv(H, kScoreThreads), chunks), dim3(kScoreThreads), 0, st, pts, n,
                               chunk, hyp, H, thr, part_cnt, part_err);
        }
        hipLaunchKernelGGL(plane_reduce_kernel, dim3((unsigned)cdiv(H, 256)), dim3(256), 0, st, part_cnt, part_err, chunks, H, cnt, err);
    }
    hipLaunchKernelGGL(plane_select_kernel, dim3(1), dim3(1), 0, st, hyp, cnt, err, H, n, ransac_n, probability, best);
    int rc = compact(PlaneInlierPred{ pts, best, thr }, PlaneIdxEmit{ inl_idx }, n, 1, counts, d_count, st);
    if (rc) return rc;
    int nb = (int)(cdiv(n, 256 * 8) < 1 ? 1 : (cdiv(n, 256 * 8) > 1024 ? 1024 : cdiv(n, 256 * 8)));
    double *cen = best + 4;
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(plane_refit_sum_kernel, dim3(nb), dim3(256), 0, st, pts, n, best, thr, cen, pass, rpart);
        hipLaunchKernelGGL(plane_refit_final_kernel, dim3(1), dim3(1), 0, st, rpart, nb, pass, cen, d_plane);
    }
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

}  // namespace kpx

using namespace kpx;

KPX_EXPORT size_t kpx_segment_plane_workspace_bytes(int64_t n, int32_t ransac_n, int32_t num_iterations)
{
    Arena a(nullptr, 0);
    plane_impl(nullptr, n, 1.0, ransac_n < 3 ? 3 : ransac_n, num_iterations, 0.5, 0, nullptr, nullptr, nullptr, a, nullptr);
    return a.off;
}
KPX_EXPORT int kpx_segment_plane(const float *pts, int64_t n, double distance_threshold, int32_t ransac_n,
                                 int32_t num_iterations, double probability, uint64_t seed, double *d_plane,
                                 int32_t *inlier_idx, int32_t *d_count, void *ws, size_t ws_bytes, void *stream)
{
    // [O3D] "ransac_n should be at least 3", "There must be at least 'ransac_n' points", probability in (0,1]
    KPX_REQUIRE(ransac_n >= 3, "segment_plane: ransac_n should be set to higher than or equal to 3");
    KPX_REQUIRE(n >= ransac_n, "segment_plane: there must be at least 'ransac_n' points");
    KPX_REQUIRE(probability > 0.0 && probability <= 1.0, "segment_plane: probability must be > 0 and <= 1.0");
    KPX_REQUIRE(num_iterations >= 0 && n < ((int64_t)1 << 31), "segment_plane: bad size");
    KPX_REQUIRE(pts && d_plane && inlier_idx && d_count && ws, "segment_plane: null pointer");
    Arena a(ws, ws_bytes);
    return plane_impl(pts, n, distance_threshold, ransac_n, num_iterations, probability, seed, d_plane, inlier_idx, d_count, a,
                      (hipStream_t)stream);
}
