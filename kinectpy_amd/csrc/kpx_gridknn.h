// kpx_gridknn.h -- device-side pieces of the exact grid neighbour search shared by kpx_knn.hip and kpx_fpfh.hip.
#pragma once
#include "kpx_internal.h"

namespace kpx {

__device__ __forceinline__ int cell_coord(double v, double org, double h, int dim)
{
    int c = (int)floor((v - org) / h);
    return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

// ---- per-thread max-heaps in LDS ---------------------------------------------------------------------
// values only (SOR): slot e of this thread lives at h[e * stride]
struct HeapD {
    double *h; int stride, k, sz;
    __device__ bool full() const { return sz == k; }
    __device__ double worst() const { return h[0]; }
    __device__ void push(double d, int)
    {
        if (sz < k) {
            int c = sz++;
            while (c > 0) {
                int p = (c - 1) >> 1;
                double hp = h[p * stride];
                if (hp < d) { h[c * stride] = hp; c = p; } else break;
            }
            h[c * stride] = d;
        } else if (d < h[0]) {
            int c = 0;
            for (;;) {
                int l = 2 * c + 1, r = l + 1;
                if (l >= k) break;
                double hl = h[l * stride];
                int b = l; double hb = hl;
                if (r < k) { double hr = h[r * stride]; if (hr > hl) { b = r; hb = hr; } }
                if (hb > d) { h[c * stride] = hb; c = b; } else break;
            }
            h[c * stride] = d;
        }
    }
};
// (d2, idx) pairs ordered lexicographically (normals: the neighbour identities matter)
struct HeapDI {
    double *h; int32_t *ix; int stride, k, sz;
    __device__ bool full() const { return sz == k; }
    __device__ double worst() const { return h[0]; }
    static __device__ bool less(double a, int32_t ai, double b, int32_t bi) { return a < b || (a == b && ai < bi); }
    __device__ void push(double d, int j)
    {
        if (sz < k) {
            int c = sz++;
            while (c > 0) {
                int p = (c - 1) >> 1;
                double hp = h[p * stride]; int32_t ip = ix[p * stride];
                if (less(hp, ip, d, j)) { h[c * stride] = hp; ix[c * stride] = ip; c = p; } else break;
            }
            h[c * stride] = d; ix[c * stride] = j;
        } else if (less(d, j, h[0], ix[0])) {
            int c = 0;
            for (;;) {
                int l = 2 * c + 1, r = l + 1;
                if (l >= k) break;
                int b = l; double hb = h[l * stride]; int32_t ib = ix[l * stride];
                if (r < k) { double hr = h[r * stride]; int32_t ir = ix[r * stride]; if (less(hb, ib, hr, ir)) { b = r; hb = hr; ib = ir; } }
                if (less(d, j, hb, ib)) { h[c * stride] = hb; ix[c * stride] = ib; c = b; } else break;
            }
            h[c * stride] = d; ix[c * stride] = j;
        }
    }
};

// Ring walk.  r2max < 0: plain kNN; otherwise only candidates with d2 < r2max ([O3D] SearchHybrid).
template <class Heap>
__device__ __forceinline__ void grid_knn_scan(const GridParams &g, const uint32_t *__restrict__ cell_start,
                                              const float *__restrict__ spts, const int32_t *__restrict__ sidx, double qx,
                                              double qy, double qz, double r2max, Heap &heap)
{
    const double q[3] = { qx, qy, qz };
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = cell_coord(q[a], g.org[a], g.h, g.dim[a]);
    int maxr = g.dim[0] > g.dim[1] ? g.dim[0] : g.dim[1];
    if (g.dim[2] > maxr) maxr = g.dim[2];
    double gate = heap.full() ? heap.worst() : INFINITY;     // register copy of the heap's worst value
    for (int r = 0; r <= maxr; ++r) {
        if (r > 0) {
            double dcov = INFINITY;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                double lo = g.org[a] + (double)(c[a] - (r - 1)) * g.h;
                double hi = g.org[a] + (double)(c[a] + r) * g.h;
                double dl = (c[a] - (r - 1) <= 0) ? INFINITY : q[a] - lo;
                double dh = (c[a] + r >= g.dim[a]) ? INFINITY : hi - q[a];
                dcov = fmin(dcov, fmin(dl, dh));
            }
            if (dcov == INFINITY) break;
            if (dcov < 0.0) dcov = 0.0;
            double cov2 = dcov * dcov * (1.0 - 1e-12);
            if (heap.full() && heap.worst() < cov2) break;
            if (r2max >= 0.0 && cov2 >= r2max) break;
        }
        const int x0 = c[0] - r, x1 = c[0] + r, y0 = c[1] - r, y1 = c[1] + r, z0 = c[2] - r, z1 = c[2] + r;
        const int xa = x0 < 0 ? 0 : x0, xb = x1 >= g.dim[0] ? g.dim[0] - 1 : x1;
        const int ya = y0 < 0 ? 0 : y0, yb = y1 >= g.dim[1] ? g.dim[1] - 1 : y1;
        const int za = z0 < 0 ? 0 : z0, zb = z1 >= g.dim[2] ? g.dim[2] - 1 : z1;
        for (int x = xa; x <= xb; ++x)
            for (int y = ya; y <= yb; ++y) {
                const bool shell_xy = (x == x0) | (x == x1) | (y == y0) | (y == y1);
                const int64_t col = ((int64_t)x * g.dim[1] + y) * g.dim[2];
                // cells of one (x,y) column are contiguous in the sorted order
                for (int part = 0; part < 2; ++part) {
                    int zs, ze;
                    if (shell_xy) { if (part) break; zs = za; ze = zb; }
                    else {
                        if (part == 0) { if (z0 < 0) continue; zs = ze = z0; }
                        else { if (z1 >= g.dim[2] || r == 0) continue; zs = ze = z1; }
                    }
                    uint32_t s0 = cell_start[col + zs], s1 = cell_start[col + ze + 1];
                    // candidates in batches of 8: all coordinate loads of a batch are issued before the first push
                    // (a push is LDS traffic and branches; interleaved with the loads it exposes one memory latency
                    // per candidate)
                    for (uint32_t sb = s0; sb < s1; sb += 8) {
                        float cx[8], cy[8], cz[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const uint32_t s = sb + e < s1 ? sb + e : s1 - 1;
                            cx[e] = spts[3 * s]; cy[e] = spts[3 * s + 1]; cz[e] = spts[3 * s + 2];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (sb + e >= s1) break;
                            const double dx = qx - (double)cx[e], dy = qy - (double)cy[e], dz = qz - (double)cz[e];
                            const double d = fma(dz, dz, fma(dy, dy, dx * dx));
                            if (r2max >= 0.0 && !(d < r2max)) continue;
                            if (d > gate) continue;                 // cannot enter a full heap (ties are decided by push)
                            heap.push(d, sidx ? sidx[sb + e] : 0);
                            gate = heap.full() ? heap.worst() : INFINITY;
                        }
                    }
                }
            }
    }
}

}  // namespace kpx
