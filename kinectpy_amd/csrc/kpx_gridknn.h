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

// ---- wave-per-query neighbour selection (SOR, normals, FPFH neighbour lists) -------------------------------------------------------
// The lanes gather the squared distances (AC3, fp64) of the points of the (2r+1)^3 cell block around the query into
// LDS (every (x, y) column of the block is one contiguous run of the cell-sorted points: 64 columns at a time, lane c
// looks up run c, a wave scan places the runs, all candidates of the chunk are fetched together and the ones that pass
// d^2 <= tau, d^2 < r2max are appended by ballot + popcount).  The k-th smallest is found by bisection on the IEEE
// bit patterns (d^2 >= 0: the patterns order like the values; one ballot + popcount per 64 candidates and step) and
// the search ends when that value lies inside the distance the block covers -- the termination rule of the ring walk
// of kpx_gridknn.h.  Otherwise the k-th candidate found so far bounds the true k-th distance: the candidates are
// gathered again, only those within it, from the block that covers it.  When the buffer fills, gathering stops; the
// cap candidates held are still real points, so their k-th smallest is a valid bound too.
__device__ __forceinline__ unsigned long long wave_all_min_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ unsigned long long wave_all_max_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}
// squared distance the block [c-r, c+r] covers around q (infinity once it holds the whole grid)
__device__ __forceinline__ double block_cover2(const GridParams &g, const double q[3], const int c[3], int r)
{
    double dcov = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double lo = g.org[a] + (double)(c[a] - r) * g.h;
        const double hi = g.org[a] + (double)(c[a] + r + 1) * g.h;
        const double dl = (c[a] - r <= 0) ? INFINITY : q[a] - lo;
        const double dh = (c[a] + r + 1 >= g.dim[a]) ? INFINITY : hi - q[a];
        dcov = fmin(dcov, fmin(dl, dh));
    }
    if (dcov == INFINITY) return INFINITY;
    if (dcov < 0.0) dcov = 0.0;
    return dcov * dcov * (1.0 - 1e-12);
}

struct WaveKnnScratch {        // LDS owned by one wave
    double *vals;              // cap candidate distances
    uint32_t *pos;             // cap sorted positions of the candidates (POS only)
    uint32_t *run_s0;          // 64
    int32_t *run_off;          // 64
    int cap;
    uint32_t *hist;            // kKnnBuckets bucket counts of the counting selection (16-byte aligned)
};
constexpr int kKnnBuckets = 256;

// ---- k-th smallest by COUNTING (round 5) ---------------------------------------------------------------------------------------------
// The rank-th smallest (1-based) of the 32-bit keys key(t), t < m, that are in the set and lie in the window [a, a + range]: the window is
// cut into <= 256 buckets of 2^shift, every key inside adds one to its bucket (LDS atomics), lane l sums its four buckets, a wave scan
// finds the bucket of the rank-th and the next level looks inside that bucket only.  It ends with the key itself (a bucket one value wide,
// exact = false) or -- usually after two levels -- with the upper end of a bucket whose LAST key is the rank-th: a pivot with exactly
// `rank` window keys at or below it (exact = true).  The caller has taken the keys below the window out of `rank` and knows that none
// above it matters.  A bisection over the same values took ~24 (high words) to ~60 (64-bit patterns) passes over the candidates.
template <class KeyF>
__device__ __forceinline__ unsigned wave_count_select(KeyF key, int m, unsigned a, unsigned range, int rank, uint32_t *__restrict__ hist, bool &exact)
{
    const int lane = threadIdx.x & 63;
    exact = false;
    while (range > 0u) {
        const int shift = range >= 256u ? 24 - __builtin_clz(range) : 0;      // range >> shift <= 255
        uint4 *hz = reinterpret_cast<uint4 *>(hist + 4 * lane);
        hz[0] = make_uint4(0u, 0u, 0u, 0u);
        wave_lds_fence();
        for (int t = lane; t < m; t += 64) {
            unsigned kv;
            if (key(t, kv) && kv - a <= range) __hip_atomic_fetch_add(hist + ((kv - a) >> shift), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        wave_lds_fence();
        const uint4 t4 = hz[0];
        const int hv[4] = { (int)t4.x, (int)t4.y, (int)t4.z, (int)t4.w };
        const int lsum = hv[0] + hv[1] + hv[2] + hv[3];
        int incl = lsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(incl, o, 64); if (lane >= o) incl += t2; }
        const int excl = incl - lsum;
        const bool mine = rank > excl && rank <= incl;
        int B = -1, cb = 0, hb = 0;
        if (mine) {
            int cum = excl;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (B < 0 && rank <= cum + hv[w]) { B = 4 * lane + w; cb = cum; hb = hv[w]; }
                cum += hv[w];
            }
        }
        const unsigned long long bm = __builtin_amdgcn_ballot_w64(mine);
        if (bm == 0ull) return a;                                   // fewer than `rank` keys in the window: the caller's count is off
        const int owner = __builtin_ctzll(bm);
        B = __shfl(B, owner, 64); cb = __shfl(cb, owner, 64); hb = __shfl(hb, owner, 64);
        const unsigned wend = a + range, bstart = a + ((unsigned)B << shift), bspan = (1u << shift) - 1u;
        const unsigned bend = wend - bstart < bspan ? wend : bstart + bspan;      // the window's end may cut its last bucket
        if (rank - cb == hb) { exact = true; return bend; }        // the rank-th is the last of its bucket
        if (shift == 0) return bstart;                              // one value wide: THE key
        rank -= cb; a = bstart; range = bend - bstart;
    }
    return a;
}
__device__ __forceinline__ unsigned wave_all_min_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)v, o, 64); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ unsigned wave_all_max_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)v, o, 64); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ int wave_all_sum_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// A threshold thr with { pattern <= thr } = the kk smallest of the m non-negative doubles vals[0 .. m) in LDS plus whatever ties with the
// kk-th (1 <= kk < m): high words first (zero distances -- the query itself, its duplicates -- are counted, not bucketed: with zero in the
// window the first level's buckets are eight binades wide), low words only among the candidates that share the kk-th high word.
__device__ __forceinline__ unsigned long long wave_kth_pattern(const double *__restrict__ vals, int m, int kk, uint32_t *__restrict__ hist)
{
    const int lane = threadIdx.x & 63;
    const uint32_t *w32 = reinterpret_cast<const uint32_t *>(vals);           // little endian: word 2 t + 1 is the high word
    unsigned vmin = 0xFFFFFFFFu, vmax = 0u;
    int zeros = 0;
    for (int t = lane; t < m; t += 64) {
        const unsigned h = w32[2 * t + 1];
        zeros += h == 0u ? 1 : 0;
        const unsigned nz = h == 0u ? 0xFFFFFFFFu : h;
        vmin = nz < vmin ? nz : vmin; vmax = h > vmax ? h : vmax;
    }
    vmin = wave_all_min_u32(vmin); vmax = wave_all_max_u32(vmax); zeros = wave_all_sum_i32(zeros);
    unsigned P = 0u;
    bool exact = false;
    if (kk > zeros) {
        P = vmin;
        if (vmin < vmax)
            P = wave_count_select([&](int t, unsigned &kv) { kv = w32[2 * t + 1]; return kv != 0u; }, m, vmin, vmax - vmin, kk - zeros, hist, exact);
        if (exact) return ((unsigned long long)P << 32) | 0xFFFFFFFFull;
    }
    int c_less = 0, c_eq = 0;
    unsigned lmin = 0xFFFFFFFFu, lmax = 0u;
    for (int t = lane; t < m; t += 64) {
        const unsigned h = w32[2 * t + 1];
        c_less += h < P ? 1 : 0;
        if (h == P) { const unsigned l = w32[2 * t]; ++c_eq; lmin = l < lmin ? l : lmin; lmax = l > lmax ? l : lmax; }
    }
    c_less = wave_all_sum_i32(c_less); c_eq = wave_all_sum_i32(c_eq);
    const int r2 = kk - c_less;                                                // rank of the kk-th among the candidates with high word P
    if (r2 >= c_eq) return ((unsigned long long)P << 32) | 0xFFFFFFFFull;      // all of them are selected
    lmin = wave_all_min_u32(lmin); lmax = wave_all_max_u32(lmax);
    unsigned Q = lmin;
    if (lmin < lmax)
        Q = wave_count_select([&](int t, unsigned &kv) { kv = w32[2 * t]; return w32[2 * t + 1] == P; }, m, lmin, lmax - lmin, r2, hist, exact);
    return ((unsigned long long)P << 32) | Q;
}
struct WaveKnnResult {
    int m;                     // candidates in vals / pos
    int kk;                    // neighbours selected: min(k, points within r2max)
    int cnt;                   // candidates with pattern <= thr (>= kk: ties at the k-th value)
    unsigned long long thr;    // bit pattern: the selected set is {d^2 pattern <= thr}
    double top;                // largest selected d^2
};
// Returns false when the query does not fit the buffer (the caller hands it to the next pass).
template <bool POS>
__device__ __forceinline__ bool wave_knn_select(const GridParams &g, const uint32_t *__restrict__ cell_start, const float *__restrict__ spts,
                                                const double q[3], int k, double r2max, const WaveKnnScratch &sc, WaveKnnResult &out)
{
    const int lane = threadIdx.x & 63;
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = cell_coord(q[a], g.org[a], g.h, g.dim[a]);
    int maxr = g.dim[0] > g.dim[1] ? g.dim[0] : g.dim[1];
    if (g.dim[2] > maxr) maxr = g.dim[2];
    bool truncated = false;
    auto gather_block = [&](int r, double tau) -> int {
        const int xa = c[0] - r < 0 ? 0 : c[0] - r, xb = c[0] + r >= g.dim[0] ? g.dim[0] - 1 : c[0] + r;
        const int ya = c[1] - r < 0 ? 0 : c[1] - r, yb = c[1] + r >= g.dim[1] ? g.dim[1] - 1 : c[1] + r;
        const int za = c[2] - r < 0 ? 0 : c[2] - r, zb = c[2] + r >= g.dim[2] ? g.dim[2] - 1 : c[2] + r;
        const int ny = yb - ya + 1, ncols = (xb - xa + 1) * ny;
        int m = 0;
        truncated = false;
        for (int c0 = 0; c0 < ncols && !truncated; c0 += 64) {
            const int nruns = ncols - c0 < 64 ? ncols - c0 : 64;
            uint32_t s0 = 0;
            int len = 0;
            if (lane < nruns) {
                const int x = xa + (c0 + lane) / ny, y = ya + (c0 + lane) % ny;
                const int64_t col = ((int64_t)x * g.dim[1] + y) * g.dim[2];
                s0 = cell_start[col + za];
                len = (int)(cell_start[col + zb + 1] - s0);
            }
            const int incl = wave_incl_scan(len);
            const int mc = __shfl(incl, 63, 64);
            if (mc == 0) continue;
            sc.run_s0[lane] = s0;
            sc.run_off[lane] = incl - len;
            wave_lds_fence();
            for (int t0 = 0; t0 < mc; t0 += 64) {
                const int t = t0 + lane;
                double d = INFINITY;
                uint32_t sp = 0;
                if (t < mc) {
                    int lo = 0, hi = nruns - 1;                                    // last run with off <= t
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (sc.run_off[mid] <= t) lo = mid; else hi = mid - 1;
                    }
                    sp = sc.run_s0[lo] + (uint32_t)(t - sc.run_off[lo]);
                    const float *pp = spts + 3 * (int64_t)sp;
                    const double dx = q[0] - (double)pp[0], dy = q[1] - (double)pp[1], dz = q[2] - (double)pp[2];
                    d = fma(dz, dz, fma(dy, dy, dx * dx));
                }
                const bool keep = t < mc && d <= tau && d < r2max;
                const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
                const int pos = m + __builtin_popcountll(km & ((1ull << lane) - 1ull));
                if (keep && pos < sc.cap) {
                    sc.vals[pos] = d;
                    if (POS) sc.pos[pos] = sp;
                }
                m += __builtin_popcountll(km);
                if (m >= sc.cap) { m = sc.cap; truncated = true; break; }
            }
            wave_lds_fence();
        }
        return m;
    };

    double tau = INFINITY;
    int r = 1;
    for (int round = 0; round <= 24; ++round) {                // safety net: past it the query goes to the next pass
        const int m = gather_block(r, tau);
        wave_lds_fence();
        const double cov2 = block_cover2(g, q, c, r);
        const bool whole = cov2 == INFINITY || cov2 >= r2max;   // the block holds everything that may be selected
        if (!truncated && m < k && !whole) {                    // too few candidates: grow by the density seen so far
            const double f = cbrt((double)(k + 1) / (double)(m > 0 ? m : 1));
            int rn = (int)((double)r * (f < 4.0 ? f : 4.0)) + 1;
            r = rn > r ? rn : r + 1;
            if (r > maxr) r = maxr;
            continue;
        }
        const int kk = m < k ? m : k;
        // the threshold of the kk smallest patterns
        unsigned long long lo = 0ull;
#ifdef KPX_KNN_BISECT
        {                                                       // (until round 5: bisection between the smallest and the largest pattern)
            unsigned long long hi = 0ull;
            lo = ~0ull;
            for (int t = lane; t < m; t += 64) {
                const unsigned long long p = (unsigned long long)__double_as_longlong(sc.vals[t]);
                lo = p < lo ? p : lo; hi = p > hi ? p : hi;
            }
            lo = wave_all_min_u64(lo); hi = wave_all_max_u64(hi);
            if (m <= k) lo = hi;                                // everything gathered is selected: no search needed
            if (m == 0) { lo = hi = 0ull; }
            while (lo < hi) {
                const unsigned long long mid = lo + ((hi - lo) >> 1);
                int cnt = 0;
                for (int t0 = 0; t0 < m; t0 += 64) {
                    const int t = t0 + lane;
                    const bool le = t < m && (unsigned long long)__double_as_longlong(sc.vals[t]) <= mid;
                    cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(le));
                }
                if (cnt == kk) { lo = hi = mid; break; }       // the set is determined
                if (cnt > kk) hi = mid; else lo = mid + 1;
            }
        }
#else
        if (m > k) lo = wave_kth_pattern(sc.vals, m, kk, sc.hist);             // by counting (above)
        else if (m > 0) {                                       // everything gathered is selected: the largest pattern
            for (int t = lane; t < m; t += 64) {
                const unsigned long long p = (unsigned long long)__double_as_longlong(sc.vals[t]);
                lo = p > lo ? p : lo;
            }
            lo = wave_all_max_u64(lo);
        }
#endif
        double top = 0.0;
        int cnt = 0;
        for (int t0 = 0; t0 < m; t0 += 64) {
            const int t = t0 + lane;
            const bool le = t < m && (unsigned long long)__double_as_longlong(sc.vals[t]) <= lo;
            if (le) top = fmax(top, sc.vals[t]);
            cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(le));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) top = fmax(top, __shfl_xor(top, o, 64));
        if (!truncated && (whole || top < cov2)) {
            out.m = m; out.kk = kk; out.cnt = cnt; out.thr = lo; out.top = top;
            return true;
        }
        // the k-th candidate found so far bounds the true k-th distance: gather again, only what lies within it, from
        // the block that covers it -- that settles the query unless even the ball overflows the buffer
        if (truncated && top >= tau) return false;
        tau = top;
        int rn = (int)(sqrt(top) / g.h) + 1;
        if (rn > maxr) rn = maxr;
        r = rn;
        wave_lds_fence();
    }
    return false;
}


// More candidates at the k-th distance than slots: the `need` lowest ORIGINAL indices among them stay (the (d^2, index)
// order of the heaps above).  Returns the largest original index that is kept among the ties (INT_MAX: all of them).
__device__ __forceinline__ int32_t wave_knn_tie_threshold(const WaveKnnScratch &sc, const WaveKnnResult &res, const int32_t *__restrict__ sidx)
{
    if (res.cnt <= res.kk) return INT_MAX;
    const int lane = threadIdx.x & 63;
    const unsigned long long topp = (unsigned long long)__double_as_longlong(res.top);
    int ntied = 0;
    for (int t0 = 0; t0 < res.m; t0 += 64) {
        const int t = t0 + lane;
        const bool tie = t < res.m && (unsigned long long)__double_as_longlong(sc.vals[t]) == topp;
        ntied += __builtin_popcountll(__builtin_amdgcn_ballot_w64(tie));
    }
    const int need = res.kk - (res.cnt - ntied);
    int32_t last = -1;
    for (int round = 0; round < need; ++round) {
        int32_t cand = INT_MAX;
        for (int t = lane; t < res.m; t += 64)
            if ((unsigned long long)__double_as_longlong(sc.vals[t]) == topp) {
                const int32_t oi = sidx[sc.pos[t]];
                if (oi > last && oi < cand) cand = oi;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int32_t t2 = __shfl_xor(cand, o, 64); cand = t2 < cand ? t2 : cand; }
        last = cand;
    }
    return last;
}
// is candidate t one of the kk selected neighbours?
__device__ __forceinline__ bool wave_knn_is_selected(const WaveKnnScratch &sc, const WaveKnnResult &res, const int32_t *__restrict__ sidx,
                                                     int32_t idx_thr, int t)
{
    const unsigned long long p = (unsigned long long)__double_as_longlong(sc.vals[t]);
    const unsigned long long topp = (unsigned long long)__double_as_longlong(res.top);
    if (p > res.thr) return false;
    if (p < topp || res.cnt == res.kk) return true;
    return p == topp && sidx[sc.pos[t]] <= idx_thr;
}

}  // namespace kpx
