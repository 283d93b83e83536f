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
// The same for samples of at most kHypLdsN points (floor_removal.py:70 draws 30): the sample's indices and its points live in LDS
// (slot t of thread l at [t][l]: conflict-free) -- the kernel above keeps the indices in global memory, re-reads them for every duplicate
// check (435 dependent loads for a sample of 30) and gathers every point twice (centroid, then moments): 52 us for 2000 hypotheses, a
// sixth of the whole call on the floor slab.  Same draws, same sums in the same order: the same planes bit for bit.
constexpr int kHypLdsN = 48;
__global__ __launch_bounds__(64) void plane_hyp_lds_kernel(const float *__restrict__ pts, int64_t n, int ransac_n, int H, uint32_t seed_lo,
                                                           uint32_t seed_hi, double *__restrict__ hyp)
{
    extern __shared__ __align__(16) int32_t hyp_lds[];
    int32_t *sid = hyp_lds;                                        // [ransac_n][64]
    float *sp = reinterpret_cast<float *>(hyp_lds + (size_t)ransac_n * 64);   // [ransac_n][3][64]
    const int l = threadIdx.x;
    const int h = blockIdx.x * 64 + l;
    if (h >= H) return;
    int got = 0;
    for (uint32_t blk = 0; got < ransac_n; ++blk) {
        uint32_t out[4];
        philox4x32(blk, (uint32_t)h, 0u, 0u, seed_lo, seed_hi, out);
        for (int w = 0; w < 4 && got < ransac_n; ++w) {
            const int32_t id = (int32_t)(((uint64_t)out[w] * (uint64_t)n) >> 32);
            bool dup = false;
            for (int t = 0; t < got; ++t) dup |= (sid[t * 64 + l] == id);
            if (!dup) sid[(got++) * 64 + l] = id;
        }
    }
    for (int t0 = 0; t0 < ransac_n; t0 += 8) {                    // the sample's points, eight gathers in flight
        float v[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t j = t0 + u < ransac_n ? sid[(t0 + u) * 64 + l] : 0;
            v[u][0] = pts[3 * j]; v[u][1] = pts[3 * j + 1]; v[u][2] = pts[3 * j + 2];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (t0 + u < ransac_n) { sp[((t0 + u) * 3 + 0) * 64 + l] = v[u][0]; sp[((t0 + u) * 3 + 1) * 64 + l] = v[u][1]; sp[((t0 + u) * 3 + 2) * 64 + l] = v[u][2]; }
    }
    auto P = [&](int t, int a) { return (double)sp[(t * 3 + a) * 64 + l]; };
    double pl[4] = { 0.0, 0.0, 0.0, 0.0 };
    const int m = ransac_n;
    if (m == 3) {
        double e0[3], e1[3];
        for (int a = 0; a < 3; ++a) { e0[a] = P(1, a) - P(0, a); e1[a] = P(2, a) - P(0, a); }
        double a = e0[1] * e1[2] - e0[2] * e1[1], b = e0[2] * e1[0] - e0[0] * e1[2], c = e0[0] * e1[1] - e0[1] * e1[0];
        const double nn = sqrt(a * a + b * b + c * c);
        if (nn != 0.0) {
            a /= nn; b /= nn; c /= nn;
            pl[0] = a; pl[1] = b; pl[2] = c; pl[3] = -(a * P(0, 0) + b * P(0, 1) + c * P(0, 2));
        }
    } else {
        double cx = 0, cy = 0, cz = 0;
        for (int t = 0; t < m; ++t) { cx += P(t, 0); cy += P(t, 1); cz += P(t, 2); }
        cx /= (double)m; cy /= (double)m; cz /= (double)m;
        double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
        for (int t = 0; t < m; ++t) {
            const double rx = P(t, 0) - cx, ry = P(t, 1) - cy, rz = P(t, 2) - cz;
            xx += rx * rx; xy += rx * ry; xz += rx * rz; yy += ry * ry; yz += ry * rz; zz += rz * rz;
        }
        const double det_x = yy * zz - yz * yz, det_y = xx * zz - xz * xz, det_z = xx * yy - xy * xy;
        double a, b, c;
        if (det_x > det_y && det_x > det_z) { a = det_x; b = xz * yz - xy * zz; c = xy * yz - xz * yy; }
        else if (det_y > det_z)             { a = xz * yz - xy * zz; b = det_y; c = xy * xz - yz * xx; }
        else                                { a = xy * yz - xz * yy; b = xy * xz - yz * xx; c = det_z; }
        const double nn = sqrt(a * a + b * b + c * c);
        if (nn != 0.0) {
            a /= nn; b /= nn; c /= nn;
            pl[0] = a; pl[1] = b; pl[2] = c; pl[3] = -(a * cx + b * cy + c * cz);
        }
    }
    hyp[4 * h] = pl[0]; hyp[4 * h + 1] = pl[1]; hyp[4 * h + 2] = pl[2]; hyp[4 * h + 3] = pl[3];
}

constexpr int kScoreThreads = 256;

__global__ __launch_bounds__(kScoreThreads) void plane_score_kernel(const float *__restrict__ pts, int64_t n, int64_t chunk,
                                                                    const double *__restrict__ hyp, int H, double thr,
                                                                    uint32_t *__restrict__ part_cnt, double *__restrict__ part_err,
                                                                    const int32_t *__restrict__ gate)
{
    if (gate && !*gate) return;                     // (matrix-core path: the full scoring only runs when too many hypotheses tie)
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


// ---- scoring on the matrix cores --------------------------------------------------------------------------------------
// The signed distance of point p to hypothesis h, fma(a,x, fma(b,y, fma(c,z,d))), is one row of a K = 4 GEMM: v_mfma_f64_16x16x4_f64
// accumulates k = 0..3 in order onto its C operand and is bit for bit that fma chain, so with A = (c, b, a, 0) per
// hypothesis, B = (z, y, x, 0) per point and C = d the matrix pipe produces exactly the contract's value for 16 hypotheses x 16
// points per instruction.  What is left for the vector unit is the inlier COUNT: |t| < thr on the bit patterns (positive doubles
// order like unsigned integers) -- high words first (one v_and, one v_cmp, one v_addc per result register), the low words only in
// the rare tile where some high word equals the threshold's.  The error sums sum|d| (tie-breaker of Open3D's better-than test) are
// NOT formed here: plane_need_kernel marks the hypotheses whose rmse the sequential replay can ask for (count equal to the running
// maximum at its position) and the sequential-order kernel below evaluates those alone, so every sum keeps its defined order.
typedef double pd4 __attribute__((ext_vector_type(4)));
#ifndef KPX_PLANE_TILES
#define KPX_PLANE_TILES 4
#endif
constexpr int kPcTiles = KPX_PLANE_TILES;        // hypothesis tiles (16 each) per wave: 64 kPcTiles hypotheses per block of four waves
__device__ __forceinline__ uint32_t pc_hi(double v) { return (uint32_t)((uint64_t)__double_as_longlong(v) >> 32); }
__device__ __forceinline__ uint32_t pc_lo(double v) { return (uint32_t)(uint64_t)__double_as_longlong(v); }

constexpr int kPcSlab = 2048;                    // points staged in LDS per trip (24 KiB as floats); every wave of the block sweeps them
__global__ __launch_bounds__(256, KPX_PLANE_TILES <= 4 ? 4 : 2) void plane_count_mfma_kernel(const float *__restrict__ pts, int64_t n, int64_t chunk, const double *__restrict__ hyp,
                                                               int H, uint32_t thr_hi, uint32_t thr_lo, uint32_t *__restrict__ part_cnt)
{
    __shared__ float sp[kPcSlab * 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
    const int h0 = (blockIdx.x * 4 + wave) * (16 * kPcTiles);
    const int64_t i0 = (int64_t)blockIdx.y * chunk;
    const int64_t i1 = i0 + chunk < n ? i0 + chunk : n;
    // A: lane (k = q, row = j) holds component k of hypothesis row in the order (c, b, a, 0); C: d of the lane's four D rows
    double a[kPcTiles];
    pd4 c[kPcTiles];
    uint32_t cnt[kPcTiles][4];
#pragma unroll
    for (int t = 0; t < kPcTiles; ++t) {
        const int h = h0 + 16 * t + j;
        a[t] = (q < 3 && h < H) ? hyp[4 * h + (2 - q)] : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hr = h0 + 16 * t + q + 4 * r;
            c[t][r] = hr < H ? hyp[4 * hr + 3] : 0.0;
            cnt[t][r] = 0u;
        }
    }
    const float nanf32 = __int_as_float(0x7FC00000);          // points past the end: NaN distances, never inliers
    const float thr_f = __int_as_float((int)thr_hi);
    for (int64_t s0 = i0; s0 < i1; s0 += kPcSlab) {
        const int m = (int)(i1 - s0 < kPcSlab ? i1 - s0 : kPcSlab);
        __syncthreads();                                       // the previous slab has been swept by every wave
        for (int e = threadIdx.x; e < kPcSlab * 3; e += 256) sp[e] = e < 3 * m ? pts[3 * s0 + e] : nanf32;
        __syncthreads();
        const int tiles = (m + 15) >> 4;
        const float *lp = sp + 3 * j + (2 - q);                // component (z, y, x) of point j of a tile; lanes q == 3 feed zeros
        for (int tl = 0; tl < tiles; ++tl) {
            const double b = q < 3 ? (double)lp[48 * tl] : 0.0;
            pd4 d[kPcTiles];
#pragma unroll
            for (int t = 0; t < kPcTiles; ++t) d[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b, c[t], 0, 0, 0);
            // |t| < thr on the HIGH WORDS, read as float32 patterns: for non-negative doubles the high words order like the values, and
            // so do they as float32 bit patterns (sign in the same place, NaN / infinity patterns = huge doubles: never below thr) --
            // which makes the absolute value a free source modifier of v_cmp_lt_f32: three vector instructions per result register
            // (compare, add-with-carry, equality probe).  The host falls back to the sequential kernel for thresholds whose high word
            // is not a normal float32 pattern.
            unsigned long long amb = 0ull;                     // (ballots OR-ed on the scalar unit: one v_cmp + one s_or per register)
#pragma unroll
            for (int t = 0; t < kPcTiles; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hf = __builtin_fabsf(__int_as_float((int)pc_hi(d[t][r])));
                    cnt[t][r] += hf < thr_f ? 1u : 0u;
                    amb |= __builtin_amdgcn_ballot_w64(hf == thr_f);
                }
            if (amb != 0ull) {                                 // some |t| shares the threshold's high word: the low words decide
#pragma unroll
                for (int t = 0; t < kPcTiles; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cnt[t][r] += ((pc_hi(d[t][r]) & 0x7FFFFFFFu) == thr_hi && pc_lo(d[t][r]) < thr_lo) ? 1u : 0u;
            }
        }
    }
    // counts of the 16 point columns of every hypothesis row -> lane j == 0 of its quad row
#pragma unroll
    for (int t = 0; t < kPcTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uint32_t v = cnt[t][r];
#pragma unroll
            for (int msk = 1; msk < 16; msk <<= 1) v += __shfl_xor(v, msk, 64);
            const int hr = h0 + 16 * t + q + 4 * r;
            if (j == 0 && hr < H) part_cnt[(int64_t)blockIdx.y * H + hr] = v;
        }
}

__global__ __launch_bounds__(256) void plane_reduce_cnt_kernel(const uint32_t *__restrict__ part_cnt, int chunks, int H, int64_t *__restrict__ cnt)
{
    int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    int64_t c = 0;
    for (int k0 = 0; k0 < chunks; k0 += 16) {                     // integers: 16 independent loads per trip
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = k0 + u < chunks ? part_cnt[(int64_t)(k0 + u) * H + h] : 0u;
#pragma unroll
        for (int u = 0; u < 16; ++u) c += v[u];
    }
    cnt[h] = c;
}

// Which hypotheses can the sequential replay (plane_select_kernel) ask an rmse of?  Only those whose inlier count equals the largest
// count seen up to and including their own position: a record needs its rmse stored, a tie needs it compared.  list[0 .. *n_list) =
// those hypotheses in ascending order (at most kPlaneNeedMax; more -- e.g. every hypothesis ties on an exactly planar cloud -- sets
// *need_all and the full sequential scoring runs instead).
constexpr int kPlaneNeedMax = 256;
__global__ __launch_bounds__(1024) void plane_need_kernel(const double *__restrict__ hyp, const int64_t *__restrict__ cnt, int H, int32_t *__restrict__ list,
                                                          int32_t *__restrict__ n_list, int32_t *__restrict__ need_all)
{
    // thread t owns hypotheses [t per, (t + 1) per): segment maxima -> exclusive max-scan over the threads -> every thread re-walks its
    // segment with the running maximum in front of it -> positions by an exclusive sum-scan
    __shared__ long long smax[1024];
    __shared__ int sh[20];
    const int per = (H + 1023) / 1024, t = threadIdx.x;
    const int h_lo = t * per, h_hi = h_lo + per < H ? h_lo + per : H;
    auto live = [&](int h) { return !(hyp[4 * h] == 0.0 && hyp[4 * h + 1] == 0.0 && hyp[4 * h + 2] == 0.0 && hyp[4 * h + 3] == 0.0); };
    long long mx = -1;
    for (int h = h_lo; h < h_hi; ++h)
        if (live(h)) mx = cnt[h] > mx ? cnt[h] : mx;
    smax[t] = mx;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                // inclusive max-scan (Hillis-Steele)
        const long long o = t >= off ? smax[t - off] : -1;
        __syncthreads();
        smax[t] = o > smax[t] ? o : smax[t];
        __syncthreads();
    }
    long long run = t ? smax[t - 1] : -1;
    run = run < 0 ? 0 : run;                                   // the replay starts from best = 0
    int mine = 0;
    for (int h = h_lo; h < h_hi; ++h)
        if (live(h) && cnt[h] >= run) { run = cnt[h]; ++mine; }
    int total;
    int pos = block_excl_scan(mine, sh, &total);
    run = t ? smax[t - 1] : -1;
    run = run < 0 ? 0 : run;
    for (int h = h_lo; h < h_hi; ++h)
        if (live(h) && cnt[h] >= run) {
            run = cnt[h];
            if (pos < kPlaneNeedMax) list[pos] = h;
            ++pos;
        }
    if (t == 0) {
        *n_list = total < kPlaneNeedMax ? total : kPlaneNeedMax;
        *need_all = total > kPlaneNeedMax ? 1 : 0;
    }
}

// plane_score_kernel's sums for the listed hypotheses only (same chunks, same order inside a chunk: the same doubles)
__global__ __launch_bounds__(kPlaneNeedMax) void plane_err_list_kernel(const float *__restrict__ pts, int64_t n, int64_t chunk, const double *__restrict__ hyp,
                                                                       const int32_t *__restrict__ list, const int32_t *__restrict__ n_list,
                                                                       const int32_t *__restrict__ need_all, double thr, double *__restrict__ part_err)
{
    if (*need_all) return;
    __shared__ float sp[kPcSlab * 3];                             // the chunk's points, a slab at a time: the sequential walk reads LDS
    const int m = *n_list;
    const bool active = (int)(threadIdx.x & ~63u) < m;            // whole waves beyond the list only help with the loads
    const int slot = (int)threadIdx.x < m ? (int)threadIdx.x : (m > 0 ? m - 1 : 0);
    const int h = m > 0 ? list[slot] : 0;
    const double a = hyp[4 * h], b = hyp[4 * h + 1], c = hyp[4 * h + 2], d = hyp[4 * h + 3];
    const int64_t i0 = (int64_t)blockIdx.x * chunk;
    const int64_t i1 = i0 + chunk < n ? i0 + chunk : n;
    double err = 0.0;
    for (int64_t s0 = i0; s0 < i1; s0 += kPcSlab) {
        const int cnt = (int)(i1 - s0 < kPcSlab ? i1 - s0 : kPcSlab);
        __syncthreads();
        for (int e = threadIdx.x; e < 3 * cnt; e += kPlaneNeedMax) sp[e] = pts[3 * s0 + e];
        __syncthreads();
        if (active)
            for (int i = 0; i < cnt; i += 8) {                    // ascending point index, as plane_score_kernel: the same doubles (eight
                double dist[8];                                   // independent distances first, then the adds in order)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int iu = i + u < cnt ? i + u : cnt - 1;
                    const double x = (double)sp[3 * iu], y = (double)sp[3 * iu + 1], z = (double)sp[3 * iu + 2];
                    dist[u] = fabs(fma(a, x, fma(b, y, fma(c, z, d))));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) err += (i + u < cnt && dist[u] < thr) ? dist[u] : 0.0;
            }
    }
    if ((int)threadIdx.x < m) part_err[(int64_t)blockIdx.x * kPlaneNeedMax + threadIdx.x] = err;
}
__global__ __launch_bounds__(kPlaneNeedMax) void plane_err_list_reduce_kernel(const double *__restrict__ part_err, int chunks, const int32_t *__restrict__ list,
                                                                              const int32_t *__restrict__ n_list, const int32_t *__restrict__ need_all,
                                                                              double *__restrict__ err)
{
    if (*need_all || (int)threadIdx.x >= *n_list) return;
    double e = 0.0;
    for (int k0 = 0; k0 < chunks; k0 += 16) {                     // 16 loads in flight, added in chunk order
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = k0 + u < chunks ? part_err[(int64_t)(k0 + u) * kPlaneNeedMax + threadIdx.x] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) if (k0 + u < chunks) e += v[u];
    }
    err[list[threadIdx.x]] = e;
}

__global__ __launch_bounds__(256) void plane_reduce_kernel(const uint32_t *__restrict__ part_cnt, const double *__restrict__ part_err,
                                                           int chunks, int H, int64_t *__restrict__ cnt, double *__restrict__ err,
                                                           const int32_t *__restrict__ gate)
{
    if (gate && !*gate) return;
    int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    int64_t c = 0; double e = 0.0;
    for (int k = 0; k < chunks; ++k) { c += part_cnt[(int64_t)k * H + h]; e += part_err[(int64_t)k * H + h]; }
    cnt[h] = c; err[h] = e;
}

// The sequential loop of [O3D] SegmentPlane (better-than test + probabilistic early exit), evaluated in parallel -- same result:
//   * a hypothesis replaces the best when its fitness is larger, or equal with a smaller rmse, so the survivor of iterations
//     [0, stop) is the first one in the order (fitness descending, rmse ascending, iteration ascending) among the non-degenerate
//     hypotheses with fitness > 0;
//   * the early-exit bound only changes when the best FITNESS changes and is a function of it alone, so at iteration `it` it is
//     f(largest fitness among the iterations before it) (infinite while that is 0), and the loop stops at the first it > bound.
// (One thread replaying 2000 hypotheses from LDS took 0.3-0.6 ms when no early exit cuts it short -- probability = 1.)
__device__ __forceinline__ double plane_break_bound(double fit, int ransac_n, double probability, int H)
{
    if (!(fit > 0.0)) return INFINITY;
    if (!(fit < 1.0)) return 0.0;
    double bi = log(1.0 - probability) / log(1.0 - pow(fit, (double)ransac_n));
    if (!(bi >= 0.0)) bi = (double)H;          // fitness^n vanishes against 1: -inf (NaN for probability 1) -- [O3D]'s size_t takes 2^63 there: no exit
    bi = bi < (double)H ? bi : (double)H;
    return floor(bi);
}
__global__ __launch_bounds__(1024) void plane_select_kernel(const double *__restrict__ hyp, const int64_t *__restrict__ cnt, const double *__restrict__ err,
                                                            int H, int64_t n, int ransac_n, double probability, double *__restrict__ best)
{
    __shared__ double smax[1024], s_fit[1024], s_rmse[1024];
    __shared__ int s_idx[1024];
    __shared__ int s_stop;
    const int t = threadIdx.x, per = (H + 1023) / 1024;
    const int h_lo = t * per < H ? t * per : H, h_hi = h_lo + per < H ? h_lo + per : H;
    auto live = [&](int h) { return !(hyp[4 * h] == 0.0 && hyp[4 * h + 1] == 0.0 && hyp[4 * h + 2] == 0.0 && hyp[4 * h + 3] == 0.0); };
    auto fitness = [&](int h) { const int64_t k = cnt[h]; return k ? (double)k / (double)n : 0.0; };
    if (t == 0) s_stop = H;
    double mx = 0.0;
    for (int h = h_lo; h < h_hi; ++h)
        if (live(h)) mx = fmax(mx, fitness(h));
    smax[t] = mx;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                // inclusive max-scan over the threads' segments
        const double o = t >= off ? smax[t - off] : 0.0;
        __syncthreads();
        smax[t] = fmax(smax[t], o);
        __syncthreads();
    }
    // first iteration the loop does not enter
    double run = t ? smax[t - 1] : 0.0;
    for (int h = h_lo; h < h_hi; ++h) {
        if ((double)h > plane_break_bound(run, ransac_n, probability, H)) { atomicMin(&s_stop, h); break; }
        if (live(h)) run = fmax(run, fitness(h));
    }
    __syncthreads();
    const int stop = s_stop;
    // the survivor among [0, stop)
    double bf = 0.0, br = 0.0;
    int bi = INT_MAX;
    for (int h = h_lo; h < h_hi && h < stop; ++h) {
        if (!live(h)) continue;
        const int64_t k = cnt[h];
        const double fit = k ? (double)k / (double)n : 0.0;
        if (!(fit > 0.0)) continue;
        const double rmse = err[h] / sqrt((double)k);
        if (bi == INT_MAX || fit > bf || (fit == bf && rmse < br)) { bf = fit; br = rmse; bi = h; }
    }
    s_fit[t] = bf; s_rmse[t] = br; s_idx[t] = bi;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (t < off) {
            const double of = s_fit[t + off], orr = s_rmse[t + off];
            const int oi = s_idx[t + off];
            const bool take = oi != INT_MAX && (s_idx[t] == INT_MAX || of > s_fit[t] || (of == s_fit[t] && (orr < s_rmse[t] || (orr == s_rmse[t] && oi < s_idx[t]))));
            if (take) { s_fit[t] = of; s_rmse[t] = orr; s_idx[t] = oi; }
        }
        __syncthreads();
    }
    if (t < 4) best[t] = s_idx[0] != INT_MAX ? hyp[4 * s_idx[0] + t] : 0.0;
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
__global__ __launch_bounds__(256) void plane_refit_final_kernel(const double *__restrict__ part, int nb, int pass, double *__restrict__ cen,
                                                                double *__restrict__ plane)
{
    __shared__ double sp[1024 * 6];                               // nb <= 1024 block partials, loaded by the block, folded in block order by one thread
    for (int e = threadIdx.x; e < nb * 6; e += blockDim.x) sp[e] = part[e];
    __syncthreads();
    if (threadIdx.x) return;
    double s[6] = { 0, 0, 0, 0, 0, 0 };
    for (int b = 0; b < nb; ++b) for (int q = 0; q < 6; ++q) s[q] += sp[b * 6 + q];
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
    int32_t *need_list = a.get<int32_t>(kPlaneNeedMax + 2);
    double *part_err_list = a.get<double>((size_t)chunks * kPlaneNeedMax);
    if (a.dry) return KPX_OK;
    KPX_ARENA_CHECK(a);
    static const bool mfma_on = [] { const char *e = getenv("KPX_PLANE_MFMA"); return !(e && e[0] == '0'); }();       // A/B switch
    if (H > 0) {
        if (ransac_n <= kHypLdsN)
            hipLaunchKernelGGL(plane_hyp_lds_kernel, dim3((unsigned)cdiv(H, 64)), dim3(64), (size_t)ransac_n * 64 * 16, st, pts, n, ransac_n, H, (uint32_t)seed,
                               (uint32_t)(seed >> 32), hyp);
        else
            hipLaunchKernelGGL(plane_hyp_kernel, dim3((unsigned)cdiv(H, 64)), dim3(64), 0, st, pts, n, ransac_n, H, (uint32_t)seed,
                               (uint32_t)(seed >> 32), ids, hyp);
        const dim3 grid((unsigned)cdiv(H, kScoreThreads), (unsigned)chunks);
        uint64_t tbits;
        memcpy(&tbits, &thr, sizeof(tbits));
        const uint32_t thi = (uint32_t)(tbits >> 32);
        // (thresholds whose high word is not a normal, positive float32 pattern -- below 2^-1015, above 2^1017, negative, NaN -- take the
        // sequential kernel)
        if (mfma_on && thi >= 0x00800000u && thi < 0x7F800000u) {
            // counts on the matrix cores; rmse only for the hypotheses the replay can ask about; the full sequential scoring only when
            // more than kPlaneNeedMax hypotheses tie with the running best (its blocks return at once otherwise)
            const uint64_t tb = tbits;
            int32_t *n_list = need_list + kPlaneNeedMax, *need_all = need_list + kPlaneNeedMax + 1;
            {
                ProfScope prof(KPX_PROF_PLANE_SCORE, 12.0 * (double)n, st);  // one algorithmic sweep of the points for all H
                hipLaunchKernelGGL(plane_count_mfma_kernel, dim3((unsigned)cdiv(H, 64 * kPcTiles), (unsigned)chunks), dim3(256), 0, st, pts, n, chunk, hyp, H, (uint32_t)(tb >> 32), (uint32_t)tb, part_cnt);
            }
            hipLaunchKernelGGL(plane_reduce_cnt_kernel, dim3((unsigned)cdiv(H, 256)), dim3(256), 0, st, part_cnt, chunks, H, cnt);
            hipLaunchKernelGGL(plane_need_kernel, dim3(1), dim3(1024), 0, st, hyp, cnt, H, need_list, n_list, need_all);
            hipLaunchKernelGGL(plane_err_list_kernel, dim3((unsigned)chunks), dim3(kPlaneNeedMax), 0, st, pts, n, chunk, hyp, need_list, n_list, need_all, thr,
                               part_err_list);
            hipLaunchKernelGGL(plane_err_list_reduce_kernel, dim3(1), dim3(kPlaneNeedMax), 0, st, part_err_list, chunks, need_list, n_list, need_all, err);
            hipLaunchKernelGGL(plane_score_kernel, grid, dim3(kScoreThreads), 0, st, pts, n, chunk, hyp, H, thr, part_cnt, part_err, need_all);
            hipLaunchKernelGGL(plane_reduce_kernel, dim3((unsigned)cdiv(H, 256)), dim3(256), 0, st, part_cnt, part_err, chunks, H, cnt, err, need_all);
        } else {
            {
                ProfScope prof(KPX_PROF_PLANE_SCORE, 12.0 * (double)n, st);
                hipLaunchKernelGGL(plane_score_kernel, grid, dim3(kScoreThreads), 0, st, pts, n, chunk, hyp, H, thr, part_cnt, part_err, (const int32_t *)nullptr);
            }
            hipLaunchKernelGGL(plane_reduce_kernel, dim3((unsigned)cdiv(H, 256)), dim3(256), 0, st, part_cnt, part_err, chunks, H, cnt, err, (const int32_t *)nullptr);
        }
    }
    hipLaunchKernelGGL(plane_select_kernel, dim3(1), dim3(1024), 0, st, hyp, cnt, err, H, n, ransac_n, probability, best);
    int rc = compact(PlaneInlierPred{ pts, best, thr }, PlaneIdxEmit{ inl_idx }, n, 1, counts, d_count, st);
    if (rc) return rc;
    int nb = (int)(cdiv(n, 256 * 8) < 1 ? 1 : (cdiv(n, 256 * 8) > 1024 ? 1024 : cdiv(n, 256 * 8)));
    double *cen = best + 4;
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(plane_refit_sum_kernel, dim3(nb), dim3(256), 0, st, pts, n, best, thr, cen, pass, rpart);
        hipLaunchKernelGGL(plane_refit_final_kernel, dim3(1), dim3(256), 0, st, rpart, nb, pass, cen, d_plane);
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
