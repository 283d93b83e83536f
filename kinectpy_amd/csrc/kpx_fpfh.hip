// kpx_fpfh.hip -- global registration (reference rows a11-a13):
//   compute_fpfh_feature(KDTreeSearchParamHybrid)                        preprocessing/registration.py:15-20
//   registration_ransac_based_on_feature_matching(..., mutual_filter,    preprocessing/registration.py:50-57
//       PointToPoint(False), 3, [EdgeLength(0.95), Distance(thr)], RANSACConvergenceCriteria(250000, 0.999))
// Kernels: hybrid neighbour lists (exact grid search, ascending (d2, idx)), SPFH histograms, FPFH weighting,
// 33-D feature nearest neighbour (LDS-tiled fp64, the oracle's accumulation order), batched RANSAC hypotheses
// (Philox sampling, 3-point Umeyama, edge-length + distance checkers), validation by an exact radius-limited
// nearest-neighbour count on the target grid, inlier ratio of the correspondence set.  The sequential
// better-than / est_k logic of Open3D is replayed on the host in iteration order.
#include <vector>

#include "kpx_gridknn.h"
#include "kpx_linalg.h"

namespace kpx {

constexpr double kSentinelF = 1e300;

// ---- neighbour lists ------------------------------------------------------------------------------------------
// Wave per query (wave_knn_select, kpx_gridknn.h): the selected neighbours are written in candidate order, except that
// slot 0 receives the smallest (d^2, index) -- the entry the feature kernels skip as "the point itself".  The feature
// sums do not depend on the order of the other slots beyond rounding.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void nbr_list_wave_kernel(const GridParams *__restrict__ gp, const uint32_t *__restrict__ cell_start,
                                                                   const float *__restrict__ spts, const int32_t *__restrict__ sidx,
                                                                   int64_t n, int k, int cap, double r2, int32_t *__restrict__ nbr,
                                                                   double *__restrict__ d2, int32_t *__restrict__ cnt,
                                                                   int32_t *__restrict__ fb_list, int32_t *__restrict__ fb_count)
{
    extern __shared__ __align__(16) double lds[];
    __shared__ uint32_t run_s0[WAVES][64];
    __shared__ int32_t run_off[WAVES][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t *posbase = reinterpret_cast<uint32_t *>(lds + (size_t)WAVES * cap);
    __shared__ __align__(16) uint32_t knn_hist[WAVES][kKnnBuckets];
    const WaveKnnScratch sc{ lds + (size_t)wave * cap, posbase + (size_t)wave * cap, run_s0[wave], run_off[wave], cap, knn_hist[wave] };
    const GridParams g = *gp;
    for (int64_t s = (int64_t)blockIdx.x * WAVES + wave; s < n; s += (int64_t)gridDim.x * WAVES) {
        const double q[3] = { (double)spts[3 * s], (double)spts[3 * s + 1], (double)spts[3 * s + 2] };
        const int64_t me = sidx[s];
        WaveKnnResult res;
        if (!wave_knn_select<true>(g, cell_start, spts, q, k, r2, sc, res)) {
            if (lane == 0) fb_list[atomicAdd(fb_count, 1)] = (int32_t)s;
            continue;
        }
        const int32_t idx_thr = wave_knn_tie_threshold(sc, res, sidx);
        int base = 0;
        double bd = INFINITY;                                  // smallest (d^2, index) written by this lane, and its slot
        int32_t bi = INT_MAX, bslot = -1;
        for (int t0 = 0; t0 < res.m; t0 += 64) {
            const int t = t0 + lane;
            const bool sel = t < res.m && wave_knn_is_selected(sc, res, sidx, idx_thr, t);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(sel);
            if (sel) {
                const int slot = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                const double d = sc.vals[t];
                const int32_t oi = sidx[sc.pos[t]];
                nbr[me * k + slot] = oi;
                d2[me * k + slot] = d;
                if (d < bd || (d == bd && oi < bi)) { bd = d; bi = oi; bslot = slot; }
            }
            base += __builtin_popcountll(m);
        }
        // wave-wide minimum -> slot 0
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double od = __shfl_xor(bd, o, 64);
            const int32_t oi = __shfl_xor(bi, o, 64), os = __shfl_xor(bslot, o, 64);
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; bslot = os; }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (lane == 0) {
            cnt[me] = res.kk;
            if (bslot > 0) {                                   // swap the minimum into slot 0
                const int32_t i0 = nbr[me * k]; const double d0 = d2[me * k];
                nbr[me * k] = bi; d2[me * k] = bd;
                nbr[me * k + bslot] = i0; d2[me * k + bslot] = d0;
            }
        }
        for (int t = res.kk + lane; t < k; t += 64) { nbr[me * k + t] = -1; d2[me * k + t] = 0.0; }
        wave_lds_fence();
    }
}

// Thread per query with a (d^2, index) heap; list != NULL: only the queries the wave kernel could not hold.
__global__ void nbr_list_kernel(const GridParams *__restrict__ gp, const uint32_t *__restrict__ cell_start,
                                const float *__restrict__ spts, const int32_t *__restrict__ sidx, int64_t n, int k, double r2,
                                int32_t *__restrict__ nbr, double *__restrict__ d2, int32_t *__restrict__ cnt,
                                const int32_t *__restrict__ list, const int32_t *__restrict__ list_count, double *__restrict__ gheap, int32_t *__restrict__ gix)
{
    // gheap / gix != NULL (max_nn beyond what LDS holds): the thread's (d^2, index) heap lives in the workspace
    extern __shared__ __align__(16) double lds[];
    const int64_t total = list ? (int64_t)*list_count : n;
    const GridParams g = *gp;
    int32_t *ilds = reinterpret_cast<int32_t *>(lds + (size_t)k * blockDim.x);
    const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = list ? (int64_t)list[e] : e;
        HeapDI heap{ gheap ? gheap + gt : lds + threadIdx.x, gheap ? gix + gt : ilds + threadIdx.x, gheap ? (int)(gridDim.x * blockDim.x) : (int)blockDim.x, k, 0 };
        grid_knn_scan(g, cell_start, spts, sidx, (double)spts[3 * s], (double)spts[3 * s + 1], (double)spts[3 * s + 2], r2, heap);
        const int64_t me = sidx[s];
        const int m = heap.sz;
        cnt[me] = m;
        // heap-sort extraction: the maximum goes to the last free slot -> ascending (d2, idx)
        for (int last = m - 1; last >= 0; --last) {
            const double dm = heap.h[0]; const int32_t im = heap.ix[0];
            const double dv = heap.h[last * heap.stride]; const int32_t iv = heap.ix[last * heap.stride];
            int c = 0;
            for (;;) {
                int l = 2 * c + 1, r = l + 1;
                if (l >= last) break;
                int b = l; double hb = heap.h[l * heap.stride]; int32_t ib = heap.ix[l * heap.stride];
                if (r < last) { double hr = heap.h[r * heap.stride]; int32_t ir = heap.ix[r * heap.stride]; if (HeapDI::less(hb, ib, hr, ir)) { b = r; hb = hr; ib = ir; } }
                if (HeapDI::less(dv, iv, hb, ib)) { heap.h[c * heap.stride] = hb; heap.ix[c * heap.stride] = ib; c = b; } else break;
            }
            if (last > 0) { heap.h[c * heap.stride] = dv; heap.ix[c * heap.stride] = iv; }
            nbr[me * k + last] = im;
            d2[me * k + last] = dm;
        }
        for (int t = m; t < k; ++t) { nbr[me * k + t] = -1; d2[me * k + t] = 0.0; }
    }
}

// ---- SPFH / FPFH ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int bin11(double x)
{
    int h = (int)floor(x);
    return h < 0 ? 0 : (h >= 11 ? 10 : h);
}
__device__ __forceinline__ void pair_features(const double p1[3], const double n1[3], const double p2[3], const double n2[3], double f[4])
{
    double dp[3] = { p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2] };
    f[0] = f[1] = f[2] = 0.0;
    f[3] = sqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2]);
    if (f[3] == 0.0) return;
    double a[3] = { n1[0], n1[1], n1[2] }, b[3] = { n2[0], n2[1], n2[2] };
    const double angle1 = (a[0] * dp[0] + a[1] * dp[1] + a[2] * dp[2]) / f[3];
    const double angle2 = (b[0] * dp[0] + b[1] * dp[1] + b[2] * dp[2]) / f[3];
    if (fabs(angle1) < fabs(angle2)) {            // acos|a1| > acos|a2|
        for (int k = 0; k < 3; ++k) { a[k] = n2[k]; b[k] = n1[k]; dp[k] = -dp[k]; }
        f[2] = -angle2;
    } else f[2] = angle1;
    double v[3] = { dp[1] * a[2] - dp[2] * a[1], dp[2] * a[0] - dp[0] * a[2], dp[0] * a[1] - dp[1] * a[0] };
    const double vn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (vn == 0.0) { f[0] = f[1] = f[2] = f[3] = 0.0; return; }
    v[0] /= vn; v[1] /= vn; v[2] /= vn;
    const double w[3] = { a[1] * v[2] - a[2] * v[1], a[2] * v[0] - a[0] * v[2], a[0] * v[1] - a[1] * v[0] };
    f[1] = v[0] * b[0] + v[1] * b[1] + v[2] * b[2];
    f[0] = atan2(w[0] * b[0] + w[1] * b[1] + w[2] * b[2], a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
}

constexpr int kFeatThreads = 128;
__global__ __launch_bounds__(kFeatThreads) void spfh_kernel(const float *__restrict__ pts, const float *__restrict__ nrm, int64_t n,
                                                            const int32_t *__restrict__ nbr, const int32_t *__restrict__ cnt, int k,
                                                            double *__restrict__ spfh)
{
    __shared__ double hist[33][kFeatThreads];
    const int64_t i = (int64_t)blockIdx.x * kFeatThreads + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 33; ++j) hist[j][threadIdx.x] = 0.0;
    if (i >= n) return;
    const int m = cnt[i];
    if (m > 1) {
        const double p1[3] = { pts[3 * i], pts[3 * i + 1], pts[3 * i + 2] }, n1[3] = { nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2] };
        const double inc = 100.0 / (double)(m - 1);
        for (int t = 1; t < m; ++t) {                  // slot 0 is the point itself
            const int64_t j = nbr[i * k + t];
            const double p2[3] = { pts[3 * j], pts[3 * j + 1], pts[3 * j + 2] }, n2[3] = { nrm[3 * j], nrm[3 * j + 1], nrm[3 * j + 2] };
            double f[4];
            pair_features(p1, n1, p2, n2, f);
            hist[bin11(11.0 * (f[0] + M_PI) / (2.0 * M_PI))][threadIdx.x] += inc;
            hist[11 + bin11(11.0 * (f[1] + 1.0) * 0.5)][threadIdx.x] += inc;
            hist[22 + bin11(11.0 * (f[2] + 1.0) * 0.5)][threadIdx.x] += inc;
        }
    }
#pragma unroll
    for (int j = 0; j < 33; ++j) spfh[i * 33 + j] = hist[j][threadIdx.x];
}

// One wave per point: lane j < 33 owns component j, so every neighbour's SPFH row is one coalesced read.
__global__ __launch_bounds__(256) void fpfh_kernel(int64_t n, const int32_t *__restrict__ nbr, const double *__restrict__ d2,
                                                   const int32_t *__restrict__ cnt, int k, const double *__restrict__ spfh,
                                                   double *__restrict__ fpfh)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int m = cnt[i];
    double acc = 0.0;
    if (m > 1) {
        for (int t = 1; t < m; ++t) {                      // slot 0 is the point itself
            const double dist = d2[i * k + t];
            if (dist == 0.0) continue;
            const double *sp = spfh + 33 * (int64_t)nbr[i * k + t];
            if (lane < 33) acc += sp[lane] / dist;
        }
        // per-feature normalisation: 100 / (sum over the 11 bins of the feature)
        const double a0 = lane < 11 ? acc : 0.0, a1 = (lane >= 11 && lane < 22) ? acc : 0.0, a2 = (lane >= 22 && lane < 33) ? acc : 0.0;
        double s0 = a0, s1 = a1, s2 = a2;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        double sum = lane < 11 ? s0 : (lane < 22 ? s1 : s2);
        if (sum != 0.0) sum = 100.0 / sum;
        if (lane < 33) acc = acc * sum + spfh[i * 33 + lane];
    }
    if (lane < 33) fpfh[i * 33 + lane] = acc;
}

// ---- 33-D feature nearest neighbour -----------------------------------------------------------------------------
// The matching stage is a dense all-pairs problem (no spatial structure to cull in 33-D): fp64 MFMA distance tiles in
// the augmented form, K = 36:
//     A[i] = (a_0 .. a_32, 1, 0, 0)        B[j] = (-2 b_0 .. -2 b_32, |b_j|^2, 0, 0)        C[i] = |a_i|^2
//     D_ij = fma chain over k = 0 .. 35 of A_ik B_kj seeded with C_i     (nine v_mfma_f64_16x16x4_f64 per tile;
//     |x|^2 = fma chain x_k x_k from 0)  -> argmin_j, ties to the lowest j.  The oracle restates the same chain.
// Block = 4 waves x 16 query rows; 128 target columns per stage go through LDS (row-major, stride 37 doubles:
// conflict-free for the 16 columns a lane group reads) already scaled by -2, their norms are formed once per stage.
typedef double fnn_d4 __attribute__((ext_vector_type(4)));
constexpr int kFnnCols = 128, kFnnStride = 37;
// gridDim.y column splits (stages of 64 columns dealt round-robin); split s writes (value, column) of its best to
// part_val / part_idx [s][na]; feature_nn_merge_kernel takes the lexicographic minimum.
__global__ __launch_bounds__(256) void feature_nn_kernel(const double *__restrict__ fa, int64_t na, const double *__restrict__ fb,
                                                         int64_t nb, double *__restrict__ part_val, int32_t *__restrict__ part_idx)
{
    __shared__ double sb[kFnnCols][kFnnStride];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
    const int64_t row_base = ((int64_t)blockIdx.x * 4 + wave) * 16;
    // A operand: component k = q + 4 s of row j
    double a[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int k = q + 4 * s;
        const int64_t row = row_base + j;
        a[s] = k < 33 ? (row < na ? fa[row * 33 + k] : 0.0) : (k == 33 ? 1.0 : 0.0);
    }
    // C operand: |a|^2 of the rows this lane sees in D (row = q + 4 r)
    fnn_d4 seed;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row_base + q + 4 * r;
        double n2 = 0.0;
        if (row < na)
            for (int k = 0; k < 33; ++k) { const double v = fa[row * 33 + k]; n2 = fma(v, v, n2); }
        seed[r] = n2;
    }
    double best[4] = { INFINITY, INFINITY, INFINITY, INFINITY };
    int32_t bcol[4] = { INT_MAX, INT_MAX, INT_MAX, INT_MAX };
    for (int64_t j0 = (int64_t)blockIdx.y * kFnnCols; j0 < nb; j0 += (int64_t)gridDim.y * kFnnCols) {
        const int cnt = nb - j0 < kFnnCols ? (int)(nb - j0) : kFnnCols;
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * 33; e += 256) sb[e / 33][e % 33] = -2.0 * fb[j0 * 33 + e];    // B rows: -2 b (exact)
        __syncthreads();
        if ((int)threadIdx.x < kFnnCols) {
            double n2 = kSentinelF;                            // columns past the end can never win
            if ((int)threadIdx.x < cnt) {
                n2 = 0.0;
                for (int k = 0; k < 33; ++k) { const double v = -0.5 * sb[threadIdx.x][k]; n2 = fma(v, v, n2); }
            } else {
                for (int k = 0; k < 33; ++k) sb[threadIdx.x][k] = 0.0;
            }
            sb[threadIdx.x][33] = n2; sb[threadIdx.x][34] = 0.0; sb[threadIdx.x][35] = 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < kFnnCols / 16; ++ct) {
            fnn_d4 acc = seed;
            const double *col = sb[ct * 16 + j] + q;           // B[k = q + 4 s][column]: no arithmetic beside the MFMAs
#pragma unroll
            for (int s = 0; s < 9; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], col[4 * s], acc, 0, 0, 0);
            const int32_t c = (int32_t)(j0 + ct * 16 + j);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool t = acc[r] < best[r];               // ascending columns per lane: first minimum kept
                best[r] = t ? acc[r] : best[r];
                bcol[r] = t ? c : bcol[r];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double v = best[r];
        int32_t c = bcol[r];
#pragma unroll
        for (int msk = 1; msk < 16; msk <<= 1) {
            const double ov = __shfl_xor(v, msk, 64);
            const int32_t oc = __shfl_xor(c, msk, 64);
            const bool take = ov < v || (ov == v && oc < c);
            v = take ? ov : v;
            c = take ? oc : c;
        }
        const int64_t row = row_base + q + 4 * r;
        if (j == 0 && row < na) { part_val[(int64_t)blockIdx.y * na + row] = v; part_idx[(int64_t)blockIdx.y * na + row] = c; }
    }
}
__global__ __launch_bounds__(256) void feature_nn_merge_kernel(const double *__restrict__ part_val, const int32_t *__restrict__ part_idx,
                                                               int64_t na, int splits, int32_t *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na) return;
    double bv = part_val[i];
    int32_t bj = part_idx[i];
    for (int s = 1; s < splits; ++s) {
        const double v = part_val[(int64_t)s * na + i];
        const int32_t c = part_idx[(int64_t)s * na + i];
        if (v < bv || (v == bv && c < bj)) { bv = v; bj = c; }
    }
    idx[i] = bj;
}

// ---- RANSAC hypotheses -----------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// one thread per iteration `itr0 + t`: sample 3 correspondences (with replacement), edge-length check, Umeyama,
// distance check.  pass[t] = 1 and T[t] (16 doubles) when the hypothesis survives.
__global__ __launch_bounds__(256) void ransac_hyp_kernel(const float *__restrict__ src, const float *__restrict__ tgt,
                                                         const int32_t *__restrict__ corres, int64_t nc, int32_t itr0, int32_t count,
                                                         uint32_t seed_lo, uint32_t seed_hi, double edge_sim, double max_dist,
                                                         uint8_t *__restrict__ pass, double *__restrict__ Ts)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    uint32_t out[4];
    philox4x32_10(0u, (uint32_t)(itr0 + t), 1u, 0u, seed_lo, seed_hi, out);
    double sp[3][3], tp[3][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int64_t pick = (int64_t)(((uint64_t)out[q] * (uint64_t)nc) >> 32);
        const float *s = src + 3 * (int64_t)corres[2 * pick], *g = tgt + 3 * (int64_t)corres[2 * pick + 1];
#pragma unroll
        for (int a = 0; a < 3; ++a) { sp[q][a] = s[a]; tp[q][a] = g[a]; }
    }
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i + 1; j < 3; ++j) {
            double ds = 0.0, dt = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) { double e = sp[i][a] - sp[j][a]; ds += e * e; e = tp[i][a] - tp[j][a]; dt += e * e; }
            ds = sqrt(ds); dt = sqrt(dt);
            ok = ok && !(ds < dt * edge_sim || dt < ds * edge_sim);
        }
    pass[t] = 0;
    if (!ok) return;
    double ms[3] = { 0, 0, 0 }, mt[3] = { 0, 0, 0 }, S[9], R[9];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int a = 0; a < 3; ++a) { ms[a] += sp[q][a]; mt[a] += tp[q][a]; }
#pragma unroll
    for (int a = 0; a < 3; ++a) { ms[a] /= 3.0; mt[a] /= 3.0; }
#pragma unroll
    for (int a = 0; a < 9; ++a) S[a] = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) S[3 * a + b] += (tp[q][a] - mt[a]) * (sp[q][b] - ms[b]);
    kabsch_rotation(S, R);
    double T[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) T[k] = (k % 5 == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) T[4 * a + b] = R[3 * a + b];
        T[4 * a + 3] = mt[a] - (R[3 * a] * ms[0] + R[3 * a + 1] * ms[1] + R[3 * a + 2] * ms[2]);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double e2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double o = fma(T[4 * a], sp[q][0], fma(T[4 * a + 1], sp[q][1], fma(T[4 * a + 2], sp[q][2], T[4 * a + 3])));
            e2 += (o - tp[q][a]) * (o - tp[q][a]);
        }
        ok = ok && !(sqrt(e2) > max_dist);
    }
    if (!ok) return;
    pass[t] = 1;
#pragma unroll
    for (int k = 0; k < 16; ++k) Ts[(int64_t)t * 16 + k] = T[k];
}
struct PassPred {
    const uint8_t *pass;
    __device__ bool operator()(int64_t i, int) const { return pass[i] != 0; }
};
struct PassEmit {
    int32_t *list;
    __device__ void operator()(int64_t i, int, int32_t dst) const { list[dst] = (int32_t)i; }
};

// single-slot "heap": nearest point with d2 < r2max
struct Nearest1 {
    double best; int32_t bi;
    __device__ bool full() const { return bi >= 0; }
    __device__ double worst() const { return best; }
    __device__ void push(double d, int j) { if (d < best || (d == best && (j < bi || bi < 0))) { best = d; bi = j; } }
};

// validation of surviving hypotheses: for hypothesis h = list[blockIdx.y], every source point transformed by T_h
// looks for its nearest target point within max_dist on the target grid; per-block (count, sum d2) partials.
__global__ __launch_bounds__(256) void ransac_validate_kernel(const float *__restrict__ src, int64_t n, const GridParams *__restrict__ gp,
                                                              const uint32_t *__restrict__ cell_start, const float *__restrict__ tpts,
                                                              const int32_t *__restrict__ list, const double *__restrict__ Ts, double r2,
                                                              int64_t *__restrict__ part_cnt, double *__restrict__ part_err,
                                                              const int32_t *__restrict__ n_list)
{
    __shared__ double shd[4];
    __shared__ long long shc[4];
    if (n_list && (int)blockIdx.y >= *n_list) return;       // speculative launch: fewer survivors than slots
    const int h = list[blockIdx.y];
    const double *T = Ts + (int64_t)h * 16;
    const GridParams g = *gp;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    long long c = 0;
    double e = 0.0;
    if (i < n) {
        const double x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
        double s[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] = fma(T[4 * k], x, fma(T[4 * k + 1], y, fma(T[4 * k + 2], z, T[4 * k + 3])));
        // points outside the grid by more than the radius cannot match
        bool far = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double lo = g.org[a], hi = g.org[a] + (double)g.dim[a] * g.h;
            const double o = s[a] < lo ? lo - s[a] : (s[a] > hi ? s[a] - hi : 0.0);
            far = far || o * o >= r2;
        }
        if (!far) {
            Nearest1 nn{ r2, -1 };
            grid_knn_scan(g, cell_start, tpts, (const int32_t *)nullptr, s[0], s[1], s[2], r2, nn);
            if (nn.best < r2) { c = 1; e = nn.best; }
        }
    }
    e = block_sum(e, shd);
    c = block_sum(c, shc);
    if (threadIdx.x == 0) {
        part_cnt[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = c;
        part_err[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = e;
    }
}
// inlier ratio of the correspondence set under T_h (EvaluateInlierCorrespondenceRatio) + reduction of the partials
__global__ __launch_bounds__(256) void ransac_score_kernel(const float *__restrict__ src, const float *__restrict__ tgt,
                                                           const int32_t *__restrict__ corres, int64_t nc, const int32_t *__restrict__ list,
                                                           const double *__restrict__ Ts, double max_dist, const int64_t *__restrict__ part_cnt,
                                                           const double *__restrict__ part_err, int nblocks, double *__restrict__ score,
                                                           const int32_t *__restrict__ n_list)
{
    __shared__ long long shc[4];
    if (n_list && (int)blockIdx.x >= *n_list) return;
    const int h = list[blockIdx.x];
    const double *T = Ts + (int64_t)h * 16;
    long long inl = 0;
    for (int64_t c = threadIdx.x; c < nc; c += blockDim.x) {
        const float *s = src + 3 * (int64_t)corres[2 * c], *g = tgt + 3 * (int64_t)corres[2 * c + 1];
        double e2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double o = fma(T[4 * a], (double)s[0], fma(T[4 * a + 1], (double)s[1], fma(T[4 * a + 2], (double)s[2], T[4 * a + 3])));
            e2 += (o - (double)g[a]) * (o - (double)g[a]);
        }
        inl += sqrt(e2) < max_dist;
    }
    inl = block_sum(inl, shc);
    if (threadIdx.x == 0) {
        long long cnt = 0; double err = 0.0;
        for (int b = 0; b < nblocks; ++b) { cnt += part_cnt[(int64_t)blockIdx.x * nblocks + b]; err += part_err[(int64_t)blockIdx.x * nblocks + b]; }
        score[4 * blockIdx.x] = (double)cnt; score[4 * blockIdx.x + 1] = err; score[4 * blockIdx.x + 2] = (double)inl;
        score[4 * blockIdx.x + 3] = (double)h;
    }
}

static int knn_lists(const float *pts, int64_t n, double radius, int k, Arena &a, Grid *g, int32_t **nbr, double **d2, int32_t **cnt, hipStream_t st)
{
    int rc = grid_build(pts, n, 16.0, a, g, st);
    if (rc) return rc;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    *nbr = a.get<int32_t>(nn * k);
    *d2 = a.get<double>(nn * k);
    *cnt = a.get<int32_t>(nn);
    int32_t *fb_list = a.get<int32_t>(nn + 1);
    const bool global_heap = k > KPX_NORMALS_LDS_NN;              // the fall-back heaps in the workspace (<= 256 MB of distances)
    int heap_blocks = 256;
    while (global_heap && heap_blocks > 8 && (size_t)heap_blocks * 64 * (size_t)k * sizeof(double) > ((size_t)256 << 20)) heap_blocks >>= 1;
    double *gheap = global_heap ? a.get<double>((size_t)heap_blocks * 64 * (size_t)k) : nullptr;
    int32_t *gix = global_heap ? a.get<int32_t>((size_t)heap_blocks * 64 * (size_t)k) : nullptr;
    if (a.dry) return KPX_OK;
    KPX_ARENA_CHECK(a);
    int32_t *fb_count = fb_list + nn;
    const int threads = global_heap ? 64 : (k <= 48 ? 128 : 64);
    const size_t lds = global_heap ? 0 : (size_t)k * threads * (sizeof(double) + sizeof(int32_t));
    static bool attr_set = false;
    if (!attr_set) {
        KPX_HIP(hipFuncSetAttribute((const void *)nbr_list_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        KPX_HIP(hipFuncSetAttribute((const void *)nbr_list_wave_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    KPX_HIP(hipMemsetAsync(fb_count, 0, sizeof(int32_t), st));
    // pass 1: one wave per query, 512- or 1024-candidate buffer; pass 2: the queries that did not fit, thread-per-query heap walk
    const int cap = k <= 48 ? 512 : 1024;
    const bool wave_pass = k <= 512;                               // (beyond, the 1024-candidate buffer cannot hold a neighbourhood: every query takes the heap walk)
    if (wave_pass)
        hipLaunchKernelGGL(nbr_list_wave_kernel<4>, dim3((unsigned)(cdiv(n, 4) > 8192 ? 8192 : cdiv(n, 4))), dim3(256),
                           (size_t)4 * cap * (sizeof(double) + sizeof(uint32_t)), st, g->params, g->cell_start, g->sorted_pts, g->sorted_idx, n, k, cap,
                           radius * radius, *nbr, *d2, *cnt, fb_list, fb_count);
    hipLaunchKernelGGL(nbr_list_kernel, dim3(global_heap ? heap_blocks : 256), dim3(threads), lds, st, g->params, g->cell_start, g->sorted_pts, g->sorted_idx, n, k,
                       radius * radius, *nbr, *d2, *cnt, wave_pass ? fb_list : (const int32_t *)nullptr, wave_pass ? fb_count : (const int32_t *)nullptr, gheap, gix);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

static int fpfh_impl(const float *pts, const float *nrm, int64_t n, double radius, int max_nn, double *fpfh, Arena &a, hipStream_t st)
{
    Grid g;
    int32_t *nbr, *cnt; double *d2;
    int kk = (int64_t)max_nn < n ? max_nn : (int)(n > 0 ? n : 1);
    int rc = knn_lists(pts, n, radius, kk, a, &g, &nbr, &d2, &cnt, st);
    if (rc) return rc;
    double *spfh = a.get<double>((size_t)(n > 0 ? n : 1) * 33);
    if (a.dry) return KPX_OK;
    KPX_ARENA_CHECK(a);
    hipLaunchKernelGGL(spfh_kernel, dim3((unsigned)cdiv(n, kFeatThreads)), dim3(kFeatThreads), 0, st, pts, nrm, n, nbr, cnt, kk, spfh);
    hipLaunchKernelGGL(fpfh_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, st, n, nbr, d2, cnt, kk, spfh, fpfh);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}


// ---- colour gradient of the target of a coloured ICP ([O3D] InitializePointCloudForColoredICP) ----------------------------
// Per point: least squares over its hybrid neighbourhood (the list's first entry, the point itself, is skipped) of
// intensity difference against the offset projected onto the tangent plane, plus the row (nn - 1) n = 0 that keeps the
// gradient in that plane; fewer than 4 neighbours: zero.  Normal equations A^T A x = A^T b, 3x3, solved by cofactors.
__global__ __launch_bounds__(256) void color_gradient_kernel(const float *__restrict__ pts, const float *__restrict__ nrm, const float *__restrict__ col,
                                                             int64_t n, const int32_t *__restrict__ nbr, const int32_t *__restrict__ cnt, int k,
                                                             double *__restrict__ grad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x[3] = { 0.0, 0.0, 0.0 };
    const int nn = cnt[i];
    if (nn >= 4) {
        const double vt[3] = { pts[3 * i], pts[3 * i + 1], pts[3 * i + 2] }, nt[3] = { nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2] };
        const double it = ((double)col[3 * i] + (double)col[3 * i + 1] + (double)col[3 * i + 2]) / 3.0;
        double a = 0, b = 0, c = 0, d = 0, e = 0, f = 0, r0 = 0, r1 = 0, r2 = 0;       // A^T A = [a b c; b d e; c e f], A^T b = r
        for (int t = 1; t < nn; ++t) {
            const int64_t j = nbr[i * k + t];
            const double q[3] = { pts[3 * j], pts[3 * j + 1], pts[3 * j + 2] };
            const double dp = (q[0] - vt[0]) * nt[0] + (q[1] - vt[1]) * nt[1] + (q[2] - vt[2]) * nt[2];
            const double u[3] = { (q[0] - dp * nt[0]) - vt[0], (q[1] - dp * nt[1]) - vt[1], (q[2] - dp * nt[2]) - vt[2] };
            const double bi = ((double)col[3 * j] + (double)col[3 * j + 1] + (double)col[3 * j + 2]) / 3.0 - it;
            a += u[0] * u[0]; b += u[0] * u[1]; c += u[0] * u[2]; d += u[1] * u[1]; e += u[1] * u[2]; f += u[2] * u[2];
            r0 += u[0] * bi; r1 += u[1] * bi; r2 += u[2] * bi;
        }
        const double w = (double)(nn - 1), w2 = w * w;
        a += w2 * nt[0] * nt[0]; b += w2 * nt[0] * nt[1]; c += w2 * nt[0] * nt[2];
        d += w2 * nt[1] * nt[1]; e += w2 * nt[1] * nt[2]; f += w2 * nt[2] * nt[2];
        const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
        const double det = a * c00 + b * c01 + c * c02;
        if (det != 0.0 && isfinite(det)) {
            const double c11 = a * f - c * c, c12 = b * c - a * e, c22 = a * d - b * b;
            x[0] = (c00 * r0 + c01 * r1 + c02 * r2) / det;
            x[1] = (c01 * r0 + c11 * r1 + c12 * r2) / det;
            x[2] = (c02 * r0 + c12 * r1 + c22 * r2) / det;
        }
    }
    grad[3 * i] = x[0]; grad[3 * i + 1] = x[1]; grad[3 * i + 2] = x[2];
}
static int color_gradient_impl(const float *pts, const float *nrm, const float *col, int64_t n, double radius, int max_nn, double *grad, Arena &a,
                               hipStream_t st)
{
    Grid g;
    int32_t *nbr, *cnt; double *d2;
    const int kk = (int64_t)max_nn < n ? max_nn : (int)(n > 0 ? n : 1);
    int rc = knn_lists(pts, n, radius, kk, a, &g, &nbr, &d2, &cnt, st);
    if (rc || a.dry) return rc;
    KPX_ARENA_CHECK(a);
    hipLaunchKernelGGL(color_gradient_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, pts, nrm, col, n, nbr, cnt, kk, grad);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

constexpr int kRansacBatch = 32768;
struct RansacBuffers {
    Grid g;
    uint8_t *pass;
    double *Ts, *score, *part_err;
    int64_t *part_cnt;
    int32_t *list, *counts, *n_pass;
    int vblocks;
};
static int ransac_carve(const float *tgt, int64_t n_src, int64_t n_tgt, int max_validate, Arena &a, RansacBuffers *b, hipStream_t st)
{
    int rc = grid_build(tgt, n_tgt, 8.0, a, &b->g, st);
    if (rc) return rc;
    b->vblocks = (int)cdiv(n_src > 0 ? n_src : 1, 256);
    b->pass = a.get<uint8_t>(kRansacBatch);
    b->Ts = a.get<double>((size_t)kRansacBatch * 16);
    b->list = a.get<int32_t>(kRansacBatch);
    b->counts = a.get<int32_t>((size_t)compact_ws_ints(kRansacBatch));
    b->n_pass = a.get<int32_t>(1);
    b->part_cnt = a.get<int64_t>((size_t)max_validate * b->vblocks);
    b->part_err = a.get<double>((size_t)max_validate * b->vblocks);
    b->score = a.get<double>((size_t)max_validate * 4);
    return KPX_OK;
}
constexpr int kMaxValidate = 512;      // validations per launch (surviving hypotheses are processed in chunks)

}  // namespace kpx

using namespace kpx;

KPX_EXPORT size_t kpx_color_gradient_workspace_bytes(int64_t n, int32_t max_nn)
{
    Arena a(nullptr, 0);
    color_gradient_impl(nullptr, nullptr, nullptr, n, 1.0, max_nn < 1 ? 1 : max_nn, nullptr, a, nullptr);
    return a.off;
}
KPX_EXPORT int kpx_color_gradient(const float *pts, const float *normals, const float *colors, int64_t n, double radius, int32_t max_nn,
                                  double *grad, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(radius > 0.0 && max_nn >= 1 && max_nn <= KPX_NORMALS_MAX_NN, "kpx_color_gradient: radius / max_nn out of range");
    KPX_REQUIRE(n >= 0 && n * (int64_t)max_nn < ((int64_t)1 << 31), "kpx_color_gradient: n x max_nn must stay below 2^31");
    if (n == 0) return KPX_OK;
    KPX_REQUIRE(pts && normals && colors && grad && ws, "kpx_color_gradient: null pointer");
    Arena a(ws, ws_bytes);
    return color_gradient_impl(pts, normals, colors, n, radius, max_nn, grad, a, (hipStream_t)stream);
}

KPX_EXPORT size_t kpx_fpfh_workspace_bytes(int64_t n, int32_t max_nn)
{
    Arena a(nullptr, 0);
    fpfh_impl(nullptr, nullptr, n, 1.0, max_nn < 1 ? 1 : max_nn, nullptr, a, nullptr);
    return a.off;
}
KPX_EXPORT int kpx_fpfh(const float *pts, const float *normals, int64_t n, double radius, int32_t max_nn, double *fpfh, void *ws,
                        size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(radius > 0.0 && max_nn >= 1, "compute_fpfh_feature: radius and max_nn must be positive");
    KPX_REQUIRE(max_nn <= KPX_NORMALS_MAX_NN, "compute_fpfh_feature: max_nn > %d is not supported", KPX_NORMALS_MAX_NN);
    KPX_REQUIRE(n >= 0 && n * (int64_t)max_nn < ((int64_t)1 << 31), "kpx_fpfh: n x max_nn must stay below 2^31");
    if (n == 0) return KPX_OK;
    KPX_REQUIRE(pts && fpfh && ws, "kpx_fpfh: null pointer");
    KPX_REQUIRE(normals, "Failed because input point cloud has no normal.");      // [O3D]
    Arena a(ws, ws_bytes);
    return fpfh_impl(pts, normals, n, radius, max_nn, fpfh, a, (hipStream_t)stream);
}

static int feature_nn_splits(int64_t na, int64_t nb)
{
    // aim for >= ~1024 blocks (4 per CU); a split needs at least one 64-column stage
    const int64_t row_blocks = cdiv(na > 0 ? na : 1, 64), stages = cdiv(nb > 0 ? nb : 1, kFnnCols);
    int64_t s = cdiv(1024, row_blocks);
    if (s > stages) s = stages;
    if (s > 16) s = 16;
    return (int)(s < 1 ? 1 : s);
}
KPX_EXPORT size_t kpx_feature_nn_workspace_bytes(int64_t na, int64_t nb)
{
    Arena a(nullptr, 0);
    const size_t e = (size_t)(na > 0 ? na : 1) * feature_nn_splits(na, nb);
    a.get<double>(e);
    a.get<int32_t>(e);
    return a.off;
}
KPX_EXPORT int kpx_feature_nn(const double *fa, int64_t na, const double *fb, int64_t nb, int32_t *idx, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(na >= 0 && nb >= 1, "kpx_feature_nn: empty feature set");
    if (na == 0) return KPX_OK;
    KPX_REQUIRE(fa && fb && idx && ws, "kpx_feature_nn: null pointer");
    const int splits = feature_nn_splits(na, nb);
    Arena a(ws, ws_bytes);
    double *pv = a.get<double>((size_t)na * splits);
    int32_t *pi = a.get<int32_t>((size_t)na * splits);
    KPX_ARENA_CHECK(a);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(feature_nn_kernel, dim3((unsigned)cdiv(na, 64), splits), dim3(256), 0, st, fa, na, fb, nb, pv, pi);
    hipLaunchKernelGGL(feature_nn_merge_kernel, dim3((unsigned)cdiv(na, 256)), dim3(256), 0, st, pv, pi, na, splits, idx);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

KPX_EXPORT size_t kpx_ransac_workspace_bytes(int64_t n_src, int64_t n_tgt)
{
    Arena a(nullptr, 0);
    RansacBuffers b;
    ransac_carve(nullptr, n_src, n_tgt, kMaxValidate, a, &b, nullptr);
    return a.off;
}
// h_result: T (16) | fitness | inlier_rmse | iterations run | validations
KPX_EXPORT int kpx_ransac_corres(const float *src, int64_t n_src, const float *tgt, int64_t n_tgt, const int32_t *corres, int64_t n_corres,
                                 double max_dist, int32_t ransac_n, double edge_similarity, int32_t max_iteration, double confidence,
                                 uint64_t seed, double *h_result, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(h_result, "kpx_ransac_corres: null result");
    for (int k = 0; k < 16; ++k) h_result[k] = (k % 5 == 0) ? 1.0 : 0.0;
    h_result[16] = h_result[17] = h_result[18] = h_result[19] = 0.0;
    KPX_REQUIRE(ransac_n == 3, "kpx_ransac_corres: ransac_n must be 3 (preprocessing/registration.py:53)");
    if (n_corres < ransac_n || !(max_dist > 0.0)) return KPX_OK;                  // [O3D] returns an empty RegistrationResult
    KPX_REQUIRE(n_src >= 1 && n_tgt >= 1 && src && tgt && corres && ws, "kpx_ransac_corres: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    Arena a(ws, ws_bytes);
    RansacBuffers b;
    int rc = ransac_carve(tgt, n_src, n_tgt, kMaxValidate, a, &b, st);
    if (rc) return rc;
    KPX_ARENA_CHECK(a);
    const double r2 = max_dist * max_dist;
    double best_fit = 0.0, best_rmse = 0.0;
    int est_k = max_iteration, validations = 0, itr_done = 0;
    // Per batch: hypotheses -> survivors compacted in iteration order -> the first kSpec survivors are validated and scored
    // speculatively (kernels guarded by the device count) -> ONE read-back (count, list, scores) -> Open3D's loop body is
    // replayed on the host in iteration order.  Batches with more than kSpec survivors validate the rest in further chunks.
    constexpr int kSpec = 64;
    std::vector<int32_t> h_list(kRansacBatch);
    std::vector<double> h_score((size_t)kMaxValidate * 4);
    for (int itr0 = 0; itr0 < max_iteration && itr0 < est_k; itr0 += kRansacBatch) {
        const int count = max_iteration - itr0 < kRansacBatch ? max_iteration - itr0 : kRansacBatch;
        hipLaunchKernelGGL(ransac_hyp_kernel, dim3((unsigned)cdiv(count, 256)), dim3(256), 0, st, src, tgt, corres, n_corres, itr0, count,
                           (uint32_t)seed, (uint32_t)(seed >> 32), edge_similarity, max_dist, b.pass, b.Ts);
        rc = compact(PassPred{ b.pass }, PassEmit{ b.list }, count, 1, b.counts, b.n_pass, st);
        if (rc) return rc;
        hipLaunchKernelGGL(ransac_validate_kernel, dim3(b.vblocks, kSpec), dim3(256), 0, st, src, n_src, b.g.params, b.g.cell_start,
                           b.g.sorted_pts, b.list, b.Ts, r2, b.part_cnt, b.part_err, b.n_pass);
        hipLaunchKernelGGL(ransac_score_kernel, dim3(kSpec), dim3(256), 0, st, src, tgt, corres, n_corres, b.list, b.Ts, max_dist, b.part_cnt,
                           b.part_err, b.vblocks, b.score, b.n_pass);
        int32_t n_pass = 0;
        KPX_HIP(hipMemcpyAsync(&n_pass, b.n_pass, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        KPX_HIP(hipMemcpyAsync(h_score.data(), b.score, (size_t)kSpec * 4 * sizeof(double), hipMemcpyDeviceToHost, st));
        KPX_HIP(hipStreamSynchronize(st));
        if (n_pass > kSpec) KPX_HIP(hipMemcpy(h_list.data(), b.list, (size_t)n_pass * sizeof(int32_t), hipMemcpyDeviceToHost));
        bool stop = false;
        int best_in_batch = -1;
        for (int c0 = 0; c0 < n_pass && !stop; c0 += (c0 == 0 ? kSpec : kMaxValidate)) {
            int nv = n_pass - c0 < kSpec ? n_pass - c0 : kSpec;
            if (c0 > 0) {                                               // beyond the speculative chunk
                nv = n_pass - c0 < kMaxValidate ? n_pass - c0 : kMaxValidate;
                if (itr0 + h_list[c0] >= est_k) break;                  // everything left is beyond the exit iteration
                hipLaunchKernelGGL(ransac_validate_kernel, dim3(b.vblocks, nv), dim3(256), 0, st, src, n_src, b.g.params, b.g.cell_start,
                                   b.g.sorted_pts, b.list + c0, b.Ts, r2, b.part_cnt, b.part_err, (const int32_t *)nullptr);
                hipLaunchKernelGGL(ransac_score_kernel, dim3(nv), dim3(256), 0, st, src, tgt, corres, n_corres, b.list + c0, b.Ts, max_dist,
                                   b.part_cnt, b.part_err, b.vblocks, b.score, (const int32_t *)nullptr);
                KPX_HIP(hipMemcpyAsync(h_score.data(), b.score, (size_t)nv * 4 * sizeof(double), hipMemcpyDeviceToHost, st));
                KPX_HIP(hipStreamSynchronize(st));
            }
            for (int v = 0; v < nv; ++v) {                               // Open3D's loop body, in iteration order
                const int itr = itr0 + (int)h_score[4 * v + 3];
                if (itr >= est_k) { stop = true; break; }
                ++validations;
                const double cnt = h_score[4 * v], err = h_score[4 * v + 1];
                const double fit = cnt > 0 ? cnt / (double)n_src : 0.0, rmse = cnt > 0 ? sqrt(err / cnt) : 0.0;
                if (fit > best_fit || (fit == best_fit && rmse < best_rmse)) {
                    best_fit = fit; best_rmse = rmse;
                    best_in_batch = (int)h_score[4 * v + 3];
                    const double ratio = h_score[4 * v + 2] / (double)n_corres;
                    const double ek = log(1.0 - confidence) / log(1.0 - pow(ratio, (double)ransac_n));
                    if (ek < (double)est_k) est_k = (int)ceil(ek);
                }
            }
        }
        if (best_in_batch >= 0)                                         // the batch's transforms are overwritten by the next one
            KPX_HIP(hipMemcpy(h_result, b.Ts + (int64_t)best_in_batch * 16, 16 * sizeof(double), hipMemcpyDeviceToHost));
        itr_done = itr0 + count < est_k ? itr0 + count : est_k;
        if (stop) break;
    }
    h_result[16] = best_fit; h_result[17] = best_rmse; h_result[18] = (double)itr_done; h_result[19] = (double)validations;
    return KPX_OK;
}
