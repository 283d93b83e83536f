// kpx_nnlocal.h -- the exact nearest-neighbour sweep with spatial tile culling (included by kpx_icp.hip only).
//
// Same arithmetic as the dense sweep (contract AC2: v_mfma_f64_16x16x4_f64 = the k-ordered fma chain seeded with
// K_i; argmin, ties to the lowest ORIGINAL target index), but only the 16-column tiles that can hold a column
// with D <= the row bound are multiplied:
//   * the target is sorted once along a 30-bit Morton curve (cubic cells over its bounding box); 16 consecutive
//     sorted points form a column tile, 16 tiles a group; every tile and every group carries its exact float
//     bounding box.  orig[col] maps a sorted column back to the caller's index.
//   * the source rows are sorted the same way once per registration (a rigid transform keeps them compact); a wave
//     owns 16 consecutive sorted rows.  Its box (from the transformed rows) and its radius R^2 = max_i bound_i - 1
//     select the groups, then the tiles, with gap^2(row box, tile box) <= R^2 (+ a margin 2^-38 (K + |t|^2) that
//     covers the rounding of the expanded metric; the skipped columns are STRICTLY farther than every row's bound, so
//     neither the minimum nor a tie can hide there).  Surviving tiles go to a list in LDS and are multiplied four per
//     trip behind the same 32-bit high-word prefilter as the dense sweep.
//   * bounds: the previous partner (ICP iterations >= 1), clamped to max_correspondence_distance^2 inside a
//     registration (rows with nothing inside report "no partner": they are not correspondences by definition,
//     pipelines/registration/Registration.cpp semantics [O3D]); without either, the far-corner distance to the
//     nearest group box is a valid radius for all 16 rows.
// The result is bit-identical to the dense sweep and to the oracle wherever a partner exists within the bound.
#pragma once
#include "kpx_morton.h"

namespace kpx {

constexpr int kLGroupTiles = 16;                 // tiles per group (256 sorted columns)
constexpr int kLRows = 16;                       // source rows per wave
constexpr int kLList = 512;                      // tile list capacity (LDS, per wave)
constexpr int kVisitSlots = 1024;                // profiling counter slots

// Sorted fp64 B operand (element (k, j) of tile t at Bs[t*64 + k*16 + j]), original index of every sorted column
// (INT_MAX in the padding), tile and group boxes (lo xyz, hi xyz as floats; empty = (+big, -big)).  One block = one group.
__global__ __launch_bounds__(256) void nn_local_prep_kernel(const float *__restrict__ tgt, int64_t m, double *__restrict__ Bs,
                                                            int32_t *__restrict__ orig, float *__restrict__ tile_box,
                                                            float *__restrict__ group_box)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = kSentinel;
    float lo[3] = { kBoxBig, kBoxBig, kBoxBig }, hi[3] = { -kBoxBig, -kBoxBig, -kBoxBig };
    if (j < m) {
        const int64_t o = orig[j];                  // written by the sort
        const float fx = tgt[3 * o], fy = tgt[3 * o + 1], fz = tgt[3 * o + 2];
        const double tx = fx, ty = fy, tz = fz;
        b0 = -2.0 * tx; b1 = -2.0 * ty; b2 = -2.0 * tz;
        b3 = fma(tx, tx, fma(ty, ty, tz * tz));
        lo[0] = hi[0] = fx; lo[1] = hi[1] = fy; lo[2] = hi[2] = fz;
    } else {
        orig[j] = INT_MAX;
    }
    double *o = Bs + (j >> 4) * 64 + (j & 15);
    o[0] = b0; o[16] = b1; o[32] = b2; o[48] = b3;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int msk = 1; msk < 16; msk <<= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], msk, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], msk, 64));
        }
    __shared__ float sb[16][6];
    if ((threadIdx.x & 15) == 0) {
        float *tb = tile_box + (j >> 4) * 6;
#pragma unroll
        for (int a = 0; a < 3; ++a) { tb[a] = lo[a]; tb[3 + a] = hi[a]; sb[threadIdx.x >> 4][a] = lo[a]; sb[threadIdx.x >> 4][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sb[0][threadIdx.x];
        for (int t = 1; t < 16; ++t) v = threadIdx.x < 3 ? fminf(v, sb[t][threadIdx.x]) : fmaxf(v, sb[t][threadIdx.x]);
        group_box[(int64_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// Per-row operands in sorted row order.  prev (indexed by ORIGINAL row) = partners of the last search or NULL;
// max_d2 > 0 clamps the bound to the correspondence distance.
__global__ __launch_bounds__(256) void nn_local_rowprep_kernel(const float *__restrict__ src, int64_t n, const float *__restrict__ tgt,
                                                               const double *__restrict__ T, const int32_t *__restrict__ done,
                                                               const int32_t *__restrict__ row_of, const int32_t *__restrict__ prev,
                                                               double max_d2, const double *__restrict__ tbbox,
                                                               double *__restrict__ init_val, int32_t *__restrict__ init_idx,
                                                               double *__restrict__ A64, double *__restrict__ K64)
{
    if (done && *done) return;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t i = row_of[r];
    double s[3];
    xform_row(T, src + 3 * i, s);
    const double seed = row_seed(s);
    reinterpret_cast<double2 *>(A64)[2 * r] = make_double2(s[0], s[1]);
    reinterpret_cast<double2 *>(A64)[2 * r + 1] = make_double2(s[2], 1.0);
    K64[r] = seed;
    double bv = INFINITY;
    int32_t bj = INT_MAX;
    if (prev) {
        const int32_t j = prev[i];
        if (j >= 0 && j != INT_MAX) {
            const float *tp = tgt + 3 * (int64_t)j;
            const double tx = tp[0], ty = tp[1], tz = tp[2];
            const double t2 = fma(tx, tx, fma(ty, ty, tz * tz));
            double d = fma(s[0], -2.0 * tx, seed);
            d = fma(s[1], -2.0 * ty, d);
            d = fma(s[2], -2.0 * tz, d);
            bv = fma(1.0, t2, d);
            bj = j;
        }
    }
    if (max_d2 > 0.0) {
        double t2max = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) t2max += fmax(tbbox[a] * tbbox[a], tbbox[3 + a] * tbbox[3 + a]);
        const double clamp = (max_d2 + 1.0) * (1.0 + 9.31322574615478515625e-10) + ldexp(seed + t2max + 1.0, -38);
        if (!(bv <= clamp)) { bv = clamp; bj = INT_MAX; }
    }
    init_val[r] = bv;
    init_idx[r] = bj;
}

__device__ __forceinline__ double box_gap2(const double slo[3], const double shi[3], const float *__restrict__ bx)
{
    double g2 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double g = fmax(0.0, fmax((double)bx[a] - shi[a], slo[a] - (double)bx[3 + a]));
        g2 = fma(g, g, g2);
    }
    return g2;
}
// OR of the four 16-lane fields of a ballot
__device__ __forceinline__ unsigned fold16(unsigned long long m) { return (unsigned)((m | (m >> 16) | (m >> 32) | (m >> 48)) & 0xFFFFull); }

// Per-lane state of one wave's 16 rows.  Lane (q = lane>>4, j = lane&15) owns rows q, q+4, q+8, q+12 (the MFMA D
// layout) and, in the culling tests, box j of the 16 groups / tiles under test.
struct WaveRows {
    double a;                  // A operand: component q of row j
    d4 seed;                   // C operand: K of the lane's rows
    const double *rows;        // LDS: the wave's 16 transformed rows, row r at rows[5 r .. 5 r + 2] (x, y, z).  The culling tests
                               // read the lane's four rows (q, q+4, q+8, q+12) from here instead of holding 24 registers for them:
                               // the kernel then fits 128 VGPRs = 4 waves per SIMD, and a latency-bound sweep lives on resident waves
    double best[4];            // running minimum (per lane: over the columns j of the tiles seen)
    int32_t bcol[4];           // its ORIGINAL target index
};
constexpr int kRowStride = 5;

// The culled sweep of one wave: on return best/bcol hold, in every lane, the row minimum (lexicographic (value,
// original column)).  list: kLList ints of LDS owned by this wave.  Returns the number of tiles multiplied.
// Group boxes the caller loaded ahead of time: box of group 64 t + lane in pre[t] (t < kGroupPre; an empty box beyond n_groups).
// The loads do not depend on the transform, so the ICP kernel issues them before its update algebra and the first
// level of the culling finds them in registers instead of waiting a memory round trip.
constexpr int kGroupPre = 1;
struct GroupPre {
    float b[kGroupPre][6];
};
__device__ __forceinline__ void group_pre_load(GroupPre &g, const float *__restrict__ group_box, int32_t n_groups, int lane)
{
#pragma unroll
    for (int t = 0; t < kGroupPre; ++t) {
        const int gi = 64 * t + lane;
        const bool on = gi < n_groups;
        const float *bx = group_box + (int64_t)(on ? gi : 0) * 6;
#pragma unroll
        for (int a = 0; a < 3; ++a) { g.b[t][a] = on ? bx[a] : kBoxBig; g.b[t][3 + a] = on ? bx[3 + a] : -kBoxBig; }
    }
}
__device__ __forceinline__ double box_gap2v(const double slo[3], const double shi[3], const float bx[6])
{
    double g2 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double g = fmax(0.0, fmax((double)bx[a] - shi[a], slo[a] - (double)bx[3 + a]));
        g2 = fma(g, g, g2);
    }
    return g2;
}

template <bool PRE>
__device__ __forceinline__ unsigned sweep_wave(WaveRows &w, const double *__restrict__ Bs, const int32_t *__restrict__ orig,
                                               const float *__restrict__ tile_box, const float *__restrict__ group_box,
                                               int32_t n_groups, const double *__restrict__ tbbox, int32_t *list, const GroupPre *pre)
{
    const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
    constexpr double kRel = 1.0 + 9.31322574615478515625e-10;      // 1 + 2^-30
    double slo[3], shi[3];
    {
        double mn = w.a, mx = w.a;
#pragma unroll
        for (int msk = 1; msk < 16; msk <<= 1) {
            mn = fmin(mn, __shfl_xor(mn, msk, 64));
            mx = fmax(mx, __shfl_xor(mx, msk, 64));
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { slo[k] = __shfl(mn, 16 * k, 64); shi[k] = __shfl(mx, 16 * k, 64); }
    }
    double kmax = wave_all_max(fmax(fmax(w.seed[0], w.seed[1]), fmax(w.seed[2], w.seed[3])));
    double t2max = 0.0;
#pragma unroll
    for (int a3 = 0; a3 < 3; ++a3) t2max += fmax(tbbox[a3] * tbbox[a3], tbbox[3 + a3] * tbbox[3 + a3]);
    const double eps = ldexp(kmax + t2max + 1.0, -38);
    // rb[r]: the row's bound on d^2 (same value in the 16 lanes of a quad); R2: the largest of them
    double rb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rb[r] = (w.best[r] - 1.0) * kRel + eps;             // +inf stays +inf
    double R2 = wave_all_max(fmax(fmax(rb[0], rb[1]), fmax(rb[2], rb[3])));
    if (!(R2 < 1e290)) {
        // some row has no finite bound: every group holds a real point, so the distance to the farthest corner of
        // the nearest group box bounds that row's nearest-neighbour distance
        double u[4] = { INFINITY, INFINITY, INFINITY, INFINITY };
        for (int g0 = 0; g0 < n_groups; g0 += 16) {
            const int g = g0 + j;
            if (g < n_groups) {
                double lo[3], hi[3];
                load_box(group_box + (int64_t)g * 6, lo, hi);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double *pr = w.rows + kRowStride * (q + 4 * r);
                    u[r] = fmin(u[r], pt_far2(pr[0], pr[1], pr[2], lo, hi));
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int msk = 1; msk < 16; msk <<= 1) u[r] = fmin(u[r], __shfl_xor(u[r], msk, 64));
            rb[r] = fmin(rb[r], u[r] * kRel + eps);
        }
        R2 = wave_all_max(fmax(fmax(rb[0], rb[1]), fmax(rb[2], rb[3])));
    }

    const double bpad = q == 3 ? kSentinel : 0.0;
    int nlist = 0;
    unsigned visited = 0;

    auto process = [&]() {
        wave_lds_fence();                                  // the list writes before the reads
        bool updated = false;
        double b[4];
        int32_t t[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            t[h] = h < nlist ? list[h] : -1;
            b[h] = t[h] >= 0 ? Bs[(int64_t)t[h] * 64 + lane] : bpad;
        }
        for (int e = 0; e < nlist; e += 4) {
            const d4 c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w.a, b[0], w.seed, 0, 0, 0);
            const d4 c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w.a, b[1], w.seed, 0, 0, 0);
            const d4 c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w.a, b[2], w.seed, 0, 0, 0);
            const d4 c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(w.a, b[3], w.seed, 0, 0, 0);
            const int32_t tc[4] = { t[0], t[1], t[2], t[3] };
#pragma unroll
            for (int h = 0; h < 4; ++h) {                  // operands of the next trip
                t[h] = e + 4 + h < nlist ? list[e + 4 + h] : -1;
                b[h] = t[h] >= 0 ? Bs[(int64_t)t[h] * 64 + lane] : bpad;
            }
            bool pass = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned hb = hi32(w.best[r]);
                pass |= (bool)((int)(hi32(c0[r]) <= hb) | (int)(hi32(c1[r]) <= hb) | (int)(hi32(c2[r]) <= hb) | (int)(hi32(c3[r]) <= hb));
            }
            if (__builtin_amdgcn_ballot_w64(pass) != 0) {
                updated = true;
                int32_t oc[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) oc[h] = tc[h] >= 0 ? orig[(int64_t)tc[h] * 16 + j] : INT_MAX;
#define KPX_NNL_EXACT(ACC, COL)                                                                     \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                    \
                    const bool tk = (int)(ACC[r] < w.best[r]) | ((int)(ACC[r] == w.best[r]) & (int)((COL) < w.bcol[r])); \
                    w.best[r] = tk ? ACC[r] : w.best[r];                                           \
                    w.bcol[r] = tk ? (COL) : w.bcol[r];                                            \
                }
                KPX_NNL_EXACT(c0, oc[0]) KPX_NNL_EXACT(c1, oc[1]) KPX_NNL_EXACT(c2, oc[2]) KPX_NNL_EXACT(c3, oc[3])
#undef KPX_NNL_EXACT
            }
        }
        visited += (unsigned)nlist;
        nlist = 0;
        if (updated) {
            // tighten the row bounds with the best value any of the row's 16 lanes holds
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = w.best[r];
#pragma unroll
                for (int msk = 1; msk < 16; msk <<= 1) v = fmin(v, __shfl_xor(v, msk, 64));
                rb[r] = fmin(rb[r], (v - 1.0) * kRel + eps);
            }
            R2 = wave_all_max(fmax(fmax(rb[0], rb[1]), fmax(rb[2], rb[3])));
        }
        wave_lds_fence();
    };
    auto any_row_within_v = [&](const float bx[6]) {
        const double lo[3] = { (double)bx[0], (double)bx[1], (double)bx[2] }, hi[3] = { (double)bx[3], (double)bx[4], (double)bx[5] };
        bool t = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double *pr = w.rows + kRowStride * (q + 4 * r);
            t |= pt_gap2(pr[0], pr[1], pr[2], lo, hi) <= rb[r];
        }
        return t;
    };
    auto any_row_within = [&](const float *__restrict__ bx) {
        const float v[6] = { bx[0], bx[1], bx[2], bx[3], bx[4], bx[5] };
        return any_row_within_v(v);
    };
    auto append_tiles = [&](int grp, bool hit) {
        const unsigned tm16 = fold16(__builtin_amdgcn_ballot_w64(hit));
        if (q == 0 && ((tm16 >> j) & 1u)) list[nlist + __builtin_popcount(tm16 & ((1u << j) - 1u))] = grp * kLGroupTiles + j;
        nlist += __builtin_popcount(tm16);
    };
    // one 64-group trip: gmask = groups whose box the wave's box can reach
    auto trip = [&](int g0, unsigned long long gmask) {
        for (int c = 0; c < 4; ++c) {
            const unsigned chunk = (unsigned)(gmask >> (16 * c)) & 0xFFFFu;
            if (!chunk) continue;
            // per-row test of the 16 groups of this chunk (lane: rows of quad q against group j)
            const int gj = g0 + 16 * c + j;
            const bool gt = ((chunk >> j) & 1u) && any_row_within(group_box + (int64_t)gj * 6);
            unsigned gm16 = fold16(__builtin_amdgcn_ballot_w64(gt));
            while (gm16) {
                // the tile boxes of TWO surviving groups are fetched per round trip (a wave usually keeps one or two groups)
                const int grp0 = g0 + 16 * c + __builtin_ctz(gm16);
                gm16 &= gm16 - 1;
                const bool two = gm16 != 0;
                const int grp1 = two ? g0 + 16 * c + __builtin_ctz(gm16) : grp0;
                gm16 &= gm16 - 1;
                const float *p0 = tile_box + (int64_t)(grp0 * kLGroupTiles + j) * 6, *p1 = tile_box + (int64_t)(grp1 * kLGroupTiles + j) * 6;
                const float v0[6] = { p0[0], p0[1], p0[2], p0[3], p0[4], p0[5] }, v1[6] = { p1[0], p1[1], p1[2], p1[3], p1[4], p1[5] };
                append_tiles(grp0, any_row_within_v(v0));
                if (two) append_tiles(grp1, any_row_within_v(v1));
                if (nlist > kLList - 32) process();
            }
        }
    };
    int g_first = 0;
    if (PRE) {
#pragma unroll
        for (int t = 0; t < kGroupPre; ++t) {
            if (64 * t < n_groups) {
                const unsigned long long gmask = __builtin_amdgcn_ballot_w64(box_gap2v(slo, shi, pre->b[t]) <= R2);   // empty boxes: gap = inf
                if (gmask) trip(64 * t, gmask);
            }
        }
        g_first = 64 * kGroupPre;
    }
    for (int g0 = g_first; g0 < n_groups; g0 += 64) {
        const int g = g0 + lane;
        const bool gp = g < n_groups && box_gap2(slo, shi, group_box + (int64_t)g * 6) <= R2;
        const unsigned long long gmask = __builtin_amdgcn_ballot_w64(gp);
        if (gmask) trip(g0, gmask);
    }
    if (nlist) process();

    // reduce over the 16 lanes that hold the same rows (lexicographic (value, original column) minimum)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double v = w.best[r];
        int32_t c = w.bcol[r];
#pragma unroll
        for (int msk = 1; msk < 16; msk <<= 1) {
            const double ov = __shfl_xor(v, msk, 64);
            const int32_t oc = __shfl_xor(c, msk, 64);
            const bool take = ov < v || (ov == v && oc < c);
            v = take ? ov : v;
            c = take ? oc : c;
        }
        w.best[r] = v;
        w.bcol[r] = c;
    }
    return visited;
}

// One wave (= one block) per 16 sorted rows.  out_val / out_idx are indexed by ORIGINAL row.
__global__ __launch_bounds__(64) void nn_local_kernel(int64_t n, const double *__restrict__ Bs, const int32_t *__restrict__ orig,
                                                      const float *__restrict__ tile_box, const float *__restrict__ group_box,
                                                      int32_t n_groups, const double *__restrict__ tbbox, const int32_t *__restrict__ done,
                                                      const double *__restrict__ A64, const double *__restrict__ K64,
                                                      const double *__restrict__ init_val, const int32_t *__restrict__ init_idx,
                                                      const int32_t *__restrict__ row_of, double *__restrict__ out_val,
                                                      int32_t *__restrict__ out_idx, unsigned long long *__restrict__ tile_visits)
{
    if (done && *done) return;
    __shared__ int32_t list[kLList];
    __shared__ double rows[kLRows * kRowStride];
    const int lane = threadIdx.x, q = lane >> 4, j = lane & 15;
    const int64_t row_base = (int64_t)blockIdx.x * kLRows;
    const int64_t last = n - 1;
    WaveRows w;
    // rows past the end repeat the last row: same points, never written
    const int64_t arow = row_base + j < last ? row_base + j : last;
    w.a = A64[arow * 4 + q];
    if (q < 3) rows[kRowStride * j + q] = w.a;
    wave_lds_fence();
    w.rows = rows;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row_base + q + 4 * r < last ? row_base + q + 4 * r : last;
        w.seed[r] = K64[row];
        w.best[r] = init_val[row];
        w.bcol[r] = init_idx[row];
    }
    const unsigned visited = sweep_wave<false>(w, Bs, orig, tile_box, group_box, n_groups, tbbox, list, nullptr);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row_base + q + 4 * r;
        if (j == 0 && row < n) {
            const int64_t i = row_of[row];
            out_val[i] = w.best[r];
            out_idx[i] = w.bcol[r];
        }
    }
    if (tile_visits && lane == 0) atomicAdd(tile_visits + (blockIdx.x & (kVisitSlots - 1)), (unsigned long long)visited);
}

}  // namespace kpx
