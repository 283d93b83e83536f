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
//     neither the minimum nor a tie can hide there).  Surviving tiles go to a list in LDS and are multiplied with their
//     operands requested kMulBatch at a time, behind the same 32-bit high-word prefilter as the dense sweep (per tile).
//     (The frame loop hands its clouds over already Z-ordered -- kpx_voxel.hip -- and the sort is skipped: presorted.)
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
constexpr int kLList = 320;                      // tile list capacity (LDS, per wave): 64 carried over + 16 groups x 16 tiles
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

// OR of the four 16-lane fields of a ballot
__device__ __forceinline__ unsigned fold16(unsigned long long m) { return (unsigned)((m | (m >> 16) | (m >> 32) | (m >> 48)) & 0xFFFFull); }

// Cross-lane moves inside a 16-lane row on the DPP path of the VALU (row_ror:n): a few cycles, where __shfl_xor goes through
// the LDS crossbar (ds_bpermute, ~100 cycles of latency per butterfly step).  Rotations by 1, 2, 4, 8 leave the reduction of the
// whole row in EVERY lane of the row.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    typedef int i2 __attribute__((ext_vector_type(2)));
    i2 b = __builtin_bit_cast(i2, v);
    b[0] = dpp_i32<CTRL>(b[0]);
    b[1] = dpp_i32<CTRL>(b[1]);
    return __builtin_bit_cast(double, b);
}
constexpr int kRor1 = 0x121, kRor2 = 0x122, kRor4 = 0x124, kRor8 = 0x128;
__device__ __forceinline__ double row16_all_max(double v)
{
    v = fmax(v, dpp_f64<kRor1>(v)); v = fmax(v, dpp_f64<kRor2>(v)); v = fmax(v, dpp_f64<kRor4>(v)); v = fmax(v, dpp_f64<kRor8>(v));
    return v;
}
__device__ __forceinline__ double row16_all_min(double v)
{
    v = fmin(v, dpp_f64<kRor1>(v)); v = fmin(v, dpp_f64<kRor2>(v)); v = fmin(v, dpp_f64<kRor4>(v)); v = fmin(v, dpp_f64<kRor8>(v));
    return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    typedef int i2 __attribute__((ext_vector_type(2)));
    i2 b = __builtin_bit_cast(i2, v);
    b[0] = __builtin_amdgcn_readlane(b[0], l);
    b[1] = __builtin_amdgcn_readlane(b[1], l);
    return __builtin_bit_cast(double, b);
}
// maximum over the wave, as a wave-uniform value
__device__ __forceinline__ double wave_uniform_max(double v)
{
    v = row16_all_max(v);
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// the same for a value every 16-lane row already holds uniformly (per-row quantities of the MFMA D layout): no reduction inside the rows
__device__ __forceinline__ double quad_uniform_max(double v)
{
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// Directed roundings of a double to float (the culling tests below run in float32 on outward-rounded operands: a cull only has to
// be SAFE -- every decision that selects a partner is still taken on exact fp64 values -- and an fp64 vector instruction costs twice a
// float32 one on this part, plus a conversion per box coordinate).
__device__ __forceinline__ float f32_down(double x)
{
    float f = (float)x;
    if ((double)f > x) f = f > 0.0f ? __int_as_float(__float_as_int(f) - 1) : (f < 0.0f ? __int_as_float(__float_as_int(f) + 1) : -1.401298464e-45f);
    return f;
}
__device__ __forceinline__ float f32_up(double x)
{
    float f = (float)x;
    if ((double)f < x) f = f > 0.0f ? __int_as_float(__float_as_int(f) + 1) : (f < 0.0f ? __int_as_float(__float_as_int(f) - 1) : 1.401298464e-45f);
    return f;
}
// LOWER bound of the squared distance between a point known as [plo, phi] per axis (its coordinates rounded down / up to float) and
// the box [lo, hi]: per axis max(0, lo - phi, plo - hi) <= the true gap up to the subtraction's rounding (2^-24 relative), the three
// squares and two additions another 3 x 2^-24; the factor 1 - 2^-20 covers all of it.  Underflow only lowers the result further.
__device__ __forceinline__ float pt_gap2_lb(const float *__restrict__ pf, const float lo[3], const float hi[3])
{
    const float gx = fmaxf(0.0f, fmaxf(lo[0] - pf[3], pf[0] - hi[0]));
    const float gy = fmaxf(0.0f, fmaxf(lo[1] - pf[4], pf[1] - hi[1]));
    const float gz = fmaxf(0.0f, fmaxf(lo[2] - pf[5], pf[2] - hi[2]));
    return fmaf(gz, gz, fmaf(gy, gy, gx * gx)) * 0.99999904632568359375f;
}

// Per-lane state of one wave's 16 rows.  Lane (q = lane>>4, j = lane&15) owns rows q, q+4, q+8, q+12 (the MFMA D layout).
struct WaveRows {
    double a;                  // A operand: component q of row j
    d4 seed;                   // C operand: K of the lane's rows
    double *rows;              // LDS: the wave's 16 transformed rows, record r at rows[kRowStride r ..]: x, y, z, then the row's current
                               // bound on d^2 (kRowBound, written by the sweep).  The culling tests read rows from here instead of
                               // holding them in registers (the iteration kernel is built for 168 VGPRs = 3 waves per SIMD, KPX_ICP_WPE;
                               // a latency-bound sweep lives on resident waves)
    float *rowsf;              // LDS: float32 mirror for the culling tests, record r at rowsf[kRowFStride r ..]: the row's coordinates rounded
                               // down (0..2) and up (3..5), its bound rounded up (6); written by the sweep itself
    double best[4];            // running minimum (per lane: over the columns j of the tiles seen)
    int32_t bcol[4];           // its ORIGINAL target index
    double light_gap2;         // out: no group box was within reach of the wave's box -> the smallest squared box-to-box gap; else -1
    // ---- certificates (sweep_wave<PRE, true>, the ICP kernels; see icp_iter_body) ----
    double rb0[4];             // in: the bound on d^2 each of the lane's rows is searched with (>= its partner's; larger = a skin around it);
                               //     < 0: the row is certified, it takes no part in the culling
    double skin;               // in: wave-uniform; a bound tightened inside the sweep keeps this much distance beyond the new partner
    unsigned act_mask;         // in: bit j = row j takes part (wave-uniform)
    unsigned sec[4];           // out (all 16 lanes of a row): high word of a LOWER bound of the smallest D among the multiplied columns other than
                               //     the row's winner (0xFFFFFFFF: none was multiplied)
    double rb_out[4], eps_out; // out: the rows' final culling bounds (every culled column has d^2 > rb_out) and the metric's rounding margin
    unsigned long long *dbg;   // LOOPED sweeps only (KPX_ICP_CHAIN_STAMPS): 100 MHz stamps of this wave's sweep, or nullptr
};
constexpr int kRowStride = 6, kRowBound = 3;
constexpr int kRowFStride = 8, kRowFBound = 6;

// LDS scratch of one wave (ints): the tile list, 16 candidate groups (box + id), up to 16 surviving groups (id, row mask)
constexpr int kLCand = 16, kLSurv = 16;
constexpr int kLScratch = kLList + 7 * kLCand + 2 * kLSurv;
#ifndef KPX_MUL_BATCH
#define KPX_MUL_BATCH 8
#endif
constexpr int kMulBatch = KPX_MUL_BATCH;         // column tiles whose operands are requested together
#ifndef KPX_TIGHTEN_GAIN
#define KPX_TIGHTEN_GAIN 0.5
#endif
constexpr double kTightenGain = KPX_TIGHTEN_GAIN; // multiply(): bounds are tightened when some row's would fall below this share of its value

// Group boxes the caller loaded ahead of time: box of group 64 t + lane in pre[t] (t < kGroupPre; an empty box beyond n_groups).
// The loads do not depend on the transform, so the ICP kernel issues them first and the culling finds them in registers
// instead of waiting a memory round trip.
constexpr int kGroupPre = 2;
struct GroupPre {
    float b[kGroupPre][6];
};
__device__ __forceinline__ void group_box_load(float b[6], const float *__restrict__ group_box, int32_t n_groups, int gi)
{
    const bool on = gi < n_groups;
    const float *bx = group_box + (int64_t)(on ? gi : 0) * 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) { b[a] = on ? bx[a] : kBoxBig; b[3 + a] = on ? bx[3 + a] : -kBoxBig; }
}
__device__ __forceinline__ void group_pre_load(GroupPre &g, const float *__restrict__ group_box, int32_t n_groups, int lane)
{
#pragma unroll
    for (int t = 0; t < kGroupPre; ++t) group_box_load(g.b[t], group_box, n_groups, 64 * t + lane);
}
// largest |t|^2 a target point can have (from the target's bounding box): scales the rounding margin of the expanded metric
__device__ __forceinline__ double target_t2max(const double *__restrict__ tbbox)
{
    double t2max = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) t2max += fmax(tbbox[a] * tbbox[a], tbbox[3 + a] * tbbox[3 + a]);
    return t2max;
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

__device__ __forceinline__ double wave_uniform_min(double v)
{
    v = row16_all_min(v);
    return fmin(fmin(readlane_f64(v, 0), readlane_f64(v, 16)), fmin(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// The culled sweep of one wave: on return best/bcol hold, in every lane, the row minimum (lexicographic (value, original
// column)).  scr: kLScratch ints of LDS owned by this wave; t2max: target_t2max(tbbox), read by the caller at its very start so
// that the load is in flight with everything else.  Returns the number of tiles multiplied in bits 0..15 and, for the
// probes, the dependent memory round trips of the wave: tile-box fetches (bits 16..31), operand fetches of the multiply loop
// (bits 32..47), groups that survived the per-row test (bits 48..63).
//
// A sweep is a chain of DEPENDENT memory round trips (~1 us each on a loaded device) with little arithmetic between them, so
// it is organised to make few of them, each as wide as the registers allow:
//   G  group level, no memory: the wave's box against 64 group boxes per lane-parallel test (boxes in registers, loaded ahead);
//      the groups that pass are handed, 16 at a time through LDS, to the per-row test (lane (q, j): rows q+4r against candidate
//      j), which also yields the mask of rows that reach each surviving group;
//   T  ONE round trip for the tile boxes of up to 16 surviving groups (lane (k, j): tile j of group 4p + k, four passes whose
//      loads are issued together); a tile is tested against the rows of its group's mask only;
//   M  ONE round trip per kMulBatch listed tiles: B operands and original column ids requested together, then the fp64 MFMAs
//      behind the 32-bit high-word prefilter; the row bounds tighten afterwards (registers and LDS).
template <bool PRE, bool CERT = false, bool LOOPED = false>
__device__ __forceinline__ unsigned long long sweep_wave(WaveRows &w, const double *__restrict__ Bs, const int32_t *__restrict__ orig,
                                                         const float *__restrict__ tile_box, const float *__restrict__ group_box,
                                                         int32_t n_groups, const double t2max, int32_t *scr, const GroupPre *pre)
{
    // (LOOPED = called inside a loop, icp_chain_kernel: the lane number behind an opaque move keeps what is derived from it from being
    // hoisted out of that loop)
    const int lane = LOOPED ? opaque_i((int)(threadIdx.x & 63)) : (int)(threadIdx.x & 63), q = lane >> 4, j = lane & 15;
    const unsigned long long lt = (1ull << lane) - 1ull;
    auto dbg_tick = [&](int slot) { if (LOOPED && w.dbg && lane == 0) w.dbg[slot] = wall_clock64(); };
    dbg_tick(0);
    int32_t *list = scr, *cand = scr + kLList, *surv = scr + kLList + 7 * kLCand;
    // records of the lane's four rows q, q + 4, q + 8, q + 12: ONE base each, the rows at constant offsets (immediates of the ds instructions)
    double *const rowq = w.rows + kRowStride * q;
    float *const rowfq = w.rowsf + kRowFStride * q;
    constexpr double kRel = 1.0 + 9.31322574615478515625e-10;      // 1 + 2^-30
    double slo[3], shi[3];
    {
        // (CERT: the box of the rows that take part -- a certified row neither reaches anything nor widens the coarse test)
        const bool act_j = !CERT || ((w.act_mask >> j) & 1u) != 0u;
        const double mn = row16_all_min(act_j ? w.a : INFINITY), mx = row16_all_max(act_j ? w.a : -INFINITY);
#pragma unroll
        for (int k = 0; k < 3; ++k) { slo[k] = readlane_f64(mn, 16 * k); shi[k] = readlane_f64(mx, 16 * k); }
    }
    const double kmax = quad_uniform_max(fmax(fmax(w.seed[0], w.seed[1]), fmax(w.seed[2], w.seed[3])));      // seeds and bounds: one value per row
    const double eps = ldexp(kmax + t2max + 1.0, -38);
    // rb[r]: the row's bound on d^2 (same value in the 16 lanes of a quad); R2: the largest of them
    double rb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rb[r] = (CERT ? w.rb0[r] : w.best[r] - 1.0) * kRel + eps;             // +inf stays +inf
    if (CERT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) w.sec[r] = 0xFFFFFFFFu;
    }
    double R2 = quad_uniform_max(fmax(fmax(rb[0], rb[1]), fmax(rb[2], rb[3])));
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
        for (int r = 0; r < 4; ++r) rb[r] = fmin(rb[r], row16_all_min(u[r]) * kRel + eps);
        R2 = quad_uniform_max(fmax(fmax(rb[0], rb[1]), fmax(rb[2], rb[3])));
    }
    // The bounds live in LDS from here on (fp64 in the row records, rounded up to float in the mirror the culling tests read): the
    // multiply loop needs every register it can get, and the bounds are touched again only when a trip tightens them.
    auto publish_bounds = [&](const double rbv[4]) {
        if (j == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rowq[4 * kRowStride * r + kRowBound] = rbv[r];
                rowfq[4 * kRowFStride * r + kRowFBound] = f32_up(rbv[r]);
            }
        }
    };
    if (q < 3) {               // lane (q, j): component q of row j, rounded outward
        w.rowsf[kRowFStride * j + q] = f32_down(w.a);
        w.rowsf[kRowFStride * j + 3 + q] = f32_up(w.a);
    }
    publish_bounds(rb);
    dbg_tick(1);

    const double bpad = q == 3 ? kSentinel : 0.0;
    int nlist = 0, ns = 0;
    unsigned visited = 0, box_trips = 0, mul_trips = 0, groups_kept = 0;

    // M: multiply the listed tiles
    auto multiply = [&]() {
        wave_lds_fence();                                  // the list writes before the reads
        bool updated = false;
        for (int e0 = 0; e0 < nlist; e0 += kMulBatch) {
            double b[kMulBatch];
            int32_t oc[kMulBatch];
#pragma unroll
            for (int h = 0; h < kMulBatch; ++h) {
                const int32_t t = e0 + h < nlist ? list[e0 + h] : -1;
                b[h] = t >= 0 ? Bs[(int64_t)t * 64 + lane] : bpad;
                oc[h] = t >= 0 ? orig[(int64_t)t * 16 + j] : INT_MAX;
            }
            ++mul_trips;
            if (mul_trips == 1) dbg_tick(5);
#pragma unroll
            for (int h0 = 0; h0 < kMulBatch; h0 += 4) {
                if (e0 + h0 >= nlist) break;
                const d4 c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w.a, b[h0], w.seed, 0, 0, 0);
                const d4 c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w.a, b[h0 + 1], w.seed, 0, 0, 0);
                const d4 c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w.a, b[h0 + 2], w.seed, 0, 0, 0);
                const d4 c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(w.a, b[h0 + 3], w.seed, 0, 0, 0);
#define KPX_NNL_TILE(ACC, COL)                                                                      \
                {                                                                                   \
                    bool pass = false;                                                              \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) pass |= hi32(ACC[r]) <= hi32(w.best[r]); \
                    if (__builtin_amdgcn_ballot_w64(pass) != 0) {                                   \
                        updated = true;                                                             \
                        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                            \
                            const bool tk = (int)(ACC[r] < w.best[r]) | ((int)(ACC[r] == w.best[r]) & (int)((COL) < w.bcol[r])); \
                            if (CERT) {     /* the loser is a runner-up candidate, unless it is the winner's own column met again */ \
                                const unsigned lose = tk ? hi32(w.best[r]) : ((COL) == w.bcol[r] ? 0xFFFFFFFFu : hi32(ACC[r])); \
                                w.sec[r] = lose < w.sec[r] ? lose : w.sec[r];                       \
                            }                                                                       \
                            w.best[r] = tk ? ACC[r] : w.best[r];                                   \
                            w.bcol[r] = tk ? (COL) : w.bcol[r];                                    \
                        }                                                                           \
                    } else if (CERT) {      /* nothing near any row's best: four v_min_u32 keep the runner-up bound */ \
                        _Pragma("unroll") for (int r = 0; r < 4; ++r) w.sec[r] = hi32(ACC[r]) < w.sec[r] ? hi32(ACC[r]) : w.sec[r]; \
                    }                                                                               \
                }
                KPX_NNL_TILE(c0, oc[h0]) KPX_NNL_TILE(c1, oc[h0 + 1]) KPX_NNL_TILE(c2, oc[h0 + 2]) KPX_NNL_TILE(c3, oc[h0 + 3])
#undef KPX_NNL_TILE
            }
        }
        if (mul_trips <= 2) dbg_tick(6);
        visited += (unsigned)nlist;
        nlist = 0;
        // tighten the row bounds with the best value any of the row's 16 lanes holds -- when that pays: the reductions below cost
        // ~0.2 us per trip, and a sweep seeded with last iteration's partners finds bounds it can hardly improve (the partner's own
        // distance); only a row whose bound would shrink to kTightenGain of its value (a cold start, a new and much nearer partner)
        // prunes enough of what is left of the sweep.  A bound that is not tightened stays valid.
        bool gain = false;
        // the bound a row would get: its best (CERT: plus the skin -- any value >= best - 1 is a valid culling bound, so the float
        // square root only has to be rounded upward once)
        auto next_bound = [&](int r) {
            double nb = w.best[r] - 1.0;
            if (CERT && w.skin > 0.0) {
                const double rr = (double)(sqrtf(f32_up(fmax(nb, 0.0))) * 1.0000002384185791015625f) + w.skin;
                nb = fmax(nb, rr * rr);
            }
            return nb;
        };
        if (updated) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gain |= next_bound(r) < kTightenGain * (double)rowfq[4 * kRowFStride * r + kRowFBound];
        }
        if (updated && __builtin_amdgcn_ballot_w64(gain) != 0) {
            double nrb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) nrb[r] = fmin(rowq[4 * kRowStride * r + kRowBound], row16_all_min(next_bound(r)) * kRel + eps);
            R2 = quad_uniform_max(fmax(fmax(nrb[0], nrb[1]), fmax(nrb[2], nrb[3])));
            wave_lds_fence();                                  // every lane has read the old bounds
            publish_bounds(nrb);
        }
        wave_lds_fence();
        dbg_tick(7);
    };

    // T: the tile boxes of the surviving groups surv[0 .. ns) -> tile list
    auto tiles_of_survivors = [&]() {
        wave_lds_fence();
        float bx[4][6];
        int32_t grp[4];
        unsigned rmask[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int k = 4 * p + q;
            const bool on = k < ns;
            grp[p] = on ? surv[2 * k] : 0;
            rmask[p] = on ? (unsigned)surv[2 * k + 1] : 0u;
            if (4 * p < ns) {                                                    // wave-uniform
                const float *tb = tile_box + ((int64_t)grp[p] * kLGroupTiles + j) * 6;
#pragma unroll
                for (int a = 0; a < 6; ++a) bx[p][a] = tb[a];
            }
        }
        ++box_trips;
        if (box_trips == 1) dbg_tick(3);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (4 * p >= ns) break;
            const float lo[3] = { bx[p][0], bx[p][1], bx[p][2] }, hi[3] = { bx[p][3], bx[p][4], bx[p][5] };
            // the rows any of the pass's four groups needs, two per trip (wave-uniform loop, broadcast LDS reads issued together)
            unsigned um = (unsigned)(__builtin_amdgcn_readlane((int)rmask[p], 0) | __builtin_amdgcn_readlane((int)rmask[p], 16) |
                                     __builtin_amdgcn_readlane((int)rmask[p], 32) | __builtin_amdgcn_readlane((int)rmask[p], 48));
            bool hit = false;
            while (um) {
                const int r0 = __builtin_ctz(um);
                um &= um - 1u;
                const int r1 = um ? __builtin_ctz(um) : r0;
                um &= um - 1u;
                const float *p0 = w.rowsf + kRowFStride * r0, *p1 = w.rowsf + kRowFStride * r1;
                float f0[8], f1[8];
#pragma unroll
                for (int e = 0; e < 7; ++e) { f0[e] = p0[e]; f1[e] = p1[e]; }
                hit |= (bool)((int)((rmask[p] >> r0) & 1u) & (int)(pt_gap2_lb(f0, lo, hi) <= f0[kRowFBound]));
                hit |= (bool)((int)((rmask[p] >> r1) & 1u) & (int)(pt_gap2_lb(f1, lo, hi) <= f1[kRowFBound]));
            }
            const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
            if (hit) list[nlist + __builtin_popcountll(hm & lt)] = grp[p] * kLGroupTiles + j;
            nlist += __builtin_popcountll(hm);
        }
        ns = 0;
        if (box_trips == 1) dbg_tick(4);
    };

    // G: lane holds the box of group gbase + lane; gmask = groups the wave's box can reach
    int g_next = 0, gbase = 0;
    unsigned long long gmask = 0ull;
    double gap_min = INFINITY;
    bool reached = false;
    float box[6] = { kBoxBig, kBoxBig, kBoxBig, -kBoxBig, -kBoxBig, -kBoxBig };
    for (;;) {
        while (ns < kLSurv && (gmask != 0ull || g_next < n_groups)) {
            if (gmask == 0ull) {
                gbase = g_next;
                g_next += 64;
                if (PRE && gbase == 0) {
#pragma unroll
                    for (int a = 0; a < 6; ++a) box[a] = pre->b[0][a];
                } else if (PRE && kGroupPre > 1 && gbase == 64) {
#pragma unroll
                    for (int a = 0; a < 6; ++a) box[a] = pre->b[kGroupPre > 1 ? 1 : 0][a];
                } else {
                    group_box_load(box, group_box, n_groups, gbase + lane);
                }
                const double gp = box_gap2v(slo, shi, box);                                   // empty boxes: a huge gap
                gap_min = fmin(gap_min, gp);
                gmask = __builtin_amdgcn_ballot_w64(gp <= R2);
                reached = reached || gmask != 0ull;
                continue;
            }
            // hand the next (at most kLSurv - ns) reachable groups to the per-row test through LDS
            const int room = kLSurv - ns;
            const int slot = __builtin_popcountll(gmask & lt);
            const bool mine = ((gmask >> lane) & 1ull) != 0ull && slot < room;
            if (mine) {
#pragma unroll
                for (int a = 0; a < 6; ++a) cand[7 * slot + a] = __float_as_int(box[a]);
                cand[7 * slot + 6] = gbase + lane;
            }
            const unsigned long long taken = __builtin_amdgcn_ballot_w64(mine);
            const int nc = __builtin_popcountll(taken);
            gmask &= ~taken;
            wave_lds_fence();
            bool h[4] = { false, false, false, false };
            int32_t gid = -1;
            if (j < nc) {
                const float lo[3] = { __int_as_float(cand[7 * j]), __int_as_float(cand[7 * j + 1]), __int_as_float(cand[7 * j + 2]) };
                const float hi[3] = { __int_as_float(cand[7 * j + 3]), __int_as_float(cand[7 * j + 4]), __int_as_float(cand[7 * j + 5]) };
                gid = cand[7 * j + 6];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float *pr = rowfq + 4 * kRowFStride * r;
                    float f[7];
#pragma unroll
                    for (int e = 0; e < 7; ++e) f[e] = pr[e];
                    h[r] = pt_gap2_lb(f, lo, hi) <= f[kRowFBound];
                }
            }
            // mask of the rows (bit q + 4 r) that reach candidate j, assembled from the four ballots
            unsigned rm = 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned long long x = __builtin_amdgcn_ballot_w64(h[r]) >> j;
                rm |= (unsigned)((x & 1ull) | ((x >> 15) & 2ull) | ((x >> 30) & 4ull) | ((x >> 45) & 8ull)) << (4 * r);
            }
            const bool sv = q == 0 && rm != 0u;
            const unsigned long long sm = __builtin_amdgcn_ballot_w64(sv);
            if (sv) {
                const int k = ns + __builtin_popcountll(sm & lt);
                surv[2 * k] = gid;
                surv[2 * k + 1] = (int32_t)rm;
            }
            ns += __builtin_popcountll(sm);
            groups_kept += (unsigned)__builtin_popcountll(sm);
            wave_lds_fence();                              // the candidate slots are reused by the next pass
        }
        const bool last = gmask == 0ull && g_next >= n_groups;
        if (box_trips == 0) dbg_tick(2);
        if (ns) tiles_of_survivors();
        if (nlist > kLList - 16 * kLSurv || (last && nlist)) multiply();
        if (last) break;
    }

    dbg_tick(8);
    w.light_gap2 = reached ? -1.0 : wave_uniform_min(gap_min);
    // reduce over the 16 lanes that hold the same rows (lexicographic (value, original column) minimum)
#define KPX_NNL_ROWMIN(CTRL)                                               \
    {                                                                      \
        const double ov = dpp_f64<CTRL>(v);                                \
        const int32_t oc = dpp_i32<CTRL>(c);                               \
        const bool take = ov < v || (ov == v && oc < c);                   \
        v = take ? ov : v;                                                 \
        c = take ? oc : c;                                                 \
    }
    // (a wave that multiplied nothing -- half of them -- still holds what it was given: one value per row in all 16 lanes)
    if (visited != 0u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = w.best[r];
            int32_t c = w.bcol[r];
            KPX_NNL_ROWMIN(kRor1) KPX_NNL_ROWMIN(kRor2) KPX_NNL_ROWMIN(kRor4) KPX_NNL_ROWMIN(kRor8)
            if (CERT) {
                // runner-up bound of the row: every lane's own, and the best of every lane that does not hold the winner (the lanes still
                // holding the partner the row came with hold ONE column, the same in all of them)
                unsigned u = w.bcol[r] == c ? w.sec[r] : (hi32(w.best[r]) < w.sec[r] ? hi32(w.best[r]) : w.sec[r]);
                unsigned o;
                o = (unsigned)dpp_i32<kRor1>((int)u); u = o < u ? o : u;
                o = (unsigned)dpp_i32<kRor2>((int)u); u = o < u ? o : u;
                o = (unsigned)dpp_i32<kRor4>((int)u); u = o < u ? o : u;
                o = (unsigned)dpp_i32<kRor8>((int)u); u = o < u ? o : u;
                w.sec[r] = u;
            }
            w.best[r] = v;
            w.bcol[r] = c;
        }
    }
    if (CERT) {
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) w.rb_out[r] = rowq[4 * kRowStride * r + kRowBound];
        w.eps_out = eps;
    }
    dbg_tick(9);
#undef KPX_NNL_ROWMIN
    return (unsigned long long)(visited & 0xFFFFu) | ((unsigned long long)(box_trips & 0xFFFFu) << 16) | ((unsigned long long)(mul_trips & 0xFFFFu) << 32) |
           ((unsigned long long)(groups_kept & 0xFFFFu) << 48);
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
    __shared__ int32_t list[kLScratch];
    __shared__ double rows[kLRows * kRowStride];
    __shared__ float rowsf[kLRows * kRowFStride];
    const int lane = threadIdx.x, q = lane >> 4, j = lane & 15;
    const double t2max = target_t2max(tbbox);
    const int64_t row_base = (int64_t)blockIdx.x * kLRows;
    const int64_t last = n - 1;
    WaveRows w;
    // rows past the end repeat the last row: same points, never written
    const int64_t arow = row_base + j < last ? row_base + j : last;
    w.a = A64[arow * 4 + q];
    if (q < 3) rows[kRowStride * j + q] = w.a;
    wave_lds_fence();
    w.rows = rows;
    w.rowsf = rowsf;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row_base + q + 4 * r < last ? row_base + q + 4 * r : last;
        w.seed[r] = K64[row];
        w.best[r] = init_val[row];
        w.bcol[r] = init_idx[row];
    }
    const unsigned visited = (unsigned)(sweep_wave<false>(w, Bs, orig, tile_box, group_box, n_groups, t2max, list, nullptr) & 0xFFFFu);
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
