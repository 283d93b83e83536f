// kpx_icp.hip -- register stage: registration_icp (preprocessing/registration.py:78-84 point-to-plane,
// manual_pointcloud_registration.py:90-98 point-to-point + Kabsch from picked pairs).
//
// Correspondence search = all-pairs nearest neighbour as an fp64 MFMA distance GEMM in the K=4 augmented
// form (contract AC2):
//     A[i] = (s_x, s_y, s_z, 1)            s = T . src_i       (fp64 fma chain, contract AC1)
//     B[j] = (-2t_x, -2t_y, -2t_z, |t|^2)  |t|^2 = fma(tx,tx, fma(ty,ty, tz*tz))
//     C[i] = K_i = fma(sx,sx, fma(sy,sy, sz*sz)) + 1
//     D_ij = fma(1,|t|^2, fma(s_z,-2t_z, fma(s_y,-2t_y, fma(s_x,-2t_x, K_i))))  = d_ij^2 + 1 > 0
//            -> argmin_j, ties to the lowest j.
// v_mfma_f64_16x16x4_f64 produces a 16x16 tile of D per instruction (bit-for-bit the k-ordered fma chain
// above, seeded with C).  D > 0, so the IEEE bit pattern orders like an unsigned integer: the running
// argmin behind each MFMA is a 32-bit compare of the HIGH words (hi(D) <= hi(best): a necessary condition
// for an update) and a wave-uniform branch; the exact fp64 (value, column) update runs only in the rare
// wave-iterations where some lane passes.  To make updates rare the sweep starts from a valid upper bound:
// the previous iteration's partner (ICP iterations >= 1) or the winner of a seed sweep over every 64th
// target tile.  fp64 VALU compares contend with the fp64 MFMA pipe on MI355X (measured: 4 v_cmp_f64 per
// MFMA cost 30 % of the MFMA rate), the 32-bit prefilter does not.
// Block = 4 waves x 32 source rows; the B stream is staged through LDS by LDS-DMA (16 KiB stages, double
// buffered) and shared by the waves; the column range is split over gridDim.y, a second kernel merges the
// splits (lexicographic (value, column)), computes the direct squared distance (AC3) of the chosen pair
// and accumulates the sums the update needs.  The ICP loop runs on the device; a small kernel solves the
// 3x3 (Kabsch) or 6x6 (point-to-plane) system, updates T and raises `done`.
#include <chrono>
#include <mutex>
#include <fcntl.h>
#include <memory>
#include <condition_variable>
#include <thread>
#include <atomic>
#include <sys/file.h>
#include <unistd.h>
#include <limits.h>
#include <stddef.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>

#include "kpx_internal.h"
#include "kpx_linalg.h"
#include "kpx_fixed.h"

namespace kpx {

typedef double d4 __attribute__((ext_vector_type(4)));
#ifndef KPX_ICP_ACQ_FENCE
#define KPX_ICP_ACQ_FENCE 1                      // the block that performs the update acquires at agent scope (one buffer_inv per registration and launch)
#endif
#ifndef KPX_ICP_SPLIT_DEFAULT
#define KPX_ICP_SPLIT_DEFAULT 2
#endif
#ifndef KPX_ICP_STATE_LDS
#define KPX_ICP_STATE_LDS 1                      // the last-block update works on an LDS copy of the registration's state (A/B: 0 = in global memory)
#endif
#ifndef KPX_ICP_WPE
#define KPX_ICP_WPE 3                            // waves per SIMD the iteration kernels are register-budgeted for
#endif

constexpr int kRT = 2;                       // 16-row tiles per wave (the sweep below is written for 2)
constexpr int kWaves = 4;
constexpr int kRowsPerBlock = kWaves * kRT * 16;   // 128
constexpr int kCT = 32;                      // 16-column tiles per LDS stage (16 KiB)
#ifndef KPX_NN_CHUNK
#define KPX_NN_CHUNK 4                       // measured 100k x 100k inside a registration: 2 -> 38.7, 4 -> 39.1, 8 -> 36.5, 16 -> 35.9 TFLOP/s
#endif

constexpr int kChunk = KPX_NN_CHUNK;         // column tiles per fast-pass chunk of the dense sweep (nn_mfma_kernel)
constexpr int kStageDoubles = kCT * 64;
#ifndef KPX_SEED_STRIDE
#define KPX_SEED_STRIDE 128                 // round 4, operands in curve order, seed = every n-th POINT of the curve: whole bare search 100k x 100k at
                                            // 8 / 16 / 32 / 64 / 128 / 256 -> 0.42 / 0.47 / 0.49 / 0.51 / 0.515 / 0.52 of the fp64 matrix peak (the
                                            // main sweep no longer cares how loose the bound is: a row reaches only the chunks around it).
                                            // Until round 3 (operands in the caller's order, every n-th TILE): 64 -> 2.60 ms, 16 -> 2.48, 8 -> 2.66
#endif
constexpr int kSeedStride = KPX_SEED_STRIDE;  // the seed sweep visits every kSeedStride-th target tile
constexpr double kSentinel = 1e300;
constexpr int kFRT = 4;                      // f32 screening sweep: 16-row tiles per wave
constexpr int kFRowsPerBlock = kWaves * kFRT * 16; // 256
constexpr int kFCT = 64;                     // f32 tiles per LDS stage (16 KiB)
constexpr int kFStageFloats = kFCT * 64;
constexpr int kCand = 64;                    // candidate slots per source row
constexpr int kAcc = 44;                     // accumulator slots: count, sum d2, sum s, sum t, sum t s^T, J^T J (21), J^T r (6)

struct IcpState {
    double T[16];
    double fitness, rmse;
    double count;
    int32_t iter, done;
    double motion, reach;      // see LightSkip: accumulated bound on how far any source point has moved; reach of a row's search
    double last_motion;        // what the latest update added to `motion` (row certificates: how calm the registration is)
    double smax;               // largest |T p| over the source's bounding box under the CURRENT T (the next update's lever arm); < 0: not known yet
};

__device__ __forceinline__ void xform_row(const double *__restrict__ T, const float *__restrict__ p, double s[3])
{
    const double x = p[0], y = p[1], z = p[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = fma(T[4 * k], x, fma(T[4 * k + 1], y, fma(T[4 * k + 2], z, T[4 * k + 3])));
}
__device__ __forceinline__ double row_seed(const double s[3]) { return fma(s[0], s[0], fma(s[1], s[1], s[2] * s[2])) + 1.0; }
__device__ __forceinline__ int opaque_i(int v) { asm volatile("" : "+v"(v)); return v; }
// bits 33..39 of a progress word: the share of the registration's rows the iteration searched, in 1/127ths rounded up (0 = none)
__host__ __device__ __forceinline__ unsigned long long progress_searched(unsigned long long searched, int64_t n)
{
    const unsigned long long cls = n > 0 ? (searched * 127ull + (unsigned long long)n - 1ull) / (unsigned long long)n : 0ull;
    return (cls > 127ull ? 127ull : cls) << 33;
}
// the searched-row counts of a registration: eight words behind its ticket, read (device-coherent) and cleared by the block that drew the
// last ticket; every lane < 8 of the calling wave takes one word, the total comes back in every lane of that wave's first 8-lane group
constexpr int kSearchedWord = 8;
__device__ __forceinline__ unsigned long long searched_take(unsigned long long *ticket, int lane_in_block)
{
    unsigned long long v = 0ull;
    if (lane_in_block < 8) {
        v = __hip_atomic_load(ticket + kSearchedWord + lane_in_block, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ticket + kSearchedWord + lane_in_block, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane_in_block < 64) {                              // (wave 0: an 8-lane butterfly; the other waves of a block do not publish)
        unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const unsigned ol = (unsigned)__shfl_xor((int)lo, o, 64), oh = (unsigned)__shfl_xor((int)hi, o, 64);
            const unsigned long long sum = (((unsigned long long)hi << 32) | lo) + (((unsigned long long)oh << 32) | ol);
            lo = (unsigned)sum; hi = (unsigned)(sum >> 32);
        }
        v = ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
// balanced tree over 16 adjacent values: (((x0+x1)+(x2+x3))+((x4+x5)+(x6+x7))) + (the same over x8..x15) -- the order in which four
// butterfly steps (lane ^ 1, lane ^ 2, half-row mirror, row mirror) add the 16 lanes of a DPP row (row16_tree_sum, kpx_icprows.h)
__device__ __forceinline__ double tile_tree16(const double *x)
{
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = x[2 * i] + x[2 * i + 1];
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
typedef unsigned u2 __attribute__((ext_vector_type(2)));
// high word of the IEEE pattern as a 32-bit register reference (a shift of the 64-bit pattern makes hipcc
// compare zero-extended 64-bit values, i.e. the slow v_cmp_*_u64 this prefilter exists to avoid)
__device__ __forceinline__ unsigned hi32(double v) { return __builtin_bit_cast(u2, v)[1]; }

// ---- target preparation: B tiles, element (k, j) of tile t at B[t*64 + k*16 + j]; Bseed = every 64th tile --
__global__ __launch_bounds__(256) void nn_prep_kernel(const float *__restrict__ tgt, int64_t m, int64_t tiles_pad, double *__restrict__ B,
                                                      int64_t seed_tiles_pad, double *__restrict__ Bseed, const int32_t *__restrict__ perm,
                                                      int32_t *__restrict__ colB, int32_t *__restrict__ colSeed)
{
    // perm (round 4): the columns stand in the target's CURVE order (perm[slot] = the caller's index of the point in column `slot`), so
    // that the 64 columns of a chunk are neighbours in space and a row's bound -- however loose -- reaches few chunks; colB / colSeed carry
    // every column's ORIGINAL index (INT_MAX in the padding): the sweep reports, and breaks ties by, those.
    const int64_t total = (tiles_pad + seed_tiles_pad) * 16;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const bool seed = q >= tiles_pad * 16;
        const int64_t jj = seed ? q - tiles_pad * 16 : q;                 // column slot inside its operand array
        const int64_t j = seed ? jj * kSeedStride : jj;                    // (curve-ordered) target slot it stands for: the seed operand is every
                                                                            // kSeedStride-th POINT of the curve, a spatially uniform sample
        double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = kSentinel;
        int32_t oj = INT_MAX;
        if (j < m) {
            oj = perm ? perm[j] : (int32_t)j;
            double tx = tgt[3 * (int64_t)oj], ty = tgt[3 * (int64_t)oj + 1], tz = tgt[3 * (int64_t)oj + 2];
            b0 = -2.0 * tx; b1 = -2.0 * ty; b2 = -2.0 * tz;
            b3 = fma(tx, tx, fma(ty, ty, tz * tz));
        }
        double *o = (seed ? Bseed : B) + (jj >> 4) * 64 + (jj & 15);
        o[0] = b0; o[16] = b1; o[32] = b2; o[48] = b3;
        (seed ? colSeed : colB)[jj] = oj;
    }
}

// ---- per-row operands of one search: transformed source, row seed, bound from a known partner, f32 screening row --
// One thread per source row, once per search (the sweeps are split over the columns: computing these in their
// prologues would repeat the fp64 work in every split).
//   A64[row] = (s_x, s_y, s_z, 1)   K64[row] = K_i            -> fp64 sweep operands
//   prev != NULL: init_val/init_idx = exact D(i, prev[i]) by the MFMA's fma chain, and its partner
//   aux  != NULL (needs prev): A32[row] = fl32(s - c, 1), thr32[row] = (C_i, round_up((U_i - 1 - |s-c|^2) + C_i + E_i)),
//                              C_i = |s-c|^2 + E_i + 1 makes the f32 metric positive (integer compares)
struct NnAux {
    double c[3];      // centre used for the f32 operands
    double rt2;       // >= max_j |t_j - c|^2
};
__global__ __launch_bounds__(256) void nn_rowprep_kernel(const float *__restrict__ src, int64_t n, const float *__restrict__ tgt,
                                                         const double *__restrict__ T, const int32_t *__restrict__ done,
                                                         const int32_t *__restrict__ prev, const NnAux *__restrict__ aux,
                                                         double *__restrict__ init_val, int32_t *__restrict__ init_idx,
                                                         double *__restrict__ A64, double *__restrict__ K64,
                                                         float *__restrict__ A32, float *__restrict__ thr32,
                                                         int32_t *__restrict__ cand_cnt)
{
    if (done && *done) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) cand_cnt[n] = 0;          // number of rows whose candidate list overflowed
    if (i >= n) return;
    cand_cnt[i] = 0;
    double s[3];
    xform_row(T, src + 3 * i, s);
    const double seed = row_seed(s);
    reinterpret_cast<double2 *>(A64)[2 * i] = make_double2(s[0], s[1]);
    reinterpret_cast<double2 *>(A64)[2 * i + 1] = make_double2(s[2], 1.0);
    K64[i] = seed;
    if (!prev) return;
    const int32_t j = prev[i];
    const float *tp = tgt + 3 * (int64_t)j;
    const double tx = tp[0], ty = tp[1], tz = tp[2];
    const double t2 = fma(tx, tx, fma(ty, ty, tz * tz));
    double d = fma(s[0], -2.0 * tx, seed);
    d = fma(s[1], -2.0 * ty, d);
    d = fma(s[2], -2.0 * tz, d);
    d = fma(1.0, t2, d);
    init_val[i] = d;
    init_idx[i] = j;
    if (!aux) return;
    const double ux = s[0] - aux->c[0], uy = s[1] - aux->c[1], uz = s[2] - aux->c[2];
    const double q = fma(ux, ux, fma(uy, uy, uz * uz));
    // E_i bounds |D32 - exact|: operand roundings 2^-24 (6X + Y) + four chain roundings of partial sums <= C + X + Y,
    // X = 2|s-c||t-c| <= 2 sqrt(q rt2), Y = |t-c|^2 <= rt2, C ~ q + E + 1  (10 % and 1e-6 slack for second-order terms)
    const double x = 2.0 * sqrt(q * aux->rt2), y = aux->rt2;
    const double e = 1.1 * 5.9604644775390625e-08 * (10.0 * x + 9.0 * y + 4.0 * (q + 2.0)) * (1.0 + 1e-6) + 1e-6;
    const float cq = __double2float_ru(q + e + 1.0);                 // row constant: D32 = cq + approx >= 1 > 0
    reinterpret_cast<float4 *>(A32)[i] = make_float4((float)ux, (float)uy, (float)uz, 1.0f);
    thr32[2 * i] = cq;
    thr32[2 * i + 1] = __double2float_ru(((d - 1.0 - q) + (double)cq) + e);
}

// ---- float32 screening sweep (ICP iterations >= 1) ------------------------------------------------------------
// With a valid upper bound U_i (the exact D of last iteration's partner under the new transform) the exact
// argmin only needs the columns whose D can be <= U_i.  Those are found by ONE sweep of the f32 MFMA
// (v_mfma_f32_16x16x4_f32, ~3x the fp64 MFMA rate) on centred coordinates:
//     approx_ij = fl32 chain of (s-c) . (-2(t-c)) + |t-c|^2  ~  d_ij^2 - |s_i-c|^2
// with the rigorous error bound  |approx - exact| <= E_i = 2^-24 (12 |s_i-c| R_t + 5 R_t^2)  (two roundings of
// every operand, four chain roundings; R_t >= max |t-c|).  Every column with approx_ij <= (U_i - 1 - |s_i-c|^2) + E_i
// (+10 % and 1e-6 slack) is appended to row i's candidate list; nn_merge_kernel then evaluates the exact fp64
// metric (the same fma chain as the MFMA path / the oracle) for the candidates and the bound's partner and takes
// the lexicographic (value, column) minimum -- the result is bit-identical to the fp64 sweep.  Rows whose list
// overflows are resolved by an exact brute-force scan (nn_overflow_kernel).
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void nn_aux_kernel(const double *__restrict__ bbox, NnAux *aux)
{
    if (threadIdx.x || blockIdx.x) return;
    double r2 = 0.0;
    for (int a = 0; a < 3; ++a) {
        double c = rint(0.5 * (bbox[a] + bbox[3 + a]));
        aux->c[a] = c;
        double e = fmax(fabs(bbox[a] - c), fabs(bbox[3 + a] - c));
        r2 += e * e;
    }
    aux->rt2 = r2 * (1.0 + 1e-12);
}

// Bf tiles: element (k, j) of tile t at Bf[t*64 + k*16 + j] (float)
__global__ __launch_bounds__(256) void nn_prep_f32_kernel(const float *__restrict__ tgt, int64_t m, int64_t tiles_pad,
                                                          const NnAux *__restrict__ aux, float *__restrict__ Bf)
{
    const double cx = aux->c[0], cy = aux->c[1], cz = aux->c[2];
    const int64_t total = tiles_pad * 16;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 3.0e38f;
        if (j < m) {
            double ux = (double)tgt[3 * j] - cx, uy = (double)tgt[3 * j + 1] - cy, uz = (double)tgt[3 * j + 2] - cz;
            b0 = (float)(-2.0 * ux); b1 = (float)(-2.0 * uy); b2 = (float)(-2.0 * uz);
            b3 = (float)fma(ux, ux, fma(uy, uy, uz * uz));
        }
        float *o = Bf + (j >> 4) * 64 + (j & 15);
        o[0] = b0; o[16] = b1; o[32] = b2; o[48] = b3;
    }
}

template <int TRIP, int MINW>
__global__ __launch_bounds__(256, MINW) void nn_screen_kernel(int64_t n, const float *__restrict__ Bf, int32_t tiles_per_split,
                                                           const int32_t *__restrict__ done, const float *__restrict__ A32,
                                                           const float *__restrict__ thr32, const int32_t *__restrict__ partner,
                                                           int32_t *__restrict__ cand_cnt, int32_t *__restrict__ cand,
                                                           int32_t *__restrict__ over_rows)
{
    if (done && *done) return;
    __shared__ __align__(16) float lds[2][kFStageFloats];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row_base = (int64_t)blockIdx.x * kFRowsPerBlock + (int64_t)wave * (kFRT * 16);
    const int64_t t0 = (int64_t)blockIdx.y * tiles_per_split;
    const int nstages = tiles_per_split / kFCT;

    // A operands: component k = lane>>4 of centred row (lane&15); thresholds of the rows this lane sees in D
    // (f32 layout: row = 4*(lane>>4) + reg)
    // the bound's own partner always passes the test and is always evaluated by nn_merge_kernel: it is not
    // appended (in steady state it is ~99 % of the hits, and an append costs a returning global atomic)
    float a[kFRT];
    f4 cq[kFRT];
    unsigned thr[kFRT][4];
#pragma unroll
    for (int rt = 0; rt < kFRT; ++rt) {
        const int64_t row = row_base + rt * 16 + (lane & 15);
        a[rt] = row < n ? A32[row * 4 + (lane >> 4)] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t drow = row_base + rt * 16 + 4 * (lane >> 4) + r;
            const float2 ct = drow < n ? reinterpret_cast<const float2 *>(thr32)[drow] : make_float2(1.0f, 0.0f);
            cq[rt][r] = ct.x;
            thr[rt][r] = __float_as_uint(ct.y);          // D32 > 0 and thr >= 0: unsigned order of the patterns
        }
    }

    const float *gB = Bf + t0 * 64;
    auto stage_load = [&](int stage, int buf) {
        const float *g = gB + (int64_t)stage * kFStageFloats;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = wave * 4 + q;      // 16 pieces of 1 KiB (= 4 tiles) per 16 KiB stage
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + piece * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void *)(&lds[buf][piece * 256]), 16, 0, 0);
        }
    };
    stage_load(0, 0);
    __syncthreads();

    for (int st = 0; st < nstages; ++st) {
        const int buf = st & 1;
        if (st + 1 < nstages) stage_load(st + 1, buf ^ 1);
        const float *lb = lds[buf] + lane;
        const int32_t tile0 = (int32_t)t0 + st * kFCT;
#pragma unroll 1
        for (int ct = 0; ct < kFCT; ct += TRIP) {
            // TRIP column tiles (TRIP x 4 MFMAs) per trip; per D row the minimum of the TRIP bit patterns
            // (v_min3_u32 / v_min_u32) and one unsigned compare against the row's threshold
            float bq[TRIP];
#pragma unroll
            for (int h = 0; h < TRIP; ++h) bq[h] = lb[(ct + h) * 64];
            f4 c[TRIP][kFRT];
#pragma unroll
            for (int rt = 0; rt < kFRT; ++rt)
#pragma unroll
                for (int h = 0; h < TRIP; ++h) c[h][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt], bq[h], cq[rt], 0, 0, 0);
            bool hit = false;
#pragma unroll
            for (int rt = 0; rt < kFRT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    unsigned m = __float_as_uint(c[0][rt][r]);
#pragma unroll
                    for (int h = 1; h < TRIP; ++h) m = min(m, __float_as_uint(c[h][rt][r]));
                    hit |= m <= thr[rt][r];
                }
            if (__builtin_amdgcn_ballot_w64(hit) != 0) {          // wave-uniform; a few columns per row per sweep
                const int32_t col0 = (tile0 + ct) * 16 + (lane & 15);
#pragma unroll
                for (int rt = 0; rt < kFRT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t row = row_base + rt * 16 + 4 * (lane >> 4) + r;
#pragma unroll
                        for (int h = 0; h < TRIP; ++h)
                            if (__float_as_uint(c[h][rt][r]) <= thr[rt][r] && col0 + h * 16 != partner[row]) {
                                const int slot = atomicAdd(&cand_cnt[row], 1);
                                if (slot < kCand) cand[row * kCand + slot] = col0 + h * 16;
                                else if (slot == kCand) over_rows[atomicAdd(&cand_cnt[n], 1)] = (int32_t)row;   // first overflow of this row
                            }
                    }
            }
        }
        __syncthreads();
    }
}

// rows whose candidate list overflowed (listed by the screening sweep): exact brute-force scan (fp64 fma chain),
// one block per row at a time.  With nothing listed the kernel returns at once.
__global__ __launch_bounds__(256) void nn_overflow_kernel(const float *__restrict__ src, int64_t n, const float *__restrict__ tgt, int64_t m,
                                                          const double *__restrict__ T, const int32_t *__restrict__ done,
                                                          int32_t *__restrict__ cand_cnt, int32_t *__restrict__ cand,
                                                          const int32_t *__restrict__ over_rows)
{
    if (done && *done) return;
    const int total = cand_cnt[n];
    __shared__ double sv[256];
    __shared__ int sj[256];
    for (int e = blockIdx.x; e < total; e += gridDim.x) {
        const int64_t row = over_rows[e];
        double s[3];
        xform_row(T, src + 3 * row, s);
        const double seed = row_seed(s);
        double bv = INFINITY;
        int bj = INT_MAX;
        for (int64_t j = threadIdx.x; j < m; j += 256) {
            const double tx = tgt[3 * j], ty = tgt[3 * j + 1], tz = tgt[3 * j + 2];
            double d = fma(s[0], -2.0 * tx, seed);
            d = fma(s[1], -2.0 * ty, d);
            d = fma(s[2], -2.0 * tz, d);
            d = fma(1.0, fma(tx, tx, fma(ty, ty, tz * tz)), d);
            if (d < bv) { bv = d; bj = (int)j; }           // ascending j per thread: first minimum kept
        }
        sv[threadIdx.x] = bv; sj[threadIdx.x] = bj;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) {
                double ov = sv[threadIdx.x + w]; int oj = sj[threadIdx.x + w];
                if (ov < sv[threadIdx.x] || (ov == sv[threadIdx.x] && oj < sj[threadIdx.x])) { sv[threadIdx.x] = ov; sj[threadIdx.x] = oj; }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) { cand[row * kCand] = sj[0]; cand_cnt[row] = 1; }
        __syncthreads();
    }
}

// ---- the MFMA nearest-neighbour sweep ------------------------------------------------------------------
// colid: the ORIGINAL target index of every column of B (the full operand or the seed operand: every kSeedStride-th tile), staged in LDS beside the tiles
// FAST: the rows arrive with TIGHT bounds (the previous partner under the new transform: every ICP iteration after the first) -- the
// stage is swept in chunks whose hot loop is MFMAs + one v_min_u32 per result register, and a chunk is swept again the exact way
// only when some row's smallest high word reaches its bound (measured 100k x 100k: 50.6 TFLOP/s for the hot loop alone = 0.64 of the
// 78.6 vendor peak, the instruction's measured issue ceiling; the per-trip prefilter form runs at 31).  !FAST: loose bounds (seed
// sweep, first search) -- nearly every chunk would be swept twice, so every trip is examined behind the prefilter as it comes.
template <bool FAST>
__global__ __launch_bounds__(256, 4) void nn_mfma_kernel(int64_t n, const double *__restrict__ B, const int32_t *__restrict__ colid, int32_t tiles_per_split,
                                                         const int32_t *__restrict__ done,
                                                         const double *__restrict__ A64, const double *__restrict__ K64,
                                                         const double *__restrict__ init_val, const int32_t *__restrict__ init_idx,
                                                         double *__restrict__ part_val, int32_t *__restrict__ part_idx, const int32_t *__restrict__ rperm)
{
    // rperm (round 4): block row r is the caller's row rperm[r] -- the source's curve order, so that the 32 rows of a wave are neighbours
    // in space and reach the SAME few chunks of the (curve-ordered) columns; operands and results stay indexed by the caller's row
    if (done && *done) return;
    __shared__ __align__(16) double lds[2][kStageDoubles];
    __shared__ __align__(16) int32_t lds_col[2][kCT * 16];           // the stage's ORIGINAL column indices (the operand stands in curve order)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row_base = (int64_t)blockIdx.x * kRowsPerBlock + (int64_t)wave * (kRT * 16);
    const int split = blockIdx.y;
    const int64_t t0 = (int64_t)split * tiles_per_split;
    const int nstages = tiles_per_split / kCT;

    // A operands: lane holds component k = lane>>4 of row (lane&15) of each of its row tiles
    double a[kRT];
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt) {
        const int64_t row = row_base + rt * 16 + (lane & 15);
        a[rt] = row < n ? A64[(int64_t)(rperm ? rperm[row] : row) * 4 + (lane >> 4)] : 0.0;
    }
    // C operands (row seeds K_i), running best and its column: D layout row = (lane>>4) + 4*reg
    d4 seed[kRT];
    double best[kRT][4];
    int32_t bcol[kRT][4];
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row_base + rt * 16 + (lane >> 4) + 4 * r;
            double kk = 1.0, bv = INFINITY;
            int32_t bj = INT_MAX;
            if (row < n) {
                const int64_t ri = rperm ? rperm[row] : row;
                kk = K64[ri];
                if (init_val) { bv = init_val[ri]; bj = init_idx[ri]; }
            }
            seed[rt][r] = kk; best[rt][r] = bv; bcol[rt][r] = bj;
        }

    // B stream: global -> LDS by LDS-DMA (1 KiB per wave-instruction, 16 pieces per 16 KiB stage)
    const double *gB = B + t0 * 64;
    const int32_t *gC = colid + t0 * 16;
    auto stage_load = [&](int stage, int buf) {
        const double *g = gB + (int64_t)stage * kStageDoubles;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = wave * 4 + q;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + piece * 128 + lane * 2),
                                             (__attribute__((address_space(3))) void *)(&lds[buf][piece * 128]), 16, 0, 0);
        }
        // 512 column ids = 2 KiB: waves 0 and 1, one 16-byte piece per lane
        if (wave < 2)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gC + (int64_t)stage * (kCT * 16) + wave * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void *)(&lds_col[buf][wave * 256]), 16, 0, 0);
    };
    stage_load(0, 0);
    __syncthreads();

    for (int st = 0; st < nstages; ++st) {
        const int buf = st & 1;
        if (st + 1 < nstages) stage_load(st + 1, buf ^ 1);
        const double *lb = lds[buf] + lane;
        // One trip = two column tiles x two row tiles = four MFMAs; examine() looks at a trip's 16 result registers: prefilter on the
        // high words (D > 0: the unsigned order of the bit patterns is the numeric order), exact (value, column) update only when
        // some lane passes (wave-uniform, rare once the bound is tight).
        auto examine = [&](const d4 &c00, const d4 &c10, const d4 &c01, const d4 &c11, const int ct) {
            bool pass = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned h0 = hi32(best[0][r]), h1 = hi32(best[1][r]);
                pass |= (bool)((int)(min(hi32(c00[r]), hi32(c01[r])) <= h0) | (int)(min(hi32(c10[r]), hi32(c11[r])) <= h1));
            }
            if (__builtin_amdgcn_ballot_w64(pass) == 0) return;
            const int32_t col0 = lds_col[buf][ct * 16 + (lane & 15)], col1 = lds_col[buf][ct * 16 + 16 + (lane & 15)];
            bool anyeq = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned h0 = hi32(best[0][r]), h1 = hi32(best[1][r]);
                anyeq |= (bool)((int)(hi32(c00[r]) == h0) | (int)(hi32(c01[r]) == h0) | (int)(hi32(c10[r]) == h1) | (int)(hi32(c11[r]) == h1));
                // two candidates of one row with equal high words: the second must be compared exactly with the first
                // once that has become the running best (found by the random cross-engine test: far-apart line clouds)
                anyeq |= (bool)((int)(hi32(c00[r]) == hi32(c01[r])) | (int)(hi32(c10[r]) == hi32(c11[r])));
            }
            if (__builtin_amdgcn_ballot_w64(anyeq) == 0) {
                // every high word differs from its bound: the high words alone decide "<" (no fp64 op)
#define KPX_NN_HI(ACC, RT, COL)                                                                     \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                    \
                    const bool t = hi32(ACC[r]) < hi32(best[RT][r]);                               \
                    best[RT][r] = t ? ACC[r] : best[RT][r];                                        \
                    bcol[RT][r] = t ? (COL) : bcol[RT][r];                                         \
                }
                KPX_NN_HI(c00, 0, col0) KPX_NN_HI(c01, 0, col1) KPX_NN_HI(c10, 1, col0) KPX_NN_HI(c11, 1, col1)
#undef KPX_NN_HI
            } else {
                // near-ties (equal high words, e.g. the bound's own column): exact lexicographic (value, column)
#define KPX_NN_EXACT(ACC, RT, COL)                                                                  \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                    \
                    const bool t = (int)(ACC[r] < best[RT][r]) | ((int)(ACC[r] == best[RT][r]) & (int)((COL) < bcol[RT][r])); \
                    best[RT][r] = t ? ACC[r] : best[RT][r];                                        \
                    bcol[RT][r] = t ? (COL) : bcol[RT][r];                                         \
                }
                KPX_NN_EXACT(c00, 0, col0) KPX_NN_EXACT(c01, 0, col1) KPX_NN_EXACT(c10, 1, col0) KPX_NN_EXACT(c11, 1, col1)
#undef KPX_NN_EXACT
            }
        };
#define KPX_NN_TRIP(P, CT)                                                                          \
        const double P##b0 = lb[(CT) * 64], P##b1 = lb[(CT) * 64 + 64];                             \
        const d4 P##00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], P##b0, seed[0], 0, 0, 0);       \
        const d4 P##10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], P##b0, seed[1], 0, 0, 0);       \
        const d4 P##01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], P##b1, seed[0], 0, 0, 0);       \
        const d4 P##11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], P##b1, seed[1], 0, 0, 0);
        if (!FAST) {
#pragma unroll 1
        for (int ct = 0; ct < kCT; ct += 2) {
            KPX_NN_TRIP(p, ct)
            examine(p00, p10, p01, p11, ct);
        }
        } else {
        // A stage is swept in chunks of kChunk column tiles.  FAST pass of a chunk: nothing but the MFMAs and one v_min_u32 per result
        // register -- the smallest HIGH WORD any column of the chunk produced for each of the lane's rows.  Only when some row's
        // minimum reaches the high word of its running best (hi(D) > hi(best) implies D > best, so a chunk that never does cannot
        // change any row's (value, column) minimum) is the chunk swept again the exact way.
#pragma unroll 1
        for (int c0 = 0; c0 < kCT; c0 += kChunk) {
            unsigned hmin[kRT][4];
#pragma unroll
            for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) hmin[rt][r] = 0xFFFFFFFFu;
#pragma unroll
            for (int ct = 0; ct < kChunk; ct += 2) {
                KPX_NN_TRIP(f, c0 + ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hmin[0][r] = min(hmin[0][r], min(hi32(f00[r]), hi32(f01[r])));
                    hmin[1][r] = min(hmin[1][r], min(hi32(f10[r]), hi32(f11[r])));
                }
            }
            bool pass = false;
#pragma unroll
            for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pass |= hmin[rt][r] <= hi32(best[rt][r]);
            if (__builtin_amdgcn_ballot_w64(pass) == 0) continue;
#pragma unroll 1
            for (int ct = c0; ct < c0 + kChunk; ct += 2) {
                KPX_NN_TRIP(p, ct)
                examine(p00, p10, p01, p11, ct);
            }
        }
        }
#undef KPX_NN_TRIP
        __syncthreads();
    }

    // reduce over the 16 lanes that hold the same rows (lexicographic (value, column) minimum)
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = best[rt][r];
            int32_t c = bcol[rt][r];
#pragma unroll
            for (int msk = 1; msk < 16; msk <<= 1) {
                double ov = __shfl_xor(v, msk, 64);
                int32_t oc = __shfl_xor(c, msk, 64);
                bool take = ov < v || (ov == v && oc < c);
                v = take ? ov : v;
                c = take ? oc : c;
            }
            int64_t row = row_base + rt * 16 + (lane >> 4) + 4 * r;
            if ((lane & 15) == 0 && row < n) {
                const int64_t ri = rperm ? rperm[row] : row;
                part_val[(int64_t)split * n + ri] = v;
                part_idx[(int64_t)split * n + ri] = c;
            }
        }
}

// ---- merge splits, direct distance, accumulation ---------------------------------------------------------
// mode: -2 = write (value, column) as the bound of the next sweep, -1 = correspondences only,
//        0 = point-to-point sums, 1 = + point-to-plane normal equations
//        2 = coloured ICP ([O3D] TransformationEstimationForColoredICP): the normal equations hold a geometric row
//            sqrt(lambda) (s x n, n | (s - t).n) and a photometric row sqrt(1 - lambda) (s x g', g' | I_s - (I_t + g.(s' - t))),
//            s' = s projected onto the target's tangent plane, g the target's colour gradient, g' = -(I - n n^T) g
struct ColorTerms {
    const float *src_col, *tgt_col;
    const double *tgt_grad;
    double sqrt_lg, sqrt_lp;
};
constexpr int kMergeThreads = 64;
__global__ __launch_bounds__(kMergeThreads) void nn_merge_kernel(const float *__restrict__ src, int64_t n, const float *__restrict__ tgt,
                                                       const float *__restrict__ tn, const double *__restrict__ T,
                                                       const int32_t *__restrict__ done, const double *__restrict__ part_val,
                                                       const int32_t *__restrict__ part_idx, int splits, double max_d2, int mode,
                                                       int32_t *__restrict__ idx_out, double *__restrict__ d2_out,
                                                       double *__restrict__ val_out, double *__restrict__ part_acc,
                                                       const int32_t *__restrict__ cand_cnt, const int32_t *__restrict__ cand, ColorTerms ct)
{
    if (done && *done) return;
    __shared__ double sh[kAcc][kMergeThreads + 1];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc[kAcc];
#pragma unroll
    for (int q = 0; q < kAcc; ++q) acc[q] = 0.0;
    if (i < n) {
        double bv = part_val[i];
        int32_t bj = part_idx[i];
        if (cand_cnt) {
            // screening path: (part_val, part_idx) hold the bound (exact D of last iteration's partner); the exact
            // metric of every screened candidate decides, by the same fma chain as the MFMA sweep
            double s[3];
            xform_row(T, src + 3 * i, s);
            const double seed = row_seed(s);
            int c = cand_cnt[i];
            c = c < kCand ? c : kCand;
            for (int e = 0; e < c; ++e) {
                const int32_t j = cand[i * kCand + e];
                const float *tp = tgt + 3 * (int64_t)j;
                const double tx = tp[0], ty = tp[1], tz = tp[2];
                double d = fma(s[0], -2.0 * tx, seed);
                d = fma(s[1], -2.0 * ty, d);
                d = fma(s[2], -2.0 * tz, d);
                d = fma(1.0, fma(tx, tx, fma(ty, ty, tz * tz)), d);
                if (d < bv || (d == bv && j < bj)) { bv = d; bj = j; }
            }
        } else {
            for (int s = 1; s < splits; ++s) {
                double v = part_val[(int64_t)s * n + i];
                int32_t j = part_idx[(int64_t)s * n + i];
                if (v < bv || (v == bv && j < bj)) { bv = v; bj = j; }
            }
        }
        const bool none = bj < 0 || bj == INT_MAX;        // culled sweep inside a registration: nothing within max_dist
        if (idx_out) idx_out[i] = none ? -1 : bj;
        if (mode == -2) {
            val_out[i] = bv;
        } else if (none) {
            if (d2_out) d2_out[i] = INFINITY;
        } else {
            double s[3];
            xform_row(T, src + 3 * i, s);
            const float *tp = tgt + 3 * (int64_t)bj;
            double t[3] = { (double)tp[0], (double)tp[1], (double)tp[2] };
            double dx = s[0] - t[0], dy = s[1] - t[1], dz = s[2] - t[2];
            double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
            if (d2_out) d2_out[i] = d2;
            if (mode >= 0 && d2 < max_d2) {
                acc[0] = 1.0; acc[1] = d2;
#pragma unroll
                for (int k = 0; k < 3; ++k) { acc[2 + k] = s[k]; acc[5 + k] = t[k]; }
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = 0; q < 3; ++q) acc[8 + 3 * p + q] = t[p] * s[q];
                if (mode == 1) {
                    const float *np_ = tn + 3 * (int64_t)bj;
                    double nx = np_[0], ny = np_[1], nz = np_[2];
                    double r = (s[0] - t[0]) * nx + (s[1] - t[1]) * ny + (s[2] - t[2]) * nz;
                    double J[6] = { s[1] * nz - s[2] * ny, s[2] * nx - s[0] * nz, s[0] * ny - s[1] * nx, nx, ny, nz };
                    int q = 17;
#pragma unroll
                    for (int p = 0; p < 6; ++p)
#pragma unroll
                        for (int c = p; c < 6; ++c) acc[q++] = J[p] * J[c];
#pragma unroll
                    for (int p = 0; p < 6; ++p) acc[38 + p] = J[p] * r;
                } else if (mode == 2) {
                    const float *np_ = tn + 3 * (int64_t)bj;
                    const double nv[3] = { np_[0], np_[1], np_[2] };
                    const double rg = (s[0] - t[0]) * nv[0] + (s[1] - t[1]) * nv[1] + (s[2] - t[2]) * nv[2];
                    const double is = ((double)ct.src_col[3 * i] + (double)ct.src_col[3 * i + 1] + (double)ct.src_col[3 * i + 2]) / 3.0;
                    const float *tc = ct.tgt_col + 3 * (int64_t)bj;
                    const double it = ((double)tc[0] + (double)tc[1] + (double)tc[2]) / 3.0;
                    const double *gp = ct.tgt_grad + 3 * (int64_t)bj;
                    const double g[3] = { gp[0], gp[1], gp[2] };
                    const double sp[3] = { s[0] - rg * nv[0], s[1] - rg * nv[1], s[2] - rg * nv[2] };
                    const double is0 = (g[0] * (sp[0] - t[0]) + g[1] * (sp[1] - t[1]) + g[2] * (sp[2] - t[2])) + it;
                    const double gn = g[0] * nv[0] + g[1] * nv[1] + g[2] * nv[2];
                    const double gm[3] = { -(g[0] - gn * nv[0]), -(g[1] - gn * nv[1]), -(g[2] - gn * nv[2]) };
                    const double JG[6] = { ct.sqrt_lg * (s[1] * nv[2] - s[2] * nv[1]), ct.sqrt_lg * (s[2] * nv[0] - s[0] * nv[2]),
                                           ct.sqrt_lg * (s[0] * nv[1] - s[1] * nv[0]), ct.sqrt_lg * nv[0], ct.sqrt_lg * nv[1], ct.sqrt_lg * nv[2] };
                    const double JI[6] = { ct.sqrt_lp * (s[1] * gm[2] - s[2] * gm[1]), ct.sqrt_lp * (s[2] * gm[0] - s[0] * gm[2]),
                                           ct.sqrt_lp * (s[0] * gm[1] - s[1] * gm[0]), ct.sqrt_lp * gm[0], ct.sqrt_lp * gm[1], ct.sqrt_lp * gm[2] };
                    const double rG = ct.sqrt_lg * rg, rI = ct.sqrt_lp * (is - is0);
                    int q = 17;
#pragma unroll
                    for (int p = 0; p < 6; ++p)
#pragma unroll
                        for (int c = p; c < 6; ++c) acc[q++] = JG[p] * JG[c] + JI[p] * JI[c];
#pragma unroll
                    for (int p = 0; p < 6; ++p) acc[38 + p] = JG[p] * rG + JI[p] * rI;
                }
            }
        }
    }
    if (mode < 0) return;
    // fixed-order block sums through LDS: slot q of lane l at sh[q][l] (row stride 65 doubles: conflict-free
    // column walks), lane q then adds its row in lane order -- no cross-lane shuffles (a 64-lane fp64 shuffle
    // tree for 44 slots costs ~500 ds_bpermutes per wave)
    const int nacc = mode >= 1 ? kAcc : 17;
#pragma unroll
    for (int q = 0; q < kAcc; ++q)
        if (q < 17 || mode >= 1) sh[q][threadIdx.x] = acc[q];
    __syncthreads();
    if ((int)threadIdx.x < nacc) {
        double v = 0.0;
        for (int l = 0; l < kMergeThreads; ++l) v += sh[threadIdx.x][l];
        part_acc[(int64_t)blockIdx.x * kAcc + threadIdx.x] = v;
    }
}

// ---- update step ------------------------------------------------------------------------------------------
__device__ void mat4_mul(const double A[16], const double B[16], double C[16])
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double v = 0.0;
            for (int k = 0; k < 4; ++k) v += A[4 * r + k] * B[4 * k + c];
            C[4 * r + c] = v;
        }
}
// Umeyama / Kabsch without scale from the sums (Eigen::umeyama, with_scaling = false)
__device__ void update_p2p(const double *acc, double U[16])
{
    for (int k = 0; k < 16; ++k) U[k] = (k % 5 == 0) ? 1.0 : 0.0;
    double cnt = acc[0];
    if (cnt < 1.0) return;
    double mu_s[3], mu_t[3], S[9], R[9];
    for (int k = 0; k < 3; ++k) { mu_s[k] = acc[2 + k] / cnt; mu_t[k] = acc[5 + k] / cnt; }
    for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q) S[3 * p + q] = acc[8 + 3 * p + q] / cnt - mu_t[p] * mu_s[q];
    kabsch_rotation(S, R);
    for (int p = 0; p < 3; ++p) {
        for (int q = 0; q < 3; ++q) U[4 * p + q] = R[3 * p + q];
        U[4 * p + 3] = mu_t[p] - (R[3 * p] * mu_s[0] + R[3 * p + 1] * mu_s[1] + R[3 * p + 2] * mu_s[2]);
    }
}
// point-to-plane: (J^T J) x = -J^T r ; T = [Rz(x2) Ry(x1) Rx(x0) | x3..5]
__device__ void update_p2plane(const double *acc, double U[16])
{
    for (int k = 0; k < 16; ++k) U[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (acc[0] < 1.0) return;
    double A[36], b[6], x[6];
    int q = 17;
    for (int p = 0; p < 6; ++p) for (int c = p; c < 6; ++c) { A[6 * p + c] = acc[q]; A[6 * c + p] = acc[q]; ++q; }
    for (int p = 0; p < 6; ++p) b[p] = -acc[38 + p];
    if (!solve6_ldlt(A, b, x)) return;
    double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
    // Rz(g) Ry(b) Rx(a)
    U[0] = cg * cb; U[1] = cg * sb * sa - sg * ca; U[2] = cg * sb * ca + sg * sa; U[3] = x[3];
    U[4] = sg * cb; U[5] = sg * sb * sa + cg * ca; U[6] = sg * sb * ca - cg * sa; U[7] = x[4];
    U[8] = -sb;     U[9] = cb * sa;                U[10] = cb * ca;               U[11] = x[5];
}

// fitness / rmse / convergence test / update of T from the accumulated sums (one thread).  k = index of the
// correspondence search the sums come from.
__device__ void icp_finish(const double *acc, int64_t n, int mode, int k, int max_iter, double rel_fit, double rel_rmse, IcpState *st,
                           double *__restrict__ result)
{
    double cnt = acc[0];
    double fit = (n > 0 && cnt > 0) ? cnt / (double)n : 0.0;
    double rmse = cnt > 0 ? sqrt(acc[1] / cnt) : 0.0;
    bool done = false;
    if (k >= 1 && fabs(st->fitness - fit) < rel_fit && fabs(st->rmse - rmse) < rel_rmse) done = true;
    st->fitness = fit; st->rmse = rmse; st->count = cnt; st->iter = k;
    if (k >= max_iter) done = true;
    if (!done) {
        double U[16], Tn[16];
        if (mode == 1) update_p2plane(acc, U); else update_p2p(acc, U);
        mat4_mul(U, st->T, Tn);
        for (int c = 0; c < 16; ++c) st->T[c] = Tn[c];
    }
    if (done) st->done = 1;
    if (result) {
        for (int c = 0; c < 16; ++c) result[c] = st->T[c];
        result[16] = fit; result[17] = rmse; result[18] = (double)k; result[19] = cnt;
    }
}

// The same step done by ONE WAVE (all 64 lanes call it; acc, st and fs in LDS or global memory visible to the wave).  The
// serial version above keeps its 6x6 factors in scratch memory (dynamically indexed arrays): ~5 us of dependent memory
// round trips in front of every block's sweep when the update runs in the iteration kernel's prologue.  Here the 6x6 system
// is solved by Gauss-Jordan elimination on the augmented matrix [J^T J | -J^T r] held in LDS, lane (i, c) owning entry (i, c):
// six rank-1 steps (no pivoting: the matrix is symmetric positive definite, as for Open3D's ldlt), then the three sine /
// cosine pairs on three lanes and the 4x4 product U T on sixteen.  Point-to-point keeps the serial Jacobi/Kabsch on lane 0.
// Mathematically the same update; rounding differs from the LDL^T order at the 1e-16 level (T is tolerance-checked).
// Blocks that provably cannot find a partner are not swept again (icp_iter_body).  A block whose four waves all found NO target
// group within reach records key = motion + g, g = the smallest distance from a wave's box to any group box (> reach).  Every
// later update moves a source point by at most  |U s - s| <= ||R_u - I||_F |s| + |t_u|,  |s| <= max over the corners c of the source's
// box of |T c|  (|.| is convex: holds for any affine T), which
// the update step adds to `motion`; `reach` bounds the square root of any row's search bound under the current T (the clamp of
// max_correspondence_distance plus the rounding margins of nn_local's metric).  While motion + reach < key no row of the block
// can have a target point within its bound, so the sweep would report "no partner" for all of them -- exactly what the rows
// already hold.  The bounds carry relative margins of 1e-9 .. 1e-6: they only delay skipping, never allow a wrong one.
struct LightSkip {
    const double *sbbox;       // bounding box of the ORIGINAL source points (lo xyz, hi xyz); nullptr: no bookkeeping
    double max_d2, t2max;
};
struct FinishScratch {
    double M[6][8];
    double trig[6];
    double U[16];
    double Tn[16];
    int flag;
};
__device__ __forceinline__ void icp_finish_wave(const double *acc, int64_t n, int mode, int k, int max_iter, double rel_fit, double rel_rmse,
                                                IcpState *st, double *__restrict__ result, FinishScratch &fs, int lane, const LightSkip ls = LightSkip{ nullptr, 0.0, 0.0 },
                                                unsigned long long *dbg = nullptr)
{
    auto tick = [&](int slot) { if (dbg && lane == 0) dbg[slot] = wall_clock64(); };
    tick(0);
    // largest |T p| over the source: |.| is convex, so it is attained at a corner of the source's bounding box (any affine T)
    // (every lane of the wave calls it: lane c & 7 takes corner c, the maximum -- exact, order-free -- is folded over the lanes)
    auto reach_of_source = [&](const double *T) {
        const int c = lane & 7;
        const double x = ls.sbbox[(c & 1) ? 3 : 0], y = ls.sbbox[(c & 2) ? 4 : 1], z = ls.sbbox[(c & 4) ? 5 : 2];
        double n2 = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) { const double v = T[4 * r] * x + T[4 * r + 1] * y + T[4 * r + 2] * z + T[4 * r + 3]; n2 += v * v; }
        double m2 = fmax(0.0, n2);
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) m2 = fmax(m2, __shfl_xor(m2, o, 64));
        return sqrt(m2) * (1.0 + 1e-9);
    };
    if (lane == 0) {
        const double cnt = acc[0];
        const double fit = (n > 0 && cnt > 0) ? cnt / (double)n : 0.0;
        const double rmse = cnt > 0 ? sqrt(acc[1] / cnt) : 0.0;
        bool done = false;
        if (k >= 1 && fabs(st->fitness - fit) < rel_fit && fabs(st->rmse - rmse) < rel_rmse) done = true;
        st->fitness = fit; st->rmse = rmse; st->count = cnt; st->iter = k;
        if (k >= max_iter) done = true;
        if (done) st->done = 1;
        fs.flag = done ? 1 : (cnt < 1.0 ? 2 : 0);              // 2: no correspondence -> identity update
    }
    wave_lds_fence();
    tick(1);
    const int flag = fs.flag;
    if (flag == 0) {
        if (lane < 16) fs.U[lane] = (lane % 5 == 0) ? 1.0 : 0.0;
        if (mode == 1) {
            const int i = lane / 7, c = lane % 7;
            if (lane < 42) {
                double v;
                if (c < 6) {
                    const int a = i < c ? i : c, b = i < c ? c : i;           // upper-triangle slot of (a, b): 17 + a(13 - a)/2 + (b - a)
                    v = acc[17 + (a * (13 - a)) / 2 + (b - a)];
                } else {
                    v = -acc[38 + i];
                }
                fs.M[i][c] = v;
            }
            wave_lds_fence();
            tick(2);
            bool ok = true;
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
                const double d = fs.M[jj][jj];
                ok = ok && (fabs(d) > 1e-300);
                double v = 0.0;
                const bool mine = lane < 42 && i != jj;
                if (mine) {
                    const double f = fs.M[i][jj] / d;
                    v = c == jj ? 0.0 : fs.M[i][c] - f * fs.M[jj][c];
                }
                wave_lds_fence();
                if (mine && ok) fs.M[i][c] = v;
                wave_lds_fence();
            }
            tick(3);
            if (ok) {
                if (lane < 3) {
                    const double a = fs.M[lane][6] / fs.M[lane][lane];
                    fs.trig[2 * lane] = cos(a);
                    fs.trig[2 * lane + 1] = sin(a);
                } else if (lane < 6) {
                    fs.U[4 * (lane - 3) + 3] = fs.M[lane][6] / fs.M[lane][lane];
                }
                wave_lds_fence();
                if (lane == 0) {
                    const double ca = fs.trig[0], sa = fs.trig[1], cb = fs.trig[2], sb = fs.trig[3], cg = fs.trig[4], sg = fs.trig[5];
                    // Rz(g) Ry(b) Rx(a)
                    fs.U[0] = cg * cb; fs.U[1] = cg * sb * sa - sg * ca; fs.U[2] = cg * sb * ca + sg * sa;
                    fs.U[4] = sg * cb; fs.U[5] = sg * sb * sa + cg * ca; fs.U[6] = sg * sb * ca - cg * sa;
                    fs.U[8] = -sb;     fs.U[9] = cb * sa;                fs.U[10] = cb * ca;
                }
            }
        } else if (lane == 0) {
            double U[16];
            update_p2p(acc, U);
#pragma unroll
            for (int e = 0; e < 16; ++e) fs.U[e] = U[e];
        }
        wave_lds_fence();
        tick(4);
        // the lever arm of this update = the reach of the source under the transform it is applied to: what the previous update left in
        // `smax` (the same function of the same T, bit for bit); the first update of a registration computes it
        double lever = 0.0;
        if (ls.sbbox) lever = st->smax >= 0.0 ? st->smax : reach_of_source(st->T);      // (wave-uniform branch: smax is one LDS / memory word)
        if (ls.sbbox && lane == 0) {                       // bound on the displacement this update gives any source point
            double rot = 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) { const double d = fs.U[4 * r + c] - (r == c ? 1.0 : 0.0); rot += d * d; }
            const double tu = sqrt(fs.U[3] * fs.U[3] + fs.U[7] * fs.U[7] + fs.U[11] * fs.U[11]);
            const double moved = (sqrt(rot) * lever + tu) * (1.0 + 1e-9) + 1e-9;
            st->motion += moved;
            st->last_motion = moved;
        }
        tick(5);
        if (lane < 16) {
            const int r = lane >> 2, c = lane & 3;
            double v = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e) v += fs.U[4 * r + e] * st->T[4 * e + c];
            fs.Tn[lane] = v;
        }
        wave_lds_fence();
        if (lane < 16) st->T[lane] = fs.Tn[lane];
    }
    wave_lds_fence();
    tick(6);
    double smax = 0.0;
    if (ls.sbbox) smax = reach_of_source(st->T);
    if (ls.sbbox && lane == 0) {                           // reach of a row's search under the transform the next sweep uses
        st->smax = smax;
        // upper bound of nn_local's row bound rb = (clamp - 1)(1 + 2^-30) + eps with clamp, eps as in icp_iter_body / sweep_wave
        const double r2 = ls.max_d2 * (1.0 + 3.7252902984619140625e-9) + 3.7252902984619140625e-9 + 1.4551915228366851806640625e-11 * (smax * smax + ls.t2max + 2.0);
        st->reach = sqrt(r2) * (1.0 + 1e-9);
    }
    tick(7);
    if (result) {
        if (lane < 16) result[lane] = st->T[lane];
        if (lane == 0) { result[16] = st->fitness; result[17] = st->rmse; result[18] = (double)k; result[19] = st->count; }
    }
}

// The same behind a CALL (icp_chain_kernel): inlined into that kernel's loop, the literal constants of cos / sin / sqrt are hoisted
// out of the loop into registers the kernel does not have, and spilled around every sweep
__device__ __attribute__((noinline)) void icp_finish_wave_call(const double *acc, int64_t n, int mode, int k, int max_iter, double rel_fit, double rel_rmse,
                                                               IcpState *st, double *result, FinishScratch *fs, int lane, const double *sbbox, double max_d2,
                                                               double t2max, unsigned long long *dbg)
{
    icp_finish_wave(acc, n, mode, k, max_iter, rel_fit, rel_rmse, st, result, *fs, lane, LightSkip{ sbbox, max_d2, t2max }, dbg);
}

// 1024 threads: thread (slot q = t & 63, slice = t >> 6) sums its slice of the per-block partials with eight
// interleaved accumulators (lanes of a wave read consecutive slots of one partial row: coalesced; a row-per-thread
// variant was 3x slower, the single CU's address path saturates), the sixteen slices are then added in order (a
// fixed summation tree: bitwise reproducible), thread 0 does the algebra.  k = index of the search just finished.
constexpr int kSolveThreads = 1024;
__global__ __launch_bounds__(kSolveThreads) void icp_solve_kernel(const double *__restrict__ part_acc, int nblocks, int64_t n, int mode, int k,
                                                                  int max_iter, double rel_fit, double rel_rmse, IcpState *st,
                                                                  double *__restrict__ result)
{
    if (st->done) return;
    __shared__ double part[kSolveThreads / 64][64];
    __shared__ double acc[kAcc];
    const int nacc = mode == 1 ? kAcc : 17;
    const int q = threadIdx.x & 63, slice = threadIdx.x >> 6;
    {
        double u[8] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
        if (q < nacc) {
            const int per = (nblocks + kSolveThreads / 64 - 1) / (kSolveThreads / 64);
            const int b0 = slice * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
            int b = b0;
            for (; b + 7 < b1; b += 8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) u[e] += part_acc[(int64_t)(b + e) * kAcc + q];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (b + e < b1) u[e] += part_acc[(int64_t)(b + e) * kAcc + q];
        }
        part[slice][q] = ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
    }
    __syncthreads();
    if (threadIdx.x < kAcc) {
        double v = 0.0;
        if ((int)threadIdx.x < nacc)
            for (int sl = 0; sl < kSolveThreads / 64; ++sl) v += part[sl][threadIdx.x];
        acc[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x) return;
    icp_finish(acc, n, mode, k, max_iter, rel_fit, rel_rmse, st, result);
}

struct Mat16 {
    double m[16];
};
__global__ void icp_init_kernel(IcpState *st, Mat16 T0)
{
    if (threadIdx.x || blockIdx.x) return;
    for (int slot = 0; slot < 2; ++slot) {               // both slots (see IcpFuse)
        for (int q = 0; q < 16; ++q) st[slot].T[q] = T0.m[q];
        st[slot].fitness = 0.0; st[slot].rmse = 0.0; st[slot].count = 0.0; st[slot].iter = 0; st[slot].done = 0;
        st[slot].motion = 0.0; st[slot].reach = INFINITY; st[slot].last_motion = INFINITY; st[slot].smax = -1.0;
    }
}
static Mat16 mat16_from(const double *h)
{
    Mat16 m;
    for (int q = 0; q < 16; ++q) m.m[q] = h[q];
    return m;
}

// ---- explicit-pair Kabsch (compute_transformation with a correspondence list) ------------------------------
__global__ __launch_bounds__(256) void pairs_acc_kernel(const float *__restrict__ src, const float *__restrict__ tgt,
                                                        const int32_t *__restrict__ corr, int64_t nc, double *__restrict__ part_acc)
{
    __shared__ double sh[4];
    double acc[17];
#pragma unroll
    for (int q = 0; q < 17; ++q) acc[q] = 0.0;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += (int64_t)gridDim.x * blockDim.x) {
        const float *sp = src + 3 * (int64_t)corr[2 * c], *tp = tgt + 3 * (int64_t)corr[2 * c + 1];
        double s[3] = { (double)sp[0], (double)sp[1], (double)sp[2] }, t[3] = { (double)tp[0], (double)tp[1], (double)tp[2] };
        acc[0] += 1.0;
        for (int k = 0; k < 3; ++k) { acc[2 + k] += s[k]; acc[5 + k] += t[k]; }
        for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q) acc[8 + 3 * p + q] += t[p] * s[q];
    }
    for (int q = 0; q < 17; ++q) {
        double v = block_sum(acc[q], sh);
        if (threadIdx.x == 0) part_acc[(int64_t)blockIdx.x * kAcc + q] = v;
    }
}
__global__ __launch_bounds__(64) void pairs_solve_kernel(const double *__restrict__ part_acc, int nblocks, double *__restrict__ T)
{
    __shared__ double acc[17];
    if (threadIdx.x < 17) {
        double s = 0.0;
        for (int b = 0; b < nblocks; ++b) s += part_acc[(int64_t)b * kAcc + threadIdx.x];
        acc[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x) return;
    double U[16];
    update_p2p(acc, U);
    for (int q = 0; q < 16; ++q) T[q] = U[q];
}


// ---- exact accumulation of the update sums (culled engine) --------------------------------------------------------
// Every block adds its 44 fp64 partial sums to 128-bit fixed-point accumulators (kpx_fixed.h): the totals do not
// depend on the order of the blocks, so the registration stays bitwise reproducible with ONE pair of words per sum
// instead of one partial row per block -- the solve kernel reads 8 x 44 pairs instead of N/64 x 44 doubles.
// kAccCopies copies (block & 7) keep the same-address atomic traffic low.
constexpr int kAccCopies = 8;
__device__ __forceinline__ double fixed_total(const unsigned long long *acc, int slot)
{
    unsigned long long lo = 0ull, hi = 0ull;
#pragma unroll
    for (int c = 0; c < kAccCopies; ++c) {
        const unsigned long long l = acc[((int64_t)c * kAcc + slot) * 2], h = acc[((int64_t)c * kAcc + slot) * 2 + 1];
        lo += l;
        hi += h + (lo < l ? 1ull : 0ull);
    }
    return fixed_value(lo, hi);
}
// the same through device-coherent loads (for a reader inside the launch that did the adds: the XCDs' L2s are not coherent)
__device__ __forceinline__ double fixed_total_coherent(unsigned long long *acc, int slot)
{
    unsigned long long l[kAccCopies], h[kAccCopies];
#pragma unroll
    for (int c = 0; c < kAccCopies; ++c) {
        l[c] = __hip_atomic_load(acc + ((int64_t)c * kAcc + slot) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h[c] = __hip_atomic_load(acc + ((int64_t)c * kAcc + slot) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned long long lo = 0ull, hi = 0ull;
#pragma unroll
    for (int c = 0; c < kAccCopies; ++c) {
        lo += l[c];
        hi += h[c] + (lo < l[c] ? 1ull : 0ull);
    }
    return fixed_value(lo, hi);
}
// sums -> update step; clears the accumulators for the next iteration (single block: no race)
// progress: optional word in pinned host memory, tag | done << 32 | iterations finished -- the batch driver reads it to keep
// a window of iterations queued per problem without copies or events (system-scope release store by one thread)
__global__ __launch_bounds__(256) void icp_solve_fixed_kernel(unsigned long long *acc, int64_t n, int mode, int k, int max_iter, double rel_fit,
                                                              double rel_rmse, IcpState *st, double *__restrict__ result,
                                                              unsigned long long *progress, unsigned long long tag)
{
    if (st->done) return;
    __shared__ double sums[kAcc];
    const int nacc = mode == 1 ? kAcc : 17;
    if (threadIdx.x < kAcc) sums[threadIdx.x] = (int)threadIdx.x < nacc ? fixed_total(acc, threadIdx.x) : 0.0;
    __syncthreads();
    for (int e = threadIdx.x; e < kAccCopies * kAcc * kFixedWords; e += 256) acc[e] = 0ull;
    __shared__ FinishScratch fs;
    if (threadIdx.x >= 64) return;
    icp_finish_wave(sums, n, mode, k, max_iter, rel_fit, rel_rmse, st, result, fs, (int)threadIdx.x);
    if (threadIdx.x) return;
    if (progress)
        __hip_atomic_store(progress, tag | ((unsigned long long)(st->done ? 1 : 0) << 32) | (unsigned long long)(unsigned)(k + 1), __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace kpx
#include "kpx_nnlocal.h"
namespace kpx {

// ---- one ICP iteration in one kernel (culled engine) -----------------------------------------------------------
// Block = 4 waves x 16 sorted rows.  Prologue: lanes 0..15 of a wave transform their row, seed it and bound it
// with last iteration's partner (clamped to the correspondence distance); the wave sweeps (sweep_wave); lanes 0..15
// then form the chosen pair's direct distance (AC3) and the row's contribution to the update sums, which are added
// in a fixed order per block and added to the exact fixed-point accumulators; icp_solve_fixed_kernel performs the
// update step (kpx_icp).  Running the update redundantly in the prologue of the next launch (IcpFuse below) puts the serial
// 6x6 / eigen algebra (~5-8 us) in front of every block's sweep against 5.7 us + 1.9 us for the solve kernel and its
// boundary: for one large registration (100k x 100k: five rounds of blocks per launch) it was 20 % slower, so kpx_icp
// keeps two kernels per iteration; kpx_icp_batch, whose small problems fit one round of blocks and whose chains are
// bound by the host's launch rate, uses the one-launch form.
// ("Last block finishes the job" inside this launch was measured twice and lost both times: with plain stores +
// __threadfence() the release writes back / invalidates the XCD's L2 once per block (10x slower); with write-through
// device-scope stores, a drained vmcnt and a relaxed ticket it still adds ~13 us at 485 blocks -- the same-address
// ticket atomics serialise at ~12 ns each and the last block reads 170 KB through sc1 loads -- against ~11 us for
// the boundary + the 1024-thread solve kernel.)
#ifndef KPX_ICP_WAVES
#define KPX_ICP_WAVES 4
#endif
constexpr int kIWaves = KPX_ICP_WAVES;       // waves (16-row tiles) per block: a block lives as long as its slowest wave
constexpr int kIThreads = kIWaves * 64;
constexpr int kIRows = kIWaves * kLRows;
// One launch per iteration (used by kpx_icp_batch, whose chains are bound by the host's launch rate once several
// registrations and two frames run side by side): launch k first performs the update of iteration k-1 -- every block
// folds the accumulator set of the previous launch and runs the (deterministic) algebra itself, block 0 publishes the
// state, the result and the progress word -- then sweeps with the new transform.  Three accumulator sets in a ring
// (launch k reads set k-1, adds to set k, block 0 clears set k+1) and two state slots (launch k reads slot k-1, writes
// slot k) keep the launches free of races.  pair == nullptr: two-kernel mode, icp_solve_fixed_kernel does the update.
constexpr int kCertHist = 64;                 // iterations whose transforms are kept for the certificates (6 bits of the word)
// The whole chain in ONE launch (icp_chain_kernel): the blocks of a registration stay resident and iterate; the hand-off between
// iterations is a RECORD per iteration -- the registration's state as the update of iteration k - 1 left it -- whose 23 words the
// winner (the block that drew the last ticket) writes with device-coherent stores and wave 0 of every block polls with
// device-coherent loads, each lane ITS word, until none of them is the "empty" pattern any more (a NaN payload no computation
// produces): every word validates itself, so there is no flag, no release and no second round trip.
constexpr int kChainRec = 32;                 // doubles per record (23 used)
constexpr int kChainRecords = 64;             // records 0 .. max_iteration + 1: the chain form serves max_iteration <= 62
constexpr int kChainWords = 24;                // = sizeof(IcpState) / 8
constexpr unsigned long long kChainEmpty = 0xFFF8C0DEC0DEC0DEull;
struct CertPolicy {
    float calm, factor, smin, smax;      // KPX_CERT_CALM / _FACTOR / _SKIN_MIN / _SKIN_MAX (fractions of the correspondence distance)
};
struct IcpFuse {
    IcpState *pair;
    unsigned long long *ring;
    int max_iter;
    double rel_fit, rel_rmse;
    double *result;
    unsigned long long *progress, tag;
    unsigned long long *ticket;        // non-null (with pair == nullptr): the LAST block of the launch to deliver its sums performs the update
    double *light_key;                 // with ticket: per block, LightSkip key (0 = sweep); nullptr: every block sweeps
    const double *sbbox;               // with light_key: the source's bounding box
    uint32_t *cert;                    // with light_key: per sorted row, the certificate: L as a float rounded down to 17 mantissa bits | the iteration
                                       // of the search in the low 6 bits (0 = none); nullptr: every row is searched
    double *thist;                     // with cert: the transforms of iterations 0 .. 63, 12 doubles each (the winner writes entry k + 1)
    int cert_check;                    // self-check mode: certified rows are searched anyway and compared (g_cert_check)
    CertPolicy pol;
    double *chain_rec;                 // icp_iter_body<true> only: the registration's records (kChainRecords x kChainRec doubles)
    unsigned long long *stamp;         // icp_iter_body<true>, KPX_ICP_CHAIN_STAMPS=1: this iteration's row of g_chain_stamp
};
constexpr int kAccSet = kAccCopies * kAcc * kFixedWords;
// Phase clock of the iteration kernel (while the profiler is armed): thread 0 of every block stores 100 MHz wall-clock stamps in
// its own row of g_icp_stamp -- [0] block start, [1] after the update prologue, [2] after row preparation, [3] after the culled
// sweep, [4] after the pair epilogue, [5] block end.  Plain stores to private slots: the clock does not disturb what it times.
// The rows of the LAST launch are read by kpx_prof_icp_phases.
constexpr int kStampBlocks = 4096;
__device__ unsigned long long g_icp_stamp[kStampBlocks][8];
// per WAVE of the last sweep launch: [0] sweep start, [1] sweep end (100 MHz), [2] the packed counters sweep_wave returns, [3] rows
// of the wave that ended with a partner
__device__ unsigned long long g_icp_wave[kStampBlocks * 4][4];
// Certificate self-check (KPX_ICP_CERT_CHECK=1): certified rows are searched all the same and the search's winner is compared with the
// partner the certificate kept.  [0] rows certified, [1] rows searched, [2] certified rows whose search disagreed, [3..7] the first
// disagreement: iteration, sorted row, kept partner, found partner, key as float bits.  Read (and cleared) by kpx_prof_icp_cert.
__device__ unsigned long long g_cert_check[8];
// Clock of the one-launch chain (KPX_ICP_CHAIN_STAMPS=1, first registration of the launch; 100 MHz stamps, one row per iteration):
// block 0: [0] record seen, [1] rows prepared, [2] sweep over, [3] sums added, [4] ticket drawn; over all blocks: [5] latest / [10]
// earliest "record seen", [11] latest "sums added", [6] latest ticket; the winner: [7] totals read, [8] update done, [9] record published.
__device__ unsigned long long g_chain_stamp[64][48];    // [32 ..]: the winner's update step from inside (icp_finish_wave, tick)    // [16 ..]: block 0 wave 0's sweep (sweep_wave, dbg_tick)
__device__ __forceinline__ void chain_tick(unsigned long long *row, int slot, bool on)
{
    if (row && on) row[slot] = wall_clock64();
}
__device__ __forceinline__ void phase_tick(unsigned long long *__restrict__ armed, int slot, unsigned bid)
{
    if (!armed || threadIdx.x || bid >= kStampBlocks) return;
    __builtin_nontemporal_store(wall_clock64(), &g_icp_stamp[bid][slot]);
}
// Certificates: rows whose partner provably cannot change are not searched again.
// A sweep knows more than the winner: every column it multiplied gives D, every box it culled was farther than the row's culling
// bound.  L = sqrt(min(final culling bound, smallest D - 1 among the multiplied columns OTHER than the winner)) is therefore a lower
// bound of the distance from the row to every other target point.  The row keeps (p_c, L): its position at that search and L.  In a
// later iteration it stands at p, every other target is still >= L - |p - p_c| away (triangle inequality, the row's OWN displacement:
// no global bound), and if the partner's own (exactly evaluated) distance d1 is smaller than that -- d1 + |p - p_c| < L, with margins
// for the float32 copy of p_c and the roundings -- the partner is the STRICT nearest neighbour: an exact search would return it, ties
// and all, so the row keeps it without one.  Rows without a partner use the reach of any row's search in place of d1.  A certificate
// is worth something only if the search looked beyond its partner: once the registration is calm (the last update moved no point by
// more than `calm` x the correspondence distance; IcpState::last_motion, LightSkip's bookkeeping) uncertified rows are searched with a
// skin around their partner, `factor` x that motion (between `smin` and `smax` x the correspondence distance): a few more tiles
// multiplied once, no search at all in the iterations that follow.  Waves whose 16 rows are all certified skip the sweep, the others
// cull with the box and bounds of their uncertified rows only.  Partners, sums and transforms are those of the full search, bit for
// bit.  The policy is CertPolicy, above IcpFuse.  (KPX_ICP_CERT=0 switches the certificates off: test_icp_update_placements_and_light_skip_are_bit_identical; KPX_ICP_CERT_CHECK=1
// searches the certified rows all the same and counts disagreements: test_icp_certificates_never_contradict_the_search).
// bid / nblocks: this block's index among the blocks of ITS registration (one launch may carry several, see icp_iter_batch_kernel)
template <bool PERSIST = false>
__device__ __forceinline__ void icp_iter_body(const unsigned bid, const unsigned nblocks, const float *__restrict__ src, int64_t n, const float *__restrict__ tgt,
                                                       const float *__restrict__ tn, const double *__restrict__ Bs,
                                                       const int32_t *__restrict__ orig, const float *__restrict__ tile_box,
                                                       const float *__restrict__ group_box, int32_t n_groups,
                                                       const double *__restrict__ tbbox, const int32_t *__restrict__ row_of,
                                                       const float *__restrict__ src_sorted, int32_t *__restrict__ idx_sorted,
                                                       float *__restrict__ ptgt_sorted,
                                                       int32_t *__restrict__ idx_cur, double *__restrict__ d2_cur, double max_d2, int mode,
                                                       int k, const IcpState *__restrict__ st, unsigned long long *acc,
                                                       unsigned long long *__restrict__ tile_visits, IcpFuse fuse)
{
    // (chain form: the thread number behind an opaque move, taken anew in every iteration -- otherwise everything derived from it,
    // LDS addresses first of all, is hoisted out of the chain's loop and held in registers this kernel does not have)
    const unsigned tix = PERSIST ? (unsigned)opaque_i((int)threadIdx.x) : threadIdx.x;
    __shared__ IcpState s_state;
    __shared__ double s_sums[kAcc];
    // PERSIST (icp_chain_kernel: this body runs once per iteration inside ONE launch, `st` is the block's LDS copy of the iteration's
    // record): what a row carries from one iteration to the next -- its coordinates, its partner (index, coordinates, normal), its
    // certificate, the block's LightSkip key -- stays in LDS; nothing but constants is read from memory after iteration 0.
    __shared__ float rowk[kIWaves][16][8];           // what a row carries across the sweep (its previous partner: coordinates, normal, index),
                                                     // parked here: a value in 16 lanes costs a whole register through the multiply loop
    __shared__ float rowsrc[kIWaves][16][3];
    __shared__ uint32_t rowc[kIWaves][16];
    __shared__ int32_t rowi[kIWaves][16][2];         // partner (bound / result), original row
    __shared__ double s_key;
    const int wave = PERSIST ? __builtin_amdgcn_readfirstlane((int)(tix >> 6)) : (int)(tix >> 6), lane = tix & 63, q = lane >> 4, j = lane & 15;
    const int64_t row_base = ((int64_t)bid * kIWaves + wave) * kLRows;
    const int64_t last = n - 1;
    const unsigned long long t_block_start = (tile_visits && tix == 0) ? wall_clock64() : 0ull;
    const double t2max = target_t2max(tbbox);
    // An iteration is a chain of dependent memory round trips, so everything that does not depend on this iteration's
    // transform is requested FIRST -- the wave's rows, their previous partners (index AND coordinates, kept in sorted-row
    // order by the previous launch: no gather through the index), the first 128 group boxes -- and arrives while the update
    // algebra of the previous iteration runs below.
    float my_src[3] = { 0.0f, 0.0f, 0.0f }, my_pt[3] = { 0.0f, 0.0f, 0.0f }, my_nrm[3] = { 0.0f, 0.0f, 0.0f };
    int32_t my_row = 0, my_prev = -1;
    uint32_t my_cert = 0u;
    const bool certs = fuse.ticket && fuse.light_key && fuse.cert;
    if (PERSIST && k > 0) {
        if (lane < 16) {
            my_row = rowi[wave][lane][1];
#pragma unroll
            for (int a = 0; a < 3; ++a) { my_src[a] = rowsrc[wave][lane][a]; my_pt[a] = rowk[wave][lane][a]; my_nrm[a] = rowk[wave][lane][3 + a]; }
            my_prev = __float_as_int(rowk[wave][lane][6]);
            if (certs) my_cert = rowc[wave][lane];
        }
    } else if (lane < 16) {
        const int64_t r = row_base + lane < last ? row_base + lane : last;
        if (idx_cur || d2_cur) my_row = row_of[r];           // only the caller-order outputs need the original row number
#pragma unroll
        for (int a = 0; a < 3; ++a) my_src[a] = src_sorted[3 * r + a];
        if (PERSIST) {
#pragma unroll
            for (int a = 0; a < 3; ++a) rowsrc[wave][lane][a] = my_src[a];
            rowc[wave][lane] = 0u;
        }
        if (k > 0) {
            my_prev = idx_sorted[r];
#pragma unroll
            for (int a = 0; a < 3; ++a) my_pt[a] = ptgt_sorted[3 * r + a];
            if (certs) my_cert = fuse.cert[r];
        }
    }
    // (certificates) the transforms of the iterations so far: a certified row's position at its search is recomputed from them, exactly
    __shared__ double s_thist[kCertHist][12];
    if (PERSIST) {                                        // (the earlier entries are still there; a barrier follows before the rows use them)
        if (certs && k < kCertHist && tix < 12) s_thist[k][tix] = st->T[tix];
    } else if (certs && k > 0)
        for (int e = tix; e < 12 * (k < kCertHist ? k : kCertHist); e += kIThreads) (&s_thist[0][0])[e] = fuse.thist[e];
    GroupPre gpre;
    group_pre_load(gpre, group_box, n_groups, lane);
    if (!PERSIST && mode == 1 && k > 0 && lane < 16) {
        // the previous partner's normal, requested through the index as soon as that has arrived (behind everything that does not
        // depend on anything): it is not needed before the pair epilogue, where an unchanged partner -- the rule in the late
        // iterations -- then costs no round trip at all
        const float *np_ = tn + 3 * (int64_t)(my_prev > 0 ? my_prev : 0);
#pragma unroll
        for (int a = 0; a < 3; ++a) my_nrm[a] = np_[a];
    }
    const double *Tk = st->T;
    // (ticket mode) the registration's state, copied into LDS while everything else loads: the block that draws the last ticket runs the
    // update step on this copy -- fitness / rmse / T / motion were four dependent global round trips inside a step that every launch waits
    // for -- and writes the new state back in one burst
    static_assert(sizeof(IcpState) % sizeof(double) == 0, "IcpState is copied as doubles");
    if (KPX_ICP_STATE_LDS && fuse.ticket && tix < sizeof(IcpState) / sizeof(double))
        reinterpret_cast<double *>(&s_state)[tix] = reinterpret_cast<const double *>(st)[tix];
    if (fuse.pair) {
        const IcpState *in = fuse.pair + ((k + 1) & 1);
        IcpState *out = fuse.pair + (k & 1);
        // state and accumulators are read in ONE round trip (the sums of a converged chain are simply not used)
        const unsigned long long *prev = fuse.ring + (int64_t)((k + 2) % 3) * kAccSet;
        if (k > 0 && tix < kAcc) s_sums[tix] = (int)tix < (mode == 1 ? kAcc : 17) ? fixed_total(prev, tix) : 0.0;
        if (tix == 0) s_state = *in;
        __syncthreads();
        if (s_state.done) {                               // converged earlier: hand the state on, nothing else to do
            if (bid == 0 && tix == 0) *out = s_state;
            return;
        }
        __shared__ FinishScratch s_fs;
        if (k > 0 && wave == 0)
            icp_finish_wave(s_sums, n, mode, k - 1, fuse.max_iter, fuse.rel_fit, fuse.rel_rmse, &s_state, bid == 0 ? fuse.result : (double *)nullptr,
                            s_fs, lane);
        __syncthreads();
        if (bid == 0) {
            unsigned long long *next = fuse.ring + (int64_t)((k + 1) % 3) * kAccSet;
            for (int e = tix; e < kAccSet; e += kIThreads) next[e] = 0ull;
            if (tix == 0) {
                *out = s_state;
                if (k > 0 && fuse.progress)
                    __hip_atomic_store(fuse.progress, fuse.tag | ((unsigned long long)(s_state.done ? 1 : 0) << 32) | (unsigned long long)(unsigned)k,
                                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (s_state.done) return;
        Tk = s_state.T;
        acc = fuse.ring + (int64_t)(k % 3) * kAccSet;
    } else if (st->done) return;
    // LightSkip: nothing of this block can have come within reach since it was last swept -> straight to the ticket
    bool skip = false;
    if (fuse.ticket && fuse.light_key) {
        const double key = PERSIST ? (k > 0 ? s_key : 0.0) : fuse.light_key[bid];
        skip = key > 0.0 && (st->motion + st->reach) * (1.0 + 1e-6) + 1e-6 < key;
    }
    __shared__ double s_light[kIWaves];
    const int nacc = mode == 1 ? kAcc : 17;
    if (tile_visits && tix == 0 && bid < kStampBlocks) { g_icp_stamp[bid][7] = skip ? 1ull : 0ull; g_icp_stamp[bid][6] = (unsigned long long)nblocks; }
    do {
    if (skip) break;
    __shared__ int32_t lists[kIWaves][kLScratch];
    __shared__ double rowd[kIWaves][16][kRowStride]; // s_x, s_y, s_z, (the sweep's row bound), K, bound / result value
    __shared__ float rowf[kIWaves][16][kRowFStride]; // float32 mirror of the rows for the sweep's culling tests
    __shared__ double sh[kAcc][kIRows + 1];
    if (tile_visits && tix == 0 && bid < kStampBlocks) {     // only launches that sweep stamp (not the converged / closing ones)
        g_icp_stamp[bid][0] = t_block_start;
        g_icp_stamp[bid][6] = (unsigned long long)nblocks;
    }
    phase_tick(tile_visits, 1, bid);
    // s_thist was staged by all threads of the block and is read by the rows of every wave (until this barrier was added the
    // per-launch form relied on the waves of a block running in step: a row could read an entry before another wave had written it)
    if (certs && (k > 0 || PERSIST)) __syncthreads();
    if (PERSIST && fuse.stamp && tix == 0) {
        const unsigned long long t = wall_clock64();
        if (bid == 0) fuse.stamp[0] = t;
        atomicMax(&fuse.stamp[5], t);
        atomicMin(&fuse.stamp[10], t);
    }

    // Row certificates (kpx_icp.hip, "Certificates" above icp_iter_body): how calm the registration is decides the skin
    const double c_reach = certs ? st->reach : 0.0;
    double c_skin = 0.0;
    if (certs && k > 0) {
        const double lm = st->last_motion, md = sqrt(max_d2);
        if (lm <= (double)fuse.pol.calm * md) c_skin = fmin(fmax((double)fuse.pol.factor * lm, (double)fuse.pol.smin * md), (double)fuse.pol.smax * md);
    }
    bool my_active = true, my_certd = false;
    if (lane < 16) {
        const int64_t i = my_row;
        double s[3];
        xform_row(Tk, my_src, s);
        const double seed = row_seed(s);
        double bv = INFINITY;
        int32_t bj = INT_MAX;
        if (k > 0) {
            const int32_t p = my_prev;
            if (p >= 0) {
                const double tx = my_pt[0], ty = my_pt[1], tz = my_pt[2];
                const double t2 = fma(tx, tx, fma(ty, ty, tz * tz));
                double d = fma(s[0], -2.0 * tx, seed);
                d = fma(s[1], -2.0 * ty, d);
                d = fma(s[2], -2.0 * tz, d);
                bv = fma(1.0, t2, d);
                bj = p;
            }
        }
        const double clamp = (max_d2 + 1.0) * (1.0 + 9.31322574615478515625e-10) + ldexp(seed + t2max + 1.0, -38);
        if (!(bv <= clamp)) { bv = clamp; bj = INT_MAX; }
        double rb0 = bv - 1.0;
        if (certs) {
            // d1: an upper bound of the distance from the row to its partner (a row without one: the reach of any row's search)
            double d1 = c_reach;
            if (bj != INT_MAX) {
                const double dx = s[0] - (double)my_pt[0], dy = s[1] - (double)my_pt[1], dz = s[2] - (double)my_pt[2];
                d1 = sqrt(fma(dz, dz, fma(dy, dy, dx * dx))) * (1.0 + 1e-12);
            }
            // certified: every other target point was >= L away from p_c, the row has moved by |p - p_c| since (p_c is kept as float32:
            // 2^-24 relative per coordinate, covered by the 1e-6 relative margin on the coordinates' scale), and its partner (or the
            // reach of a row without one) is nearer than what is left of L.  A row that HAD a partner and lost it to the clamp is
            // searched (the certificate says nothing about that partner).
            const bool keeps = (my_prev >= 0) == (bj != INT_MAX);
            const int kc = (int)(my_cert & 63u);
            const float Lc = __uint_as_float(my_cert & ~63u);
            double pc[3] = { 0.0, 0.0, 0.0 };
            if (Lc > 0.0f) {                                     // the row's position at that search: AC1 with that iteration's transform
                const double *Th = s_thist[kc];
                const double x = my_src[0], y = my_src[1], z = my_src[2];
#pragma unroll
                for (int a = 0; a < 3; ++a) pc[a] = fma(Th[4 * a], x, fma(Th[4 * a + 1], y, fma(Th[4 * a + 2], z, Th[4 * a + 3])));
            }
            const double ex = s[0] - pc[0], ey = s[1] - pc[1], ez = s[2] - pc[2];
            const double moved = sqrt(fma(ez, ez, fma(ey, ey, ex * ex))) * (1.0 + 1e-12);
            const bool certd = Lc > 0.0f && keeps && (d1 + moved) * (1.0 + 1e-6) + 1e-6 < (double)Lc;
            my_active = (!certd || fuse.cert_check) && row_base + lane <= last;
            my_certd = certd && row_base + lane <= last;
            if (certd && !fuse.cert_check) rb0 = -1.0;
            else if (c_skin > 0.0) { const double rr = d1 + c_skin; rb0 = fmax(rb0, rr * rr); }
        }
        rowd[wave][lane][0] = s[0]; rowd[wave][lane][1] = s[1]; rowd[wave][lane][2] = s[2];
        rowd[wave][lane][3] = rb0;
        rowd[wave][lane][4] = seed; rowd[wave][lane][5] = bv;
        rowi[wave][lane][0] = bj; rowi[wave][lane][1] = (int32_t)i;
#pragma unroll
        for (int a = 0; a < 3; ++a) { rowk[wave][lane][a] = my_pt[a]; rowk[wave][lane][3 + a] = my_nrm[a]; }
        rowk[wave][lane][6] = __int_as_float(my_prev);
    }
    if (PERSIST && fuse.stamp) {          // who is searched, and why (chain clock: [26] waves, [27] rows, [28] rows without partner, [29] rows without certificate)
        const bool searched = lane < 16 && my_active && !my_certd;
        const unsigned long long sm = __builtin_amdgcn_ballot_w64(searched);
        const unsigned long long np = __builtin_amdgcn_ballot_w64(searched && my_prev < 0);
        const unsigned long long nc = __builtin_amdgcn_ballot_w64(searched && (my_cert & ~63u) == 0u);
        if (lane == 0 && sm) {
            atomicAdd(&fuse.stamp[26], 1ull); atomicAdd(&fuse.stamp[27], (unsigned long long)__builtin_popcountll(sm));
            atomicAdd(&fuse.stamp[28], (unsigned long long)__builtin_popcountll(np)); atomicAdd(&fuse.stamp[29], (unsigned long long)__builtin_popcountll(nc));
        }
    }
    const unsigned act_mask = (unsigned)(__builtin_amdgcn_ballot_w64(lane < 16 && my_active) & 0xFFFFull);
    const unsigned certd_mask = (unsigned)(__builtin_amdgcn_ballot_w64(lane < 16 && my_certd) & 0xFFFFull);
    wave_lds_fence();
    WaveRows w;
    w.a = q < 3 ? rowd[wave][j][q] : 1.0;
    w.rows = &rowd[wave][0][0];
    w.rowsf = &rowf[wave][0][0];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rr = q + 4 * r;
        w.seed[r] = rowd[wave][rr][4];
        w.best[r] = rowd[wave][rr][5];
        w.bcol[r] = rowi[wave][rr][0];
        w.rb0[r] = rowd[wave][rr][3];
    }
    w.skin = c_skin;
    w.act_mask = act_mask;
    w.light_gap2 = -1.0;
    w.dbg = (PERSIST && fuse.stamp && bid == 0 && wave == 0) ? fuse.stamp + 16 : (unsigned long long *)nullptr;
    phase_tick(tile_visits, 2, bid);
    if (PERSIST) chain_tick(fuse.stamp, 1, bid == 0 && tix == 0);
    const unsigned long long t_sweep = tile_visits ? wall_clock64() : 0ull;
    unsigned long long swept = 0ull;
    int lane_p = 0, wave_p = 0, q_p = 0, j_p = 0;
    int64_t row_base_p = 0;
    auto rederive = [&]() {
        lane_p = opaque_i((int)(threadIdx.x & 63)); wave_p = opaque_i((int)(threadIdx.x >> 6)); q_p = lane_p >> 4; j_p = lane_p & 15;
        row_base_p = ((int64_t)bid * kIWaves + wave_p) * kLRows;
    };
    if (act_mask != 0u) {                                 // (a wave whose 16 rows are all certified keeps what it came with)
        wave_lds_fence();                                 // rowd[..][3] is the sweep's own slot from here on
        swept = sweep_wave<true, true, PERSIST>(w, Bs, orig, tile_box, group_box, n_groups, t2max, lists[wave], &gpre);
        rederive();
        // new keys for the rows that were searched: L^2 = min(final culling bound, runner-up among the multiplied columns), both on d^2
        if (certs && fuse.cert_check && j_p == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = q_p + 4 * r;
                if (((certd_mask >> rr) & 1u) != 0u && w.bcol[r] != rowi[wave_p][rr][0]) {
                    if (atomicAdd(&g_cert_check[2], 1ull) == 0ull) {
                        g_cert_check[3] = (unsigned long long)k; g_cert_check[4] = (unsigned long long)(row_base_p + rr);
                        g_cert_check[5] = (unsigned long long)(unsigned)rowi[wave_p][rr][0]; g_cert_check[6] = (unsigned long long)(unsigned)w.bcol[r];
                        g_cert_check[7] = (unsigned long long)((PERSIST ? rowc[wave_p][rr] : fuse.cert[row_base_p + rr]) & ~63u);
                    }
                }
            }
        }
        if (certs && j_p == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = q_p + 4 * r;
                if (((act_mask >> rr) & 1u) != 0u && ((certd_mask >> rr) & 1u) == 0u) {
                    typedef unsigned uu2 __attribute__((ext_vector_type(2)));
                    const uu2 pat = { 0u, w.sec[r] };
                    const double d2nd = w.sec[r] == 0xFFFFFFFFu ? INFINITY : __builtin_bit_cast(double, pat) - 1.0 - w.eps_out;
                    const double l2 = fmin(w.rb_out[r], d2nd);
                    // L rounded DOWN to a float with its low six mantissa bits cleared; those bits carry the iteration (k < 64: later iterations
                    // of a longer chain are searched every time)
                    const uint32_t lb = l2 > 0.0 && k < kCertHist ? (__float_as_uint(f32_down(sqrt(l2) * (1.0 - 1e-7))) & ~63u) : 0u;
                    const uint32_t cw = lb > 63u ? (lb | (uint32_t)k) : 0u;
                    if (PERSIST) rowc[wave_p][rr] = cw; else fuse.cert[row_base_p + rr] = cw;
                }
            }
        }
    }
    if (act_mask == 0u) rederive();
    // how much of the registration is still searched: what the host picks the next launches' form by (icp_rows_kernel once most rows
    // carry a certificate).  One returning add per BLOCK into one of eight words behind the ticket (kSearchedWord: same-address atomics
    // serialise at ~12 ns each -- one word per registration cost 20 us per launch), waited for like the sums' adds.
    __shared__ int s_cnt[kIWaves];
    if (lane_p == 0) s_cnt[wave_p] = __builtin_popcount(act_mask & ~certd_mask);
    const unsigned visited = (unsigned)(swept & 0xFFFFu);
    if (certs && fuse.cert_check && lane_p == 0) {
        if (bid == 0 && wave_p == 0 && g_cert_check[2] == 0ull) {       // no disagreement so far: [3..7] report the chain's state at its last launch
            g_cert_check[3] = (unsigned long long)k; g_cert_check[4] = __builtin_bit_cast(unsigned long long, st->last_motion);
            g_cert_check[5] = __builtin_bit_cast(unsigned long long, st->motion); g_cert_check[6] = __builtin_bit_cast(unsigned long long, c_skin);
        }
        atomicAdd(&g_cert_check[0], (unsigned long long)__builtin_popcount(certd_mask));
        atomicAdd(&g_cert_check[1], (unsigned long long)__builtin_popcount(act_mask & ~certd_mask));
    }
    // (LightSkip speaks for ALL rows of a block: a wave that left certified rows out of its box does not count as light)
    if (lane_p == 0) s_light[wave_p] = act_mask == 0xFFFFu ? w.light_gap2 : -1.0;
    if (tile_visits && lane_p == 0 && bid < kStampBlocks && kIWaves <= 4) {
        unsigned long long *o = g_icp_wave[bid * 4 + wave_p];
        int with = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) with += (w.bcol[r] >= 0 && w.bcol[r] != INT_MAX) ? 1 : 0;      // lane 0: rows 0, 4, 8, 12 (a sample)
        o[0] = t_sweep; o[1] = wall_clock64(); o[2] = swept; o[3] = (unsigned long long)with;
    }
    phase_tick(tile_visits, 3, bid);
    if (PERSIST) chain_tick(fuse.stamp, 2, bid == 0 && lane_p == 0 && wave_p == 0);
    if (PERSIST && fuse.stamp && bid == 0 && lane_p == 0 && wave_p == 0) { fuse.stamp[13] = swept; fuse.stamp[14] = (unsigned long long)act_mask | ((unsigned long long)certd_mask << 16); }
    if (tile_visits && lane_p == 0) atomicAdd(tile_visits + ((bid * kIWaves + wave_p) & (kVisitSlots - 1)), (unsigned long long)visited);
    wave_lds_fence();
    if (j_p == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rowi[wave_p][q_p + 4 * r][0] = w.bcol[r];
    }
    wave_lds_fence();

    // the chosen pairs: direct distance, contribution to the sums (one row per lane 0..15)
    // (lane / wave numbers re-derived behind an opaque move: the LDS addresses the epilogue needs are then computed HERE instead of being
    // carried through the sweep -- the kernel sits on its 168-register budget and carried addresses were spilled to scratch memory,
    // i.e. to HBM traffic)
    const int lane_e = lane_p, wave_e = wave_p;
    if (lane_e < 16) {
        const int col = wave_e * 16 + lane_e;
        for (int a = 0; a < nacc; ++a) sh[a][col] = 0.0;
        if (row_base_p + lane_e <= last) {
            const int32_t bj = rowi[wave_e][lane_e][0];
            const int64_t i = rowi[wave_e][lane_e][1];
            const bool none = bj < 0 || bj == INT_MAX;
            // Partners in the caller's row order (idx_cur / d2_cur: scattered 4- and 8-byte stores) only where a caller asked for
            // them (kpx_icp with idx / d2 outputs); the sorted-order copies the NEXT launch bounds its rows with only when the
            // partner changed -- in the late iterations of a registration almost no row changes its partner.
            const int32_t out_j = none ? -1 : bj;
            const int32_t prev_j = __float_as_int(rowk[wave_e][lane_e][6]);
            const bool changed = k == 0 || out_j != prev_j;
            if (idx_cur) idx_cur[i] = out_j;
            if (PERSIST) rowk[wave_e][lane_e][6] = __int_as_float(out_j);
            else if (changed) idx_sorted[row_base_p + lane_e] = out_j;
            if (none) {
                if (d2_cur) d2_cur[i] = INFINITY;
            } else {
                const double s[3] = { rowd[wave_e][lane_e][0], rowd[wave_e][lane_e][1], rowd[wave_e][lane_e][2] };
                // The partner's coordinates and its normal are gathered through the index only where the partner CHANGED: those of an
                // unchanged partner came at the launch's start (coordinates with the row: ptgt_sorted; the normal through the previous
                // index).  In the late iterations whole waves skip this dependent round trip; both parts of a changed partner are
                // requested together.
                float tf[3] = { rowk[wave_e][lane_e][0], rowk[wave_e][lane_e][1], rowk[wave_e][lane_e][2] }, nf[3] = { rowk[wave_e][lane_e][3], rowk[wave_e][lane_e][4], rowk[wave_e][lane_e][5] };
                if (changed) {
                    const float *tp = tgt + 3 * (int64_t)bj;
#pragma unroll
                    for (int c = 0; c < 3; ++c) tf[c] = tp[c];
                    if (mode == 1) {
                        const float *np_ = tn + 3 * (int64_t)bj;
#pragma unroll
                        for (int c = 0; c < 3; ++c) nf[c] = np_[c];
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {                                                   // the next iteration bounds this row with it
                        if (PERSIST) { rowk[wave_e][lane_e][c] = tf[c]; rowk[wave_e][lane_e][3 + c] = nf[c]; }
                        else ptgt_sorted[3 * (row_base_p + lane_e) + c] = tf[c];
                    }
                }
                const double t[3] = { (double)tf[0], (double)tf[1], (double)tf[2] };
                const double dx = s[0] - t[0], dy = s[1] - t[1], dz = s[2] - t[2];
                const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                if (d2_cur) d2_cur[i] = d2;
                if (d2 < max_d2) {
                    sh[0][col] = 1.0; sh[1][col] = d2;
#pragma unroll
                    for (int c = 0; c < 3; ++c) { sh[2 + c][col] = s[c]; sh[5 + c][col] = t[c]; }
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int c = 0; c < 3; ++c) sh[8 + 3 * a + c][col] = t[a] * s[c];
                    if (mode == 1) {
                        const double nx = nf[0], ny = nf[1], nz = nf[2];
                        const double res = (s[0] - t[0]) * nx + (s[1] - t[1]) * ny + (s[2] - t[2]) * nz;
                        const double J[6] = { s[1] * nz - s[2] * ny, s[2] * nx - s[0] * nz, s[0] * ny - s[1] * nx, nx, ny, nz };
                        int slot = 17;
#pragma unroll
                        for (int a = 0; a < 6; ++a)
#pragma unroll
                            for (int c = a; c < 6; ++c) sh[slot++][col] = J[a] * J[c];
#pragma unroll
                        for (int a = 0; a < 6; ++a) sh[38 + a][col] = J[a] * res;
                    }
                }
            }
        }
    }
    __syncthreads();
    phase_tick(tile_visits, 4, bid);
    // (the thread number re-derived behind the sweep, like the lane state of the pair epilogue: the accumulator slot's address, a 64-bit
    // value per lane known from the kernel's first instruction, was otherwise computed there and carried -- spilled -- across the sweep;
    // every spilled dword is 256 B of scratch per wave written back to HBM at the end of the launch: 0.5 MB per launch at 31k rows)
    const int tix_e = wave_e * 64 + lane_e;
    if (tix_e < nacc) {
        // The sums' contract (round 5): per 16-row TILE a balanced tree over adjacent rows (tile_tree16: what four DPP steps give a
        // wave that holds one row per lane, icp_rows_kernel), the tiles' partials then added EXACTLY in 128-bit fixed point -- so the
        // totals do not depend on how tiles are dealt out to waves, blocks or launches, and every form of the iteration agrees bit for bit.
        unsigned long long lo = 0ull, hi = 0ull;
#pragma unroll
        for (int t = 0; t < kIWaves; ++t) {
            unsigned long long l, h;
            fixed_split(tile_tree16(&sh[tix_e][16 * t]), l, h);
            fixed_accumulate(lo, hi, l, h);
        }
        unsigned long long *slot = acc + (((int64_t)(bid & (kAccCopies - 1)) * kAcc + tix_e) * kFixedWords);
        if (fuse.ticket) fixed_add_words_performed(slot, lo, hi); else fixed_add_words(slot, lo, hi);
    } else if (!PERSIST && fuse.ticket && tix_e == nacc) {
        int cnt = 0;
#pragma unroll
        for (int wv = 0; wv < kIWaves; ++wv) cnt += s_cnt[wv];
        if (cnt) {
            const unsigned long long back = __hip_atomic_fetch_add(fuse.ticket + kSearchedWord + (bid & 7u), (unsigned long long)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(back));
        }
    }
    if (fuse.light_key && tix == kIThreads - 1) {      // (after the barrier above: s_light is complete)
        double g2 = INFINITY;
        bool light = true;
#pragma unroll
        for (int wv = 0; wv < kIWaves; ++wv) { light = light && s_light[wv] >= 0.0; g2 = fmin(g2, s_light[wv]); }
        const double nk = light ? st->motion + sqrt(g2) * (1.0 - 1e-9) : 0.0;
        if (PERSIST) s_key = nk; else fuse.light_key[bid] = nk;
    }
    phase_tick(tile_visits, 5, bid);
    } while (false);
    if (PERSIST && fuse.stamp && tix == 0) {
        const unsigned long long t = wall_clock64();
        if (bid == 0) fuse.stamp[3] = t;
        atomicMax(&fuse.stamp[11], t);
    }
    if (!fuse.ticket) return;
    // "The last block finishes the job": every add above has RETURNED (it has been performed at the device's point of coherence),
    // the barrier orders the block's ticket behind them, and the block that draws the last ticket of its registration reads the
    // totals -- 8 x 44 pairs of words -- with device-coherent loads, clears them for the next launch and performs the update.
    // Only relaxed atomics on the producers' side: no release fence, which on this part writes back the XCD's L2 (measured 10x slower,
    // once per block).  The CONSUMER side is by the book: the winner -- one block per registration and launch -- acquires at agent
    // scope behind its ticket (a single buffer_inv; same-box A/B against none: equal within noise, profiles/r03/exp_icp_acquire_fence.txt).
    // This is the `sc1` form of the valid hand-offs of MI355X_MICROARCH.md ("Correctness boundaries"): the handed-off bytes are
    // produced by atomics (performed at the memory side, never resident in a CU's L1), drained before the ticket because every
    // add RETURNS, and read by the winner with device-coherent (sc1) loads only (fixed_total_coherent) -- no plain load of them
    // anywhere.  It rests on gfx950 behaviour, not on the HIP memory model: KPX_ICP_SPLIT=1 (update in its own kernel, ordered by the
    // kernel boundary) is the portable fall-back, and test_update_placements_agree_with_four_frames_in_flight compares the three
    // placements bit for bit under four frames in flight.
    // The state is written with plain stores: its readers are the blocks of the NEXT launch, behind the kernel boundary.
    __shared__ unsigned s_ticket;
    __syncthreads();
    if (tix == 0) s_ticket = (unsigned)__hip_atomic_fetch_add(fuse.ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (PERSIST && fuse.stamp && tix == 0) {
        const unsigned long long t = wall_clock64();
        if (bid == 0) fuse.stamp[4] = t;
        atomicMax(&fuse.stamp[6], t);
    }
    if (s_ticket != nblocks - 1u) return;
#if KPX_ICP_ACQ_FENCE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the winner only: one per registration and launch
#endif
    // (the winner's thread number behind an opaque move as well: addresses derived from it are then formed here, not in front of the sweep)
    const int tix_w = opaque_i((int)threadIdx.x);
    if (tix_w < kAcc) s_sums[tix_w] = tix_w < nacc ? fixed_total_coherent(acc, tix_w) : 0.0;
    const unsigned long long n_searched = PERSIST ? 0ull : searched_take(fuse.ticket, tix_w);
    __syncthreads();
    if (PERSIST) chain_tick(fuse.stamp, 7, tix == 0);
    for (int e = tix_w; e < kAccSet; e += kIThreads) __hip_atomic_store(acc + e, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tix == 0) __hip_atomic_store(fuse.ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // PERSIST: the blocks that see the next record add to these accumulators at once, so every clearing store (and the ticket's) must
    // have been performed before the record is published: each wave drains its stores, the block meets at a barrier (wave 0 after the
    // update algebra, which hides the drain), then wave 0 publishes
    if (PERSIST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave != 0) {
        if (PERSIST) __syncthreads();
        return;
    }
    __shared__ FinishScratch s_tail;
    // (t2max again from the target's box, behind an opaque move of its address: a double every lane would otherwise carry across the sweep)
    const double *tbbox_w = tbbox;
    asm volatile("" : "+s"(tbbox_w));
    const double t2max_w = target_t2max(tbbox_w);
    IcpState *stw = const_cast<IcpState *>(st);
    IcpState *work = KPX_ICP_STATE_LDS ? &s_state : stw;
    if (PERSIST)
        icp_finish_wave_call(s_sums, n, mode, k, fuse.max_iter, fuse.rel_fit, fuse.rel_rmse, work, (double *)nullptr, &s_tail, lane,
                             fuse.light_key ? fuse.sbbox : (const double *)nullptr, max_d2, t2max_w, fuse.stamp ? fuse.stamp + 32 : (unsigned long long *)nullptr);
    else
        icp_finish_wave(s_sums, n, mode, k, fuse.max_iter, fuse.rel_fit, fuse.rel_rmse, work, fuse.result, s_tail, lane,
                        LightSkip{ fuse.light_key ? fuse.sbbox : (const double *)nullptr, max_d2, t2max_w });
    wave_lds_fence();
    if (PERSIST) chain_tick(fuse.stamp, 8, lane == 0);
    if (PERSIST) {
        static_assert(KPX_ICP_STATE_LDS, "the chain form updates the LDS copy of the state");
        static_assert(sizeof(IcpState) == kChainWords * sizeof(double), "record layout");
        __syncthreads();
        if (lane < kChainWords)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(fuse.chain_rec + (size_t)kChainRec * (k + 1)) + lane,
                               reinterpret_cast<const unsigned long long *>(&s_state)[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the result is written ONCE, by the winner of the last iteration: the winners of a chain sit on different XCDs, and plain stores of
        // several of them to the same words would reach memory in whatever order their L2s are written back at the end of the kernel
        if (work->done && fuse.result) {
            if (lane < 16) fuse.result[lane] = work->T[lane];
            if (lane == 0) { fuse.result[16] = work->fitness; fuse.result[17] = work->rmse; fuse.result[18] = (double)k; fuse.result[19] = work->count; }
        }
        if (lane == 0 && fuse.progress && work->done)
            __hip_atomic_store(fuse.progress, fuse.tag | (1ull << 32) | (unsigned long long)(unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        chain_tick(fuse.stamp, 9, lane == 0);
        if (fuse.stamp && lane == 0) fuse.stamp[12] = __builtin_amdgcn_s_memtime();      // shader clock (against [9]: the clock the chip runs the chain at)
        return;
    }
    if (KPX_ICP_STATE_LDS && lane < (int)(sizeof(IcpState) / sizeof(double))) reinterpret_cast<double *>(stw)[lane] = reinterpret_cast<const double *>(&s_state)[lane];
    if (fuse.cert && fuse.thist && k + 1 < kCertHist && lane < 12) fuse.thist[12 * (k + 1) + lane] = work->T[lane];     // what iteration k + 1 transforms with
    if (lane == 0 && fuse.progress)
        __hip_atomic_store(fuse.progress, fuse.tag | progress_searched(n_searched, n) | ((unsigned long long)(work->done ? 1 : 0) << 32) | (unsigned long long)(unsigned)(k + 1),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}


__global__ __launch_bounds__(kIThreads) __attribute__((amdgpu_waves_per_eu(KPX_ICP_WPE, KPX_ICP_WPE))) void icp_iter_kernel(const float *__restrict__ src, int64_t n, const float *__restrict__ tgt,
                                                       const float *__restrict__ tn, const double *__restrict__ Bs,
                                                       const int32_t *__restrict__ orig, const float *__restrict__ tile_box,
                                                       const float *__restrict__ group_box, int32_t n_groups,
                                                       const double *__restrict__ tbbox, const int32_t *__restrict__ row_of,
                                                       const float *__restrict__ src_sorted, int32_t *__restrict__ idx_sorted,
                                                       float *__restrict__ ptgt_sorted,
                                                       int32_t *__restrict__ idx_cur, double *__restrict__ d2_cur, double max_d2, int mode,
                                                       int k, const IcpState *__restrict__ st, unsigned long long *acc,
                                                       unsigned long long *__restrict__ tile_visits, IcpFuse fuse)
{
    icp_iter_body(blockIdx.x, gridDim.x, src, n, tgt, tn, Bs, orig, tile_box, group_box, n_groups, tbbox, row_of, src_sorted, idx_sorted, ptgt_sorted,
                  idx_cur, d2_cur, max_d2, mode, k, st, acc, tile_visits, fuse);
}

// Several registrations onto ONE shared target in one launch per iteration (kpx_icp_batch): block b belongs to the problem
// whose block range holds it.  A frame's three or seven registrations then cost one chain of launches instead of three or
// seven -- every kernel boundary writes back / invalidates the XCDs' L2s for everything else running on the device, so the
// number of launches per frame, not their size, is what the frame rate of the pipeline follows.  A problem that has
// converged keeps its blocks in the later launches: they read its state and return.
constexpr int kIcpBatchMax = 8;
struct IcpProblem {
    const float *src;
    const int32_t *row_of;
    float *src_sorted;
    int32_t *idx_sorted;
    float *ptgt_sorted;
    int32_t *idx_cur;
    double *d2_cur;
    IcpState *pair;
    unsigned long long *ring;
    double *result;
    unsigned long long *progress;
    double *light_key;
    const double *sbbox;
    uint32_t *cert;
    double *thist;
    double *chain_rec;
    int64_t n;
    uint32_t block0, blocks;
    // round 5: every problem carries its own target operands, iteration number and progress tag, so that one launch can hold the
    // registrations of SEVERAL frames in flight, each at the iteration it has reached (IcpEngine below; icp_chain_kernel, which iterates
    // by itself over one shared target, takes these as kernel arguments instead)
    const float *tgt, *tn;
    const double *Bs;
    const int32_t *orig;
    const float *tile_box, *group_box;
    const double *tbbox;
    int32_t n_groups, k;
    unsigned long long tag;
};
struct IcpBatchArgs {
    IcpProblem p[kIcpBatchMax];
    int32_t count;
};
__global__ __launch_bounds__(kIThreads) __attribute__((amdgpu_waves_per_eu(KPX_ICP_WPE, KPX_ICP_WPE))) void icp_iter_batch_kernel(IcpBatchArgs args, double max_d2, int mode, int max_iter, double rel_fit,
                                                       double rel_rmse, unsigned long long *__restrict__ tile_visits, int split, int light,
                                                       CertPolicy pol)
{
    int pi = 0;
#pragma unroll
    for (int c = 1; c < kIcpBatchMax; ++c) pi += (c < args.count && blockIdx.x >= args.p[c].block0) ? 1 : 0;
    const IcpProblem &P = args.p[pi];
    const unsigned bid = blockIdx.x - P.block0;
    const float *__restrict__ tgt = P.tgt, *__restrict__ tn = P.tn, *__restrict__ tile_box = P.tile_box, *__restrict__ group_box = P.group_box;
    const double *__restrict__ Bs = P.Bs, *__restrict__ tbbox = P.tbbox;
    const int32_t *__restrict__ orig = P.orig;
    const int32_t n_groups = P.n_groups;
    const int k = P.k;
    const unsigned long long tag = P.tag;
    if (k > max_iter && bid != 0) return;                   // the closing launch only performs the last update (one block per problem)
    // split: the update runs in icp_solve_batch_kernel between the sweeps (state slot 0, first accumulator set): the sweep's blocks
    // then live 8 us instead of 12 -- under load (several frames in flight) the device's wave slots are what the sweeps compete for
    // split == 2: no update kernel either -- the last block of every registration's sweep performs it (ticket: first word of the
    // second accumulator set, which only the one-launch form uses)
    const IcpFuse fuse{ split ? (IcpState *)nullptr : P.pair, P.ring, max_iter, rel_fit, rel_rmse, P.result, P.progress, tag,
                        split == 2 ? P.ring + kAccSet : (unsigned long long *)nullptr, (light & 1) ? P.light_key : (double *)nullptr, P.sbbox,
                        (light & 2) ? P.cert : (uint32_t *)nullptr, (light & 2) ? P.thist : (double *)nullptr, (light & 4) ? 1 : 0, pol, nullptr, nullptr };
    icp_iter_body(bid, P.blocks, P.src, P.n, tgt, tn, Bs, orig, tile_box, group_box, n_groups, tbbox, P.row_of, P.src_sorted, P.idx_sorted, P.ptgt_sorted,
                  P.idx_cur, P.d2_cur, max_d2, mode, k, P.pair, P.ring, tile_visits, fuse);
}

// The whole chain of a group of registrations in ONE launch: block b iterates over k on the rows it owns (icp_iter_body<true>), the
// block that draws the last ticket of iteration k performs the update and publishes record k + 1, everybody else waits for it (see
// kChainRec).  No kernel boundary, no host poll, no re-read of the rows: an iteration costs the ticket, the update algebra and one
// coherent round trip instead of a launch.
// REQUIRES every block of the launch to be resident at the same time (a block that is not can never deliver its sums): the host
// launches this form only when the grid fits the device beside every other chain kernel it has in flight (chain_reserve), and every
// wait is bounded -- a block that has waited `limit_ticks` (100 MHz wall clock, counted from ITS start) raises *abort_word (pinned
// host memory), poisons its registration's result and leaves; the others follow on their own clocks.  The library reports a raised
// word at the next call (KPX_ERR_HIP); KPX_ICP_CHAIN=0 selects the launch-per-iteration form.
__global__ __launch_bounds__(kIThreads) __attribute__((amdgpu_waves_per_eu(KPX_ICP_WPE, KPX_ICP_WPE))) void icp_chain_kernel(IcpBatchArgs args, const float *__restrict__ tgt,
                                                       const float *__restrict__ tn, const double *__restrict__ Bs,
                                                       const int32_t *__restrict__ orig, const float *__restrict__ tile_box,
                                                       const float *__restrict__ group_box, int32_t n_groups,
                                                       const double *__restrict__ tbbox, double max_d2, int mode, int max_iter, double rel_fit,
                                                       double rel_rmse, unsigned long long tag, unsigned long long *__restrict__ tile_visits, int light,
                                                       CertPolicy pol, unsigned long long limit_ticks, unsigned long long *abort_word)
{
    int pi = 0;
#pragma unroll
    for (int c = 1; c < kIcpBatchMax; ++c) pi += (c < args.count && blockIdx.x >= args.p[c].block0) ? 1 : 0;
    const IcpProblem &P = args.p[pi];
    const unsigned bid = blockIdx.x - P.block0;
    __shared__ IcpState s_cur;
    __shared__ int s_abort;
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    for (int k = 0; k <= max_iter; ++k) {
        // (the operands' addresses behind opaque moves, per iteration: their loop-invariant loads -- the first group boxes, the problem's
        // descriptor -- would otherwise be hoisted out of the loop and live in registers across it)
        asm volatile("" : "+s"(tgt), "+s"(tn), "+s"(Bs), "+s"(orig), "+s"(tile_box), "+s"(group_box), "+s"(tbbox));
        asm volatile("" : "+s"(max_d2), "+s"(mode), "+s"(n_groups), "+s"(pol.calm), "+s"(pol.factor), "+s"(pol.smin), "+s"(pol.smax));
        const IcpFuse fuse{ (IcpState *)nullptr, P.ring, max_iter, rel_fit, rel_rmse, P.result, P.progress, tag, P.ring + kAccSet,
                            (light & 1) ? P.light_key : (double *)nullptr, P.sbbox, (light & 2) ? P.cert : (uint32_t *)nullptr,
                            (light & 2) ? P.thist : (double *)nullptr, (light & 4) ? 1 : 0, pol, P.chain_rec,
                            ((light & 8) && pi == 0 && k < 64) ? &g_chain_stamp[k][0] : (unsigned long long *)nullptr };
        if (threadIdx.x < kChainWords) {
            const unsigned long long *w = reinterpret_cast<const unsigned long long *>(P.chain_rec + (size_t)kChainRec * k) + threadIdx.x;
            unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (v == kChainEmpty) {
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 255u) == 0u && wall_clock64() - t0 > limit_ticks) { s_abort = 1; break; }
                v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            reinterpret_cast<unsigned long long *>(&s_cur)[threadIdx.x] = v;
        }
        __syncthreads();
        if (s_abort) {
            // the abort is reported PER CALL: every result word of the registration is NaN (nothing stale leaks out, and whoever reads the
            // result -- ops.icp_batch, kpx_frame_step*, the exchange header of the sharded step -- sees it for THIS call); the pinned word is
            // the process-wide diagnostic behind it
            if (threadIdx.x == 0) __hip_atomic_store(abort_word, tag | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (P.result && threadIdx.x < 20) P.result[threadIdx.x] = __builtin_nan("");
            return;
        }
        if (s_cur.done) break;
        icp_iter_body<true>(bid, P.blocks, P.src, P.n, tgt, tn, Bs, orig, tile_box, group_box, n_groups, tbbox, P.row_of, P.src_sorted, P.idx_sorted, P.ptgt_sorted,
                            P.idx_cur, P.d2_cur, max_d2, mode, k, &s_cur, P.ring, tile_visits, fuse);
        __syncthreads();                                     // (the body's early returns meet here before s_cur is written again)
    }
}

// The update step of every registration of a batch, one block each (split mode: see icp_iter_batch_kernel)
__global__ __launch_bounds__(256) void icp_solve_batch_kernel(IcpBatchArgs args, int mode, int k, int max_iter, double rel_fit, double rel_rmse,
                                                              unsigned long long tag)
{
    const IcpProblem &P = args.p[blockIdx.x];
    IcpState *st = P.pair;
    if (st->done) return;
    __shared__ double sums[kAcc];
    __shared__ FinishScratch fs;
    unsigned long long *acc = P.ring;
    const int nacc = mode == 1 ? kAcc : 17;
    if (threadIdx.x < kAcc) sums[threadIdx.x] = (int)threadIdx.x < nacc ? fixed_total(acc, threadIdx.x) : 0.0;
    __syncthreads();
    for (int e = threadIdx.x; e < kAccCopies * kAcc * kFixedWords; e += 256) acc[e] = 0ull;
    if (threadIdx.x >= 64) return;
    icp_finish_wave(sums, P.n, mode, k, max_iter, rel_fit, rel_rmse, st, P.result, fs, (int)threadIdx.x);
    if (threadIdx.x == 0 && P.progress)
        __hip_atomic_store(P.progress, tag | ((unsigned long long)(st->done ? 1 : 0) << 32) | (unsigned long long)(unsigned)(k + 1), __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

// Start of a batch chain in ONE launch: per problem both state slots <- the initial transform, the accumulator ring cleared,
// the rows gathered into Morton order (what icp_init_kernel + a memset + gather_rows_kernel did per problem)
struct Mat16x8 {
    double m[kIcpBatchMax][16];
};
__global__ __launch_bounds__(256) void icp_batch_init_kernel(IcpBatchArgs args, Mat16x8 T0)
{
    int pi = 0;
#pragma unroll
    for (int c = 1; c < kIcpBatchMax; ++c) pi += (c < args.count && blockIdx.x >= args.p[c].block0) ? 1 : 0;
    const IcpProblem &P = args.p[pi];
    const unsigned bid = blockIdx.x - P.block0;
    const int c = threadIdx.x & 3;
    for (int rr = threadIdx.x >> 2; rr < kIRows; rr += 64) {
        const int64_t r = (int64_t)bid * kIRows + rr;
        if (r < P.n && c < 3) P.src_sorted[3 * r + c] = P.src[3 * (int64_t)P.row_of[r] + c];
    }
    if (threadIdx.x < 7) {                                  // the block's LightSkip keys at every granularity: 1 x 64 rows, 2 x 32, 4 x 16
        const int64_t n64 = (P.n + 63) / 64, n32 = (P.n + 31) / 32, n16 = (P.n + 15) / 16;
        const int t = threadIdx.x;
        const int64_t e = t == 0 ? (int64_t)bid : t < 3 ? n64 + 2 * (int64_t)bid + (t - 1) : n64 + n32 + 4 * (int64_t)bid + (t - 3);
        const int64_t lim = t == 0 ? n64 : t < 3 ? n64 + n32 : n64 + n32 + n16;
        if (e < lim) P.light_key[e] = 0.0;
    }
    if (bid == 0) {
        for (int e = threadIdx.x; e < 3 * kAccSet; e += 256) P.ring[e] = 0ull;
        if (threadIdx.x >= 64 && threadIdx.x < 76) P.thist[threadIdx.x - 64] = T0.m[pi][threadIdx.x - 64];
        if (P.chain_rec) {                                  // the chain form: record 0 = the initial state (below), every other record empty
            unsigned long long *rw = reinterpret_cast<unsigned long long *>(P.chain_rec);
            for (int e = kChainRec + threadIdx.x; e < kChainRecords * kChainRec; e += 256) rw[e] = kChainEmpty;
            IcpState *r0 = reinterpret_cast<IcpState *>(P.chain_rec);
            if (threadIdx.x >= 128 && threadIdx.x < 144) r0->T[threadIdx.x - 128] = T0.m[pi][threadIdx.x - 128];
            if (threadIdx.x == 144) { r0->fitness = 0.0; r0->rmse = 0.0; r0->count = 0.0; r0->iter = 0; r0->done = 0; r0->motion = 0.0; r0->reach = INFINITY; r0->last_motion = INFINITY; r0->smax = -1.0; }
        }
        if (threadIdx.x < 32) {
            IcpState *st = P.pair + (threadIdx.x >> 4);
            st->T[threadIdx.x & 15] = T0.m[pi][threadIdx.x & 15];
            if ((threadIdx.x & 15) == 0) { st->fitness = 0.0; st->rmse = 0.0; st->count = 0.0; st->iter = 0; st->done = 0; st->motion = 0.0; st->reach = INFINITY; st->last_motion = INFINITY; st->smax = -1.0; }
        }
    }
}

}  // namespace kpx
#include "kpx_icprows.h"
namespace kpx {

// tiles multiplied by nn_local_kernel while the profiler is armed (one atomic per wave, spread over kVisitSlots
// addresses: same-address atomics from thousands of waves serialise in L2); read by kpx_prof_end
__device__ unsigned long long g_nn_visits[kVisitSlots];
static unsigned long long *nn_visits_ptr()
{
    static unsigned long long *p = nullptr;
    if (!p && hipGetSymbolAddress((void **)&p, HIP_SYMBOL(g_nn_visits)) != hipSuccess) p = nullptr;
    return p;
}
// -> h_out8 (16 doubles): average us per block in the five phases of the LAST sweep launch, [5] blocks of that launch, [6] dispatch
// ramp (latest block start - earliest block start), [7] span (earliest start -> latest end), [8..12] slowest block per phase,
// [13] longest block lifetime
int icp_phase_take(double *h_out8)
{
    static unsigned long long v[kStampBlocks][8];
    unsigned long long *p = nullptr;
    if (hipGetSymbolAddress((void **)&p, HIP_SYMBOL(g_icp_stamp)) != hipSuccess) return KPX_ERR_HIP;
    if (hipMemcpy(v, p, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return KPX_ERR_HIP;
    for (int q = 0; q < 16; ++q) h_out8[q] = 0.0;
    int blocks = (int)v[0][6];
    if (blocks < 1) return KPX_OK;
    if (blocks > kStampBlocks) blocks = kStampBlocks;
    for (int b = 0; b < blocks; ++b) h_out8[14] += v[b][7] ? 1.0 : 0.0;        // blocks LightSkip left out of the last launch
    unsigned long long s_min = ~0ull, s_max = 0ull, e_max = 0ull;
    int counted = 0;
    for (int b = 0; b < blocks; ++b) {
        if (v[b][7] || !(v[b][5] >= v[b][0]) || v[b][1] < v[b][0]) continue;          // a block of an "already converged" launch stamps nothing new
        for (int q = 0; q < 5; ++q) {
            const double d = (double)(v[b][q + 1] - v[b][q]) * 0.01;
            h_out8[q] += d;
            if (d > h_out8[8 + q]) h_out8[8 + q] = d;                      // [8..12]: the slowest block of each phase
        }
        if ((double)(v[b][5] - v[b][0]) * 0.01 > h_out8[13]) h_out8[13] = (double)(v[b][5] - v[b][0]) * 0.01;   // longest block lifetime
        s_min = v[b][0] < s_min ? v[b][0] : s_min;
        s_max = v[b][0] > s_max ? v[b][0] : s_max;
        e_max = v[b][5] > e_max ? v[b][5] : e_max;
        ++counted;
    }
    if (!counted) return KPX_OK;
    for (int q = 0; q < 5; ++q) h_out8[q] /= (double)counted;
    h_out8[5] = (double)counted;
    h_out8[6] = (double)(s_max - s_min) * 0.01;
    h_out8[7] = (double)(e_max - s_min) * 0.01;
    return KPX_OK;
}
// raw rows of g_icp_wave for the blocks of the last sweep launch -> h_out (4 u64 per wave); returns the number of waves
int64_t icp_wave_take(unsigned long long *h_out, int64_t cap_waves)
{
    unsigned long long *p = nullptr, *ps = nullptr;
    if (hipGetSymbolAddress((void **)&p, HIP_SYMBOL(g_icp_wave)) != hipSuccess) return -1;
    if (hipGetSymbolAddress((void **)&ps, HIP_SYMBOL(g_icp_stamp)) != hipSuccess) return -1;
    unsigned long long first[8];
    if (hipMemcpy(first, ps, sizeof(first), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    int64_t waves = (int64_t)first[6] * kIWaves;
    if (waves > kStampBlocks * 4) waves = kStampBlocks * 4;
    if (waves > cap_waves) waves = cap_waves;
    if (waves > 0 && hipMemcpy(h_out, p, (size_t)waves * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return waves;
}
double nn_local_take_visits()
{
    static unsigned long long v[kVisitSlots], zero[kVisitSlots];
    unsigned long long *p = nn_visits_ptr();
    if (!p || hipMemcpy(v, p, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 0.0;
    (void)hipMemcpy(p, zero, sizeof(zero), hipMemcpyHostToDevice);
    double sum = 0.0;
    for (int i = 0; i < kVisitSlots; ++i) sum += (double)v[i];
    return sum;
}

// the one-launch chain: on unless KPX_ICP_CHAIN=0; kpx_icp_chain() switches it at run time (A/B measurements inside one process)
static int g_chain_form = -1;
static int g_chain_launches = 0;                // chains launched by this process (kpx_icp_chain(-2): tests check that the form they test ran)
static bool chain_form_on()
{
    if (g_chain_form < 0) { const char *e = getenv("KPX_ICP_CHAIN"); g_chain_form = (e && e[0] == '0') ? 0 : 1; }
    return g_chain_form != 0;
}
static int g_nn_engine = -1;          // KPX_NN_ENGINE_*; -1 = not chosen yet (environment decides at first use)
static bool g_nn_fp64_only = false;   // KPX_NN_ENGINE_DENSE_FP64: the all-pairs engine without its float32 screening sweep
static bool local_engine()
{
    if (g_nn_engine < 0) { const char *e = getenv("KPX_NN_ENGINE"); g_nn_engine = (e && e[0] == 'd') ? KPX_NN_ENGINE_DENSE : KPX_NN_ENGINE_CULLED; }
    return g_nn_engine == KPX_NN_ENGINE_CULLED;
}

struct NnPlan {
    int64_t tiles_pad, seed_tiles_pad, n_src, n_tgt, f_tiles_pad;
    int32_t l_groups;                                       // culled sweep: groups of 256 sorted target columns
    int32_t tiles_per_split, splits, row_blocks;
    int32_t f_tiles_per_split, f_splits, f_row_blocks;     // float32 screening sweep
};
static NnPlan nn_plan(int64_t n, int64_t m)
{
    NnPlan p;
    p.n_src = n; p.n_tgt = m;
    p.l_groups = (int32_t)cdiv(m > 0 ? m : 1, 16 * kLGroupTiles);
    int64_t tiles = cdiv(m > 0 ? m : 1, 16);
    int64_t stages = cdiv(tiles, kCT);
    p.row_blocks = (int32_t)cdiv(n > 0 ? n : 1, kRowsPerBlock);
    int64_t want = cdiv(4096, p.row_blocks);            // aim for >= ~4096 workgroups (16 per CU, 5 resident)
    if (want > stages) want = stages;
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    int64_t stages_per_split = cdiv(stages, want);
    p.splits = (int32_t)cdiv(stages, stages_per_split);
    p.tiles_per_split = (int32_t)(stages_per_split * kCT);
    p.tiles_pad = (int64_t)p.splits * p.tiles_per_split;
    p.seed_tiles_pad = cdiv(cdiv(tiles, kSeedStride), kCT) * kCT;
    {
        int64_t fstages = cdiv(tiles, kFCT);
        p.f_row_blocks = (int32_t)cdiv(n > 0 ? n : 1, kFRowsPerBlock);
        int64_t fwant = cdiv(2048, p.f_row_blocks);
        if (fwant > fstages) fwant = fstages;
        if (fwant < 1) fwant = 1;
        if (fwant > 64) fwant = 64;
        int64_t per = cdiv(fstages, fwant);
        p.f_splits = (int32_t)cdiv(fstages, per);
        p.f_tiles_per_split = (int32_t)(per * kFCT);
        p.f_tiles_pad = (int64_t)p.f_splits * p.f_tiles_per_split;
    }
    return p;
}

// Which sweep serves iteration k: the float32 screening sweep pays off when the cloud barely moved since the
// last search (short candidate lists); after a large update (first iterations) the bound is loose, lists
// overflow and the exact fp64 sweep is cheaper.  The host sees (fitness, rmse) at every poll and uses their
// relative change as the motion proxy.
struct ScreenPolicy {
    double fit = -1.0, rmse = -1.0;
    bool calm = false;
    void observe(double f, double r)
    {
        calm = fit >= 0.0 && fabs(f - fit) <= 0.02 && fabs(r - rmse) <= 0.05 * (rmse > 1e-12 ? rmse : 1e-12);
        fit = f; rmse = r;
    }
    bool allow(int k) const { return k >= 2 && calm; }
};

struct NnBuffers {
    double *B, *Bseed, *part_val, *part_acc, *init_val, *d2_cur, *tbbox, *A64, *K64;
    float *Bf, *A32, *thr32;
    NnAux *aux;
    int32_t *part_idx, *init_idx, *idx_cur, *cand_cnt, *cand, *overflow;
    int32_t *colB, *colSeed;                            // dense engine: original index of every column of B / Bseed (curve-ordered operand)
    IcpState *state;
    double *T0;
    // culled sweep
    double *Bs;
    int32_t *orig_t, *row_of, *idx_sorted;
    float *src_sorted, *ptgt_sorted;                    // rows in Morton order; coordinates of each row's last partner, same order
    unsigned long long *acc_fixed;                      // [3][kAccCopies][kAcc][kFixedWords] exact accumulators
    double *light_key;                                  // per block of the iteration kernel (LightSkip)
    uint32_t *cert_sorted;                              // per sorted row: certificate (icp_iter_body)
    double *thist;                                      // transforms of the iterations so far (certificates)
    double *chain_rec;                                  // records of the one-launch chain (kChainRecords x kChainRec)
    float *tile_box, *group_box;
    SortScratch sort_t, sort_s;
};
static void nn_carve_target(Arena &a, const NnPlan &p, NnBuffers *b)
{
    b->Bs = a.get<double>((size_t)p.l_groups * kLGroupTiles * 64);
    b->orig_t = a.get<int32_t>((size_t)p.l_groups * kLGroupTiles * 16);
    b->tile_box = a.get<float>((size_t)p.l_groups * kLGroupTiles * 6);
    b->group_box = a.get<float>((size_t)p.l_groups * 6);
    sort_carve(a, p.n_tgt, &b->sort_t);
    b->B = a.get<double>((size_t)p.tiles_pad * 64);
    b->Bseed = a.get<double>((size_t)p.seed_tiles_pad * 64);
    b->colB = a.get<int32_t>((size_t)p.tiles_pad * 16);
    b->colSeed = a.get<int32_t>((size_t)p.seed_tiles_pad * 16);
    b->Bf = a.get<float>((size_t)p.f_tiles_pad * 64);
    b->aux = a.get<NnAux>(1);
    b->tbbox = a.get<double>((size_t)kBboxBlocks * 6 + 8);
}
static void nn_carve_source(Arena &a, int64_t n, const NnPlan &p, NnBuffers *b)
{
    const size_t nn = (size_t)(n > 0 ? n : 1);
    b->part_val = a.get<double>((size_t)p.splits * nn);
    b->part_idx = a.get<int32_t>((size_t)p.splits * nn);
    b->part_acc = a.get<double>((size_t)cdiv((int64_t)nn, kMergeThreads) * kAcc);
    b->init_val = a.get<double>(nn);
    b->init_idx = a.get<int32_t>(nn);
    b->idx_cur = a.get<int32_t>(nn);
    b->d2_cur = a.get<double>(nn);
    b->state = a.get<IcpState>(2);                     // two slots: the one-launch-per-iteration mode alternates between them
    b->T0 = a.get<double>(16);
    b->A64 = a.get<double>(nn * 4);
    b->K64 = a.get<double>(nn);
    b->A32 = a.get<float>(nn * 4);
    b->thr32 = a.get<float>(nn * 2);                     // (row constant, threshold) pairs
    b->cand_cnt = a.get<int32_t>(nn + 1);               // [n] counters + number of overflowed rows
    b->cand = a.get<int32_t>(nn * kCand);
    b->overflow = a.get<int32_t>(nn);                   // list of overflowed rows
    b->row_of = a.get<int32_t>(nn);
    b->idx_sorted = a.get<int32_t>(nn);
    b->src_sorted = a.get<float>(nn * 3);
    b->ptgt_sorted = a.get<float>(nn * 3);
    b->acc_fixed = a.get<unsigned long long>((size_t)3 * kAccCopies * kAcc * kFixedWords + 8);      // ring of three sets (icp_iter_kernel, IcpFuse)
    b->light_key = a.get<double>((size_t)(cdiv((int64_t)nn, 64) + cdiv((int64_t)nn, 32) + cdiv((int64_t)nn, 16)));   // per 64 rows (both forms), then per 32 and per 16 (icp_rows_kernel<., R>)
    b->cert_sorted = a.get<uint32_t>(nn);
    b->thist = a.get<double>((size_t)kCertHist * 12);
    b->chain_rec = a.get<double>((size_t)kChainRecords * kChainRec);
    sort_carve(a, n, &b->sort_s);
}
static void nn_carve(Arena &a, int64_t n, int64_t m, const NnPlan &p, NnBuffers *b)
{
    nn_carve_target(a, p, b);
    nn_carve_source(a, n, p, b);
}

static bool screening_enabled()
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("KPX_NN_SCREEN"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0 && !g_nn_fp64_only;
}
static __global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src, int64_t n, const int32_t *__restrict__ row_of,
                                                                 float *__restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t i = row_of[r];
    out[3 * r] = src[3 * i]; out[3 * r + 1] = src[3 * i + 1]; out[3 * r + 2] = src[3 * i + 2];
}
// ordered: b.row_of already holds the Morton order (morton_order_batch)
static bool dense_sort_on()
{
    static const bool on = [] { const char *e = getenv("KPX_NN_DENSE_SORT"); return !(e && e[0] == '0'); }();      // A/B switch: the all-pairs operands in curve order
    return on;
}
static int nn_prep_source(const float *src, const NnPlan &p, const NnBuffers &b, hipStream_t st, bool ordered = false)
{
    if (!local_engine()) return dense_sort_on() ? morton_order(src, p.n_src, b.sort_s, b.row_of, st) : KPX_OK;
    KPX_HIP(hipMemsetAsync(b.acc_fixed, 0, ((size_t)3 * kAccCopies * kAcc * kFixedWords + 8) * sizeof(unsigned long long), st));
    int rc = ordered ? KPX_OK : morton_order(src, p.n_src, b.sort_s, b.row_of, st);
    if (rc) return rc;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)cdiv(p.n_src, 256)), dim3(256), 0, st, src, p.n_src, b.row_of, b.src_sorted);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}
// one ICP iteration (search k + update) of the culled engine: two launches
static void icp_iter_launch(const float *src, const float *tgt, const float *tn, const NnPlan &p, const NnBuffers &b, double max_d2, int mode,
                            int k, int max_iter, double rel_fit, double rel_rmse, double *d_result, hipStream_t st,
                            unsigned long long *progress = nullptr, unsigned long long tag = 0, bool want_pairs = true)
{
    {
        ProfScope prof(KPX_PROF_NN_LOCAL, 0.0, st);
        hipLaunchKernelGGL(icp_iter_kernel, dim3((unsigned)cdiv(p.n_src, kIRows)), dim3(kIThreads), 0, st, src, p.n_src, tgt, tn, b.Bs, b.orig_t,
                           b.tile_box, b.group_box, p.l_groups, b.sort_t.bbox, b.row_of, b.src_sorted, b.idx_sorted, b.ptgt_sorted,
                           want_pairs ? b.idx_cur : (int32_t *)nullptr, want_pairs ? b.d2_cur : (double *)nullptr, max_d2, mode, k, b.state,
                           b.acc_fixed, prof_armed() ? nn_visits_ptr() : (unsigned long long *)nullptr, IcpFuse{});
    }
    hipLaunchKernelGGL(icp_solve_fixed_kernel, dim3(1), dim3(256), 0, st, b.acc_fixed, p.n_src, mode, k, max_iter, rel_fit, rel_rmse, b.state,
                       d_result, progress, tag);
}
// one ICP iteration in ONE launch (IcpFuse): launch k = update of iteration k-1 + search k; k = max_iter + 1 is the closing
// launch (update only: one block)
static void icp_fused_launch(const float *src, const float *tgt, const float *tn, const NnPlan &p, const NnBuffers &b, double max_d2, int mode,
                             int k, int max_iter, double rel_fit, double rel_rmse, double *d_result, hipStream_t st,
                             unsigned long long *progress, unsigned long long tag)
{
    const IcpFuse fuse{ b.state, b.acc_fixed, max_iter, rel_fit, rel_rmse, d_result, progress, tag, nullptr, nullptr, nullptr, nullptr, nullptr, 0, CertPolicy{ 0.0f, 0.0f, 0.0f, 0.0f } };
    const unsigned blocks = k > max_iter ? 1u : (unsigned)cdiv(p.n_src, kIRows);
    ProfScope prof(KPX_PROF_NN_LOCAL, 0.0, st);
    hipLaunchKernelGGL(icp_iter_kernel, dim3(blocks), dim3(kIThreads), 0, st, src, p.n_src, tgt, tn, b.Bs, b.orig_t, b.tile_box, b.group_box,
                       p.l_groups, b.sort_t.bbox, b.row_of, b.src_sorted, b.idx_sorted, b.ptgt_sorted, b.idx_cur, b.d2_cur, max_d2, mode, k, b.state,
                       b.acc_fixed, prof_armed() ? nn_visits_ptr() : (unsigned long long *)nullptr, fuse);
}
// ordered: b.orig_t / b.sort_t.bbox already hold the target's Morton order and bounding box (morton_order_batch)
static int nn_prep(const float *tgt, const NnPlan &p, const NnBuffers &b, hipStream_t st, bool ordered = false)
{
    if (local_engine()) {
        int rc = ordered ? KPX_OK : morton_order(tgt, p.n_tgt, b.sort_t, b.orig_t, st);
        if (rc) return rc;
        hipLaunchKernelGGL(nn_local_prep_kernel, dim3((unsigned)p.l_groups), dim3(256), 0, st, tgt, p.n_tgt, b.Bs, b.orig_t, b.tile_box,
                           b.group_box);
        KPX_LAUNCH_CHECK();
        return KPX_OK;
    }
    // the all-pairs operand in the target's curve order (KPX_NN_DENSE_SORT=0: in the caller's order, as until round 3)
    const bool dense_sort = dense_sort_on();
    if (dense_sort) {
        int rc = morton_order(tgt, p.n_tgt, b.sort_t, b.orig_t, st);
        if (rc) return rc;
    }
    int64_t work = (p.tiles_pad + p.seed_tiles_pad) * 16;
    hipLaunchKernelGGL(nn_prep_kernel, dim3((unsigned)(cdiv(work, 256) > 2048 ? 2048 : cdiv(work, 256))), dim3(256), 0, st, tgt,
                       p.n_tgt, p.tiles_pad, b.B, p.seed_tiles_pad, b.Bseed, dense_sort ? b.orig_t : (const int32_t *)nullptr, b.colB, b.colSeed);
    {   // float32 screening operand: centre + radius from the target's bounding box
        double *bbox = b.tbbox + (size_t)kBboxBlocks * 6;
        int rc = bbox_f32(tgt, p.n_tgt, bbox, b.tbbox, st);
        if (rc) return rc;
        hipLaunchKernelGGL(nn_aux_kernel, dim3(1), dim3(1), 0, st, bbox, b.aux);
        int64_t fw = p.f_tiles_pad * 16;
        hipLaunchKernelGGL(nn_prep_f32_kernel, dim3((unsigned)(cdiv(fw, 256) > 2048 ? 2048 : cdiv(fw, 256))), dim3(256), 0, st, tgt,
                           p.n_tgt, p.f_tiles_pad, b.aux, b.Bf);
    }
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

// One correspondence search.  have_prev: b.idx_cur holds the partners of the previous search (bound from them),
// otherwise a seed sweep over every 64th target tile provides the bound.
static int nn_search_launch(const float *src, const float *tgt, const float *tn, const NnPlan &p, const NnBuffers &b,
                            const double *T, const int32_t *done, bool have_prev, bool allow_screen, double max_d2, int mode, hipStream_t st,
                            ColorTerms ct = ColorTerms{})
{
    const int64_t n = p.n_src;
    const dim3 thr(256);
    if (local_engine()) {
        hipLaunchKernelGGL(nn_local_rowprep_kernel, dim3((unsigned)cdiv(n, 256)), thr, 0, st, src, n, tgt, T, done, b.row_of,
                           have_prev ? b.idx_cur : (const int32_t *)nullptr, mode >= 0 ? max_d2 : 0.0, b.sort_t.bbox, b.init_val, b.init_idx,
                           b.A64, b.K64);
        {
            ProfScope prof(KPX_PROF_NN_LOCAL, 0.0, st);
            hipLaunchKernelGGL(nn_local_kernel, dim3((unsigned)cdiv(n, kLRows)), dim3(64), 0, st, n, b.Bs, b.orig_t, b.tile_box, b.group_box,
                               p.l_groups, b.sort_t.bbox, done, b.A64, b.K64, b.init_val, b.init_idx, b.row_of, b.part_val, b.part_idx,
                               prof_armed() ? nn_visits_ptr() : (unsigned long long *)nullptr);
        }
        hipLaunchKernelGGL(nn_merge_kernel, dim3((unsigned)cdiv(n, kMergeThreads)), dim3(kMergeThreads), 0, st, src, n, tgt, tn, T, done,
                           b.part_val, b.part_idx, 1, max_d2, mode, b.idx_cur, b.d2_cur, (double *)nullptr, b.part_acc,
                           (const int32_t *)nullptr, (const int32_t *)nullptr, ct);
        KPX_LAUNCH_CHECK();
        return KPX_OK;
    }
    const bool screen = have_prev && allow_screen && screening_enabled();
    hipLaunchKernelGGL(nn_rowprep_kernel, dim3((unsigned)cdiv(n, 256)), thr, 0, st, src, n, tgt, T, done,
                       have_prev ? b.idx_cur : (const int32_t *)nullptr, screen ? b.aux : (const NnAux *)nullptr, b.init_val, b.init_idx,
                       b.A64, b.K64, b.A32, b.thr32, b.cand_cnt);
    if (screen) {
        {
            ProfScope prof(KPX_PROF_NN_SCREEN, 8.0 * (double)p.n_src * (double)p.n_tgt, st);   // 4 MAC per (source, target) pair
            hipLaunchKernelGGL((nn_screen_kernel<4, 3>), dim3(p.f_row_blocks, p.f_splits), thr, 0, st, n, b.Bf, p.f_tiles_per_split, done,
                               b.A32, b.thr32, b.init_idx, b.cand_cnt, b.cand, b.overflow);
        }
        hipLaunchKernelGGL(nn_overflow_kernel, dim3(1024), thr, 0, st, src, n, tgt,
                           p.n_tgt, T, done, b.cand_cnt, b.cand, b.overflow);
        hipLaunchKernelGGL(nn_merge_kernel, dim3((unsigned)cdiv(n, kMergeThreads)), dim3(kMergeThreads), 0, st, src, n, tgt, tn, T, done, b.init_val, b.init_idx, 1,
                           max_d2, mode, b.idx_cur, b.d2_cur, (double *)nullptr, b.part_acc, b.cand_cnt, b.cand, ct);
        KPX_LAUNCH_CHECK();
        return KPX_OK;
    }
    const int32_t *rperm = dense_sort_on() ? b.row_of : (const int32_t *)nullptr;     // (nn_prep_source ordered the rows)
    {
    // (timed as ONE unit: a bare search's seed sweep + its merge belong to the all-pairs sweep they make cheaper -- with bounds from
    // every 16th tile the main sweep alone runs at 38 TFLOP/s, from every 64th at 33, but the seed sweep costs what it saves beyond that)
    ProfScope prof(KPX_PROF_NN_MFMA, 8.0 * (double)p.n_src * (double)p.n_tgt, st);     // 4 MAC per (source, target) pair
    if (!have_prev) {
        hipLaunchKernelGGL(nn_mfma_kernel<false>, dim3(p.row_blocks, 1), thr, 0, st, n, b.Bseed, b.colSeed, (int32_t)p.seed_tiles_pad,
                           done, b.A64, b.K64, (const double *)nullptr, (const int32_t *)nullptr, b.part_val, b.part_idx, rperm);
        hipLaunchKernelGGL(nn_merge_kernel, dim3((unsigned)cdiv(n, kMergeThreads)), dim3(kMergeThreads), 0, st, src, n, tgt, tn, T, done, b.part_val, b.part_idx, 1,
                           0.0, -2, b.init_idx, (double *)nullptr, b.init_val, b.part_acc, (const int32_t *)nullptr, (const int32_t *)nullptr,
                           ColorTerms{});
    }
    static const int fast_env = [] { const char *e = getenv("KPX_NN_FAST"); return e ? atoi(e) : -1; }();     // A/B switch: 0 / 1 force a form
    // (with the operands in curve order the per-trip form wins in both cases: its prefilter rarely passes, and the chunked form's
    // bookkeeping is then pure overhead -- 0.545 vs 0.532 of the matrix peak warm, 0.458 vs 0.453 cold)
    const bool fast = fast_env >= 0 ? fast_env != 0 : (have_prev && !dense_sort_on());
    if (fast)
        hipLaunchKernelGGL(nn_mfma_kernel<true>, dim3(p.row_blocks, p.splits), thr, 0, st, n, b.B, b.colB, p.tiles_per_split, done, b.A64, b.K64,
                           b.init_val, b.init_idx, b.part_val, b.part_idx, rperm);
    else
        hipLaunchKernelGGL(nn_mfma_kernel<false>, dim3(p.row_blocks, p.splits), thr, 0, st, n, b.B, b.colB, p.tiles_per_split, done, b.A64, b.K64,
                           b.init_val, b.init_idx, b.part_val, b.part_idx, rperm);
    }
    hipLaunchKernelGGL(nn_merge_kernel, dim3((unsigned)cdiv(n, kMergeThreads)), dim3(kMergeThreads), 0, st, src, n, tgt, tn, T, done, b.part_val, b.part_idx,
                       p.splits, max_d2, mode, b.idx_cur, b.d2_cur, (double *)nullptr, b.part_acc, (const int32_t *)nullptr,
                       (const int32_t *)nullptr, ct);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

}  // namespace kpx

using namespace kpx;

KPX_EXPORT int kpx_prof_icp_waves(uint64_t *h_out, int64_t cap_waves, int64_t *h_count)
{
    KPX_REQUIRE(h_out && h_count && cap_waves >= 0, "kpx_prof_icp_waves: null pointer or negative capacity");
    *h_count = icp_wave_take((unsigned long long *)h_out, cap_waves);
    return *h_count < 0 ? KPX_ERR_HIP : KPX_OK;
}

KPX_EXPORT int kpx_prof_icp_cert(uint64_t *h_out8)
{
    KPX_REQUIRE(h_out8, "kpx_prof_icp_cert: null pointer");
    unsigned long long *p = nullptr;
    static const unsigned long long zero[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    KPX_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(g_cert_check)));
    KPX_HIP(hipMemcpy(h_out8, p, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    KPX_HIP(hipMemcpy(p, zero, sizeof(zero), hipMemcpyHostToDevice));
    return KPX_OK;
}
KPX_EXPORT int kpx_icp_chain(int32_t on)
{
    if (on == -2) return __atomic_load_n(&g_chain_launches, __ATOMIC_RELAXED);
    const int cur = chain_form_on() ? 1 : 0;
    if (on >= 0) g_chain_form = on ? 1 : 0;
    return cur;
}
// The chain clock (g_chain_stamp): 64 x 16 words, read and reset ([10], the earliest-block slot, to all ones)
KPX_EXPORT int kpx_prof_icp_chain(uint64_t *h_out3072)
{
    KPX_REQUIRE(h_out3072, "kpx_prof_icp_chain: null pointer");
    unsigned long long *p = nullptr;
    static unsigned long long init[64 * 48];
    for (int i = 0; i < 64 * 48; ++i) init[i] = (i % 48) == 10 ? ~0ull : 0ull;
    KPX_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(g_chain_stamp)));
    KPX_HIP(hipMemcpy(h_out3072, p, sizeof(init), hipMemcpyDeviceToHost));
    KPX_HIP(hipMemcpy(p, init, sizeof(init), hipMemcpyHostToDevice));
    return KPX_OK;
}
KPX_EXPORT int kpx_prof_icp_phases(double *h_out8)
{
    KPX_REQUIRE(h_out8, "kpx_prof_icp_phases: null pointer");
    return icp_phase_take(h_out8);
}
KPX_EXPORT int kpx_nn_engine(int32_t engine)
{
    const int cur = local_engine() ? KPX_NN_ENGINE_CULLED : (g_nn_fp64_only ? KPX_NN_ENGINE_DENSE_FP64 : KPX_NN_ENGINE_DENSE);
    if (engine < 0) return cur;
    KPX_REQUIRE(engine == KPX_NN_ENGINE_CULLED || engine == KPX_NN_ENGINE_DENSE || engine == KPX_NN_ENGINE_DENSE_FP64, "kpx_nn_engine: unknown engine %d", engine);
    g_nn_engine = engine == KPX_NN_ENGINE_CULLED ? KPX_NN_ENGINE_CULLED : KPX_NN_ENGINE_DENSE;
    g_nn_fp64_only = engine == KPX_NN_ENGINE_DENSE_FP64;
    return cur;
}

KPX_EXPORT size_t kpx_nn_workspace_bytes(int64_t n_src, int64_t n_tgt)
{
    Arena a(nullptr, 0);
    NnBuffers b;
    nn_carve(a, n_src, n_tgt, nn_plan(n_src, n_tgt), &b);
    return a.off;
}
KPX_EXPORT int kpx_nn_search(const float *src, int64_t n_src, const float *tgt, int64_t n_tgt, const double *d_T, int32_t *idx,
                             double *d2, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(n_src >= 0 && n_tgt >= 1, "kpx_nn_search: empty target");
    KPX_REQUIRE(n_src < ((int64_t)1 << 31) && n_tgt < ((int64_t)1 << 31) - 65536, "kpx_nn_search: cloud too large");
    if (n_src == 0) return KPX_OK;
    KPX_REQUIRE(src && tgt && d_T && idx && d2 && ws, "kpx_nn_search: null pointer");
    hipStream_t st = (hipStream_t)stream;
    Arena a(ws, ws_bytes);
    NnPlan p = nn_plan(n_src, n_tgt);
    NnBuffers b;
    nn_carve(a, n_src, n_tgt, p, &b);
    KPX_ARENA_CHECK(a);
    int rc = nn_prep(tgt, p, b, st);
    if (rc) return rc;
    rc = nn_prep_source(src, p, b, st);
    if (rc) return rc;
    rc = nn_search_launch(src, tgt, nullptr, p, b, d_T, nullptr, false, false, 0.0, -1, st);
    if (rc) return rc;
    KPX_HIP(hipMemcpyAsync(idx, b.idx_cur, (size_t)n_src * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    KPX_HIP(hipMemcpyAsync(d2, b.d2_cur, (size_t)n_src * sizeof(double), hipMemcpyDeviceToDevice, st));
    return KPX_OK;
}

KPX_EXPORT size_t kpx_kabsch_workspace_bytes(int64_t n_corr)
{
    Arena a(nullptr, 0);
    a.get<double>((size_t)256 * kAcc);
    return a.off;
}
KPX_EXPORT int kpx_kabsch(const float *src, const float *tgt, const int32_t *corr, int64_t n_corr, double *d_T, void *ws,
                          size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(n_corr >= 0 && d_T && ws, "kpx_kabsch: bad arguments");
    KPX_REQUIRE(n_corr == 0 || (src && tgt && corr), "kpx_kabsch: null pointer");
    hipStream_t st = (hipStream_t)stream;
    Arena a(ws, ws_bytes);
    double *part = a.get<double>((size_t)256 * kAcc);
    KPX_ARENA_CHECK(a);
    int nb = (int)(cdiv(n_corr > 0 ? n_corr : 1, 256) > 256 ? 256 : cdiv(n_corr > 0 ? n_corr : 1, 256));
    hipLaunchKernelGGL(pairs_acc_kernel, dim3(nb), dim3(256), 0, st, src, tgt, corr, n_corr, part);
    hipLaunchKernelGGL(pairs_solve_kernel, dim3(1), dim3(64), 0, st, part, nb, d_T);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

KPX_EXPORT size_t kpx_icp_workspace_bytes(int64_t n_src, int64_t n_tgt)
{
    return kpx_nn_workspace_bytes(n_src, n_tgt);
}
KPX_EXPORT int kpx_icp(const float *src, int64_t n_src, const float *tgt, const float *tgt_normals, int64_t n_tgt,
                       double max_dist, const double *h_init, int32_t mode, int32_t max_iteration, double relative_fitness,
                       double relative_rmse, int32_t poll_interval, double *d_result, int32_t *idx, double *d2, void *ws,
                       size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(mode == KPX_ICP_POINT_TO_POINT || mode == KPX_ICP_POINT_TO_PLANE, "kpx_icp: unknown estimation mode");
    KPX_REQUIRE(mode != KPX_ICP_POINT_TO_PLANE || tgt_normals,
                "TransformationEstimationPointToPlane and TransformationEstimationColoredICP require pre-computed normal vectors for target PointCloud.");
    KPX_REQUIRE(max_dist > 0.0, "Invalid max_correspondence_distance.");          // [O3D]
    KPX_REQUIRE(n_src >= 1 && n_tgt >= 1 && max_iteration >= 0, "kpx_icp: empty cloud");
    KPX_REQUIRE(n_src < ((int64_t)1 << 31) && n_tgt < ((int64_t)1 << 31) - 65536, "kpx_icp: cloud too large");
    KPX_REQUIRE(src && tgt && h_init && d_result && ws, "kpx_icp: null pointer");
    hipStream_t st = (hipStream_t)stream;
    Arena a(ws, ws_bytes);
    NnPlan p = nn_plan(n_src, n_tgt);
    NnBuffers b;
    nn_carve(a, n_src, n_tgt, p, &b);
    KPX_ARENA_CHECK(a);
    hipLaunchKernelGGL(icp_init_kernel, dim3(1), dim3(1), 0, st, b.state, mat16_from(h_init));
    int rc = nn_prep(tgt, p, b, st);
    if (rc) return rc;
    rc = nn_prep_source(src, p, b, st);
    if (rc) return rc;
    const double md2 = max_dist * max_dist;
    if (local_engine()) {
        // one launch per iteration; kernels queued behind a raised `done` return at once, so the flag is only read back
        // every poll_interval iterations (0 = never)
        for (int k = 0; k <= max_iteration; ++k) {
            icp_iter_launch(src, tgt, tgt_normals, p, b, md2, mode, k, max_iteration, relative_fitness, relative_rmse, d_result, st, nullptr, 0,
                            idx != nullptr || d2 != nullptr);
            if (poll_interval > 0 && (k + 1) % poll_interval == 0 && k < max_iteration) {
                int32_t h_done = 0;
                KPX_HIP(hipMemcpyAsync(&h_done, &b.state->done, sizeof(int32_t), hipMemcpyDeviceToHost, st));
                KPX_HIP(hipStreamSynchronize(st));
                if (h_done) break;
            }
        }
    } else {
        ScreenPolicy policy;
        for (int k = 0; k <= max_iteration; ++k) {
            rc = nn_search_launch(src, tgt, tgt_normals, p, b, b.state->T, &b.state->done, k > 0,
                                  poll_interval == 1 ? policy.allow(k) : k >= 2, md2, mode, st);
            if (rc) return rc;
            hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(kSolveThreads), 0, st, b.part_acc, (int)cdiv(n_src, kMergeThreads), n_src, mode, k,
                               max_iteration, relative_fitness, relative_rmse, b.state, d_result);
            if (poll_interval > 0 && (k + 1) % poll_interval == 0 && k < max_iteration) {
                IcpState h_state;
                KPX_HIP(hipMemcpyAsync(&h_state.fitness, &b.state->fitness, sizeof(IcpState) - offsetof(IcpState, fitness), hipMemcpyDeviceToHost, st));
                KPX_HIP(hipStreamSynchronize(st));
                if (h_state.done) break;
                policy.observe(h_state.fitness, h_state.rmse);
            }
        }
    }
    if (idx) KPX_HIP(hipMemcpyAsync(idx, b.idx_cur, (size_t)n_src * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (d2) KPX_HIP(hipMemcpyAsync(d2, b.d2_cur, (size_t)n_src * sizeof(double), hipMemcpyDeviceToDevice, st));
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

// ---- coloured ICP (preprocessing/registration.py:89-114) -----------------------------------------------------
// [O3D] registration_colored_icp = the registration_icp loop (same correspondences, fitness, inlier rmse and convergence
// test) with TransformationEstimationForColoredICP as the update; the target's colour gradient comes from
// kpx_color_gradient.  Search -> sums -> solve are three launches per iteration (nn_merge_kernel mode 2 + icp_solve_kernel):
// the function is unused in the reference, so this path is built for parity, not for speed.
KPX_EXPORT size_t kpx_colored_icp_workspace_bytes(int64_t n_src, int64_t n_tgt) { return kpx_icp_workspace_bytes(n_src, n_tgt); }
KPX_EXPORT int kpx_colored_icp(const float *src, const float *src_colors, int64_t n_src, const float *tgt, const float *tgt_colors,
                               const float *tgt_normals, const double *tgt_gradient, int64_t n_tgt, double max_dist, const double *h_init,
                               double lambda_geometric, int32_t max_iteration, double relative_fitness, double relative_rmse,
                               int32_t poll_interval, double *d_result, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(tgt_normals, "TransformationEstimationPointToPlane and TransformationEstimationColoredICP require pre-computed normal vectors for target PointCloud.");
    KPX_REQUIRE(src_colors && tgt_colors && tgt_gradient, "kpx_colored_icp: colours of both clouds and the target's colour gradient are required");
    KPX_REQUIRE(max_dist > 0.0, "Invalid max_correspondence_distance.");
    KPX_REQUIRE(lambda_geometric >= 0.0 && lambda_geometric <= 1.0, "kpx_colored_icp: lambda_geometric must lie in [0, 1]");
    KPX_REQUIRE(n_src >= 1 && n_tgt >= 1 && max_iteration >= 0, "kpx_colored_icp: empty cloud");
    KPX_REQUIRE(n_src < ((int64_t)1 << 31) && n_tgt < ((int64_t)1 << 31) - 65536, "kpx_colored_icp: cloud too large");
    KPX_REQUIRE(src && tgt && h_init && d_result && ws, "kpx_colored_icp: null pointer");
    hipStream_t st = (hipStream_t)stream;
    Arena a(ws, ws_bytes);
    NnPlan p = nn_plan(n_src, n_tgt);
    NnBuffers b;
    nn_carve(a, n_src, n_tgt, p, &b);
    KPX_ARENA_CHECK(a);
    hipLaunchKernelGGL(icp_init_kernel, dim3(1), dim3(1), 0, st, b.state, mat16_from(h_init));
    int rc = nn_prep(tgt, p, b, st);
    if (rc) return rc;
    rc = nn_prep_source(src, p, b, st);
    if (rc) return rc;
    const ColorTerms ct{ src_colors, tgt_colors, tgt_gradient, sqrt(lambda_geometric), sqrt(1.0 - lambda_geometric) };
    const double md2 = max_dist * max_dist;
    for (int k = 0; k <= max_iteration; ++k) {
        rc = nn_search_launch(src, tgt, tgt_normals, p, b, b.state->T, &b.state->done, k > 0, false, md2, 2, st, ct);
        if (rc) return rc;
        hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(kSolveThreads), 0, st, b.part_acc, (int)cdiv(n_src, kMergeThreads), n_src, 1, k,
                           max_iteration, relative_fitness, relative_rmse, b.state, d_result);
        if (poll_interval > 0 && (k + 1) % poll_interval == 0 && k < max_iteration) {
            int32_t h_done = 0;
            KPX_HIP(hipMemcpyAsync(&h_done, &b.state->done, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            KPX_HIP(hipStreamSynchronize(st));
            if (h_done) break;
        }
    }
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

// ---- several registrations onto one shared target -----------------------------------------------------------------
// (preprocessing/data.py:144-161 registers every sub device onto the same master cloud.)  The target operand is
// prepared once on the caller's stream; the problems then run side by side on internal lanes (their kernels are
// short and latency-bound: one problem alone leaves most of the chip idle), each with two chunks of iterations in
// flight and its state copied to pinned memory after every chunk, so neither the lanes nor the host wait for a round
// trip.  The caller's stream continues after all lanes have finished.
KPX_EXPORT size_t kpx_icp_batch_workspace_bytes(int32_t count, const int64_t *h_n_src, int64_t n_tgt)
{
    Arena a(nullptr, 0);
    NnBuffers b;
    if (count < 1 || !h_n_src) return 0;
    NnPlan tplan = nn_plan(h_n_src[0], n_tgt);          // the shared operand is padded for the largest split plan
    for (int i = 1; i < count; ++i) {
        NnPlan q = nn_plan(h_n_src[i], n_tgt);
        if (q.tiles_pad > tplan.tiles_pad) tplan.tiles_pad = q.tiles_pad;
        if (q.f_tiles_pad > tplan.f_tiles_pad) tplan.f_tiles_pad = q.f_tiles_pad;
    }
    nn_carve_target(a, tplan, &b);
    int64_t total = n_tgt;
    for (int i = 0; i < count; ++i) { nn_carve_source(a, h_n_src[i], nn_plan(h_n_src[i], n_tgt), &b); total += h_n_src[i]; }
    if (count + 1 <= kMortonBatchMax && total < ((int64_t)1 << 31)) {
        MortonBatchScratch ms;
        morton_batch_carve(a, total, &ms);
    }
    return a.off;
}
KPX_EXPORT int kpx_icp_batch(int32_t count, const float *const *h_src, const int64_t *h_n_src, const float *tgt,
                             const float *tgt_normals, int64_t n_tgt, double max_dist, const double *h_init, int32_t mode,
                             int32_t max_iteration, double relative_fitness, double relative_rmse, double *d_results, void *ws,
                             size_t ws_bytes, void *stream)
{
    return kpx::icp_batch_ordered(count, h_src, h_n_src, tgt, tgt_normals, n_tgt, max_dist, h_init, mode, max_iteration, relative_fitness, relative_rmse,
                                  d_results, ws, ws_bytes, stream, false);
}
// ---- the one-launch chain: who may be resident ---------------------------------------------------------------------------
// icp_chain_kernel needs all its blocks resident together, so the host admits a chain only while the blocks of every chain kernel it
// has in flight on the device (this process: the deployment is one process per GPU) plus the new ones fit a budget below what the
// device holds (blocks per CU by the occupancy API minus one -- MI355X_MICROARCH.md, "Correctness boundaries": the API can be one
// block per CU high -- times the CUs).  A chain that does not fit runs in the launch-per-iteration form: nothing ever waits for
// another chain.  Finished chains are retired by querying the event recorded behind them.
namespace {
struct ChainSlot {
    hipEvent_t ev;
    unsigned blocks;
    bool busy, made;
};
struct ChainBook {
    std::mutex mu;
    ChainSlot slot[16] = {};
    long budget = -1;                                    // blocks; -1 = not asked yet
};
ChainBook g_chain_book[16];
unsigned long long *g_chain_abort = nullptr;             // pinned: raised by a chain block that gave up waiting
std::once_flag g_chain_abort_once;
unsigned long long *chain_abort_word()
{
    std::call_once(g_chain_abort_once, [] {
        if (hipHostMalloc((void **)&g_chain_abort, 64, hipHostMallocDefault) != hipSuccess) g_chain_abort = nullptr;
        else *g_chain_abort = 0ull;
    });
    return g_chain_abort;
}
template <class F> bool chain_launch_if_fits(unsigned blocks, hipStream_t st, F &&launch)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
    ChainBook &bk = g_chain_book[dev];
    std::lock_guard<std::mutex> lock(bk.mu);
    if (bk.budget < 0) {
        int per_cu = 0, cus = 0;
        static const long forced = [] { const char *e = getenv("KPX_ICP_CHAIN_BUDGET"); return e ? atol(e) : -1L; }();
        // ONE process per GPU may run chains: two processes each admitting chains against the whole device could leave blocks of both
        // waiting for wave slots the other holds (ended only by the timeout).  The first process to take the device's lock file keeps
        // it for its lifetime; the others (ranks sharing a GPU in a rehearsal, a second service on the same card) run a launch per
        // iteration.  No lock (no writable /tmp, no bus id) = no chains.
        static const bool no_lock = [] { const char *e = getenv("KPX_ICP_CHAIN_LOCK"); return e && e[0] == '0'; }();   // the caller vouches for being alone
        bool mine = no_lock;
        char bus[64] = { 0 };
        if (!mine && hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev) == hipSuccess) {
            char path[128];
            for (char *c = bus; *c; ++c) if (!((*c >= '0' && *c <= '9') || (*c >= 'a' && *c <= 'f') || (*c >= 'A' && *c <= 'F'))) *c = '_';
            // (per user, never through a planted symlink, not inherited across exec; containers that share a GPU but not /dev/shm each
            // believe they are alone: such deployments set KPX_ICP_CHAIN=0 -- INTEGRATION.md)
            snprintf(path, sizeof(path), "/dev/shm/kpx_chain_%u_%s.lock", (unsigned)getuid(), bus);
            const int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
            if (fd >= 0) {
                if (flock(fd, LOCK_EX | LOCK_NB) == 0) mine = true;        // (kept open: the lock lives as long as the process)
                else close(fd);
            }
        }
        if (!mine) bk.budget = 0;
        else if (forced >= 0) bk.budget = forced;
        else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)icp_chain_kernel, kIThreads, 0) == hipSuccess &&
                 hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && per_cu > 1)
            bk.budget = (long)(per_cu - 1) * cus;
        else bk.budget = 0;
    }
    long used = 0;
    int free_slot = -1;
    for (int i = 0; i < 16; ++i) {
        ChainSlot &c = bk.slot[i];
        if (c.busy && hipEventQuery(c.ev) == hipSuccess) c.busy = false;
        if (c.busy) used += c.blocks;
        else if (free_slot < 0) free_slot = i;
    }
    if (free_slot < 0 || used + (long)blocks > bk.budget) return false;
    ChainSlot &c = bk.slot[free_slot];
    if (!c.made) {
        if (hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) != hipSuccess) return false;
        c.made = true;
    }
    launch();
    __atomic_fetch_add(&g_chain_launches, 1, __ATOMIC_RELAXED);
    if (hipEventRecord(c.ev, st) != hipSuccess) return true;     // launched all the same; the slot just is not booked
    c.blocks = blocks;
    c.busy = true;
    return true;
}
}  // namespace
static int g_busy_threads = 0;
static thread_local int t_busy_depth = 0;
kpx::BusyScope::BusyScope() : counted(t_busy_depth++ == 0) { if (counted) __atomic_fetch_add(&g_busy_threads, 1, __ATOMIC_RELAXED); }
kpx::BusyScope::~BusyScope() { --t_busy_depth; if (counted) __atomic_fetch_sub(&g_busy_threads, 1, __ATOMIC_RELAXED); }
int kpx::busy_threads() { return __atomic_load_n(&g_busy_threads, __ATOMIC_RELAXED); }
// A chain kernel gave up waiting since the last call (see icp_chain_kernel): 1 once, then cleared
int kpx::icp_chain_abort_take()
{
    unsigned long long *w = chain_abort_word();
    if (!w) return 0;
    const unsigned long long v = __atomic_exchange_n(w, 0ull, __ATOMIC_ACQ_REL);
    return v ? 1 : 0;
}

namespace kpx {

// icp_rows_kernel with R rows per wave (KPX_ICP_ROWS_R: 64 / 32 / 16): problems given as IcpProblems, each at its own iteration k[c]
static int rows_per_wave()
{
    static const int r = [] { const char *e = getenv("KPX_ICP_ROWS_R"); const int v = e ? atoi(e) : 0; return v == 16 || v == 32 || v == 64 ? v : 64; }();
    return r;
}
static void rows_launch(const IcpProblem *const *probs, const int *ks, int cnt, double md2, int mode, int max_iter, double rel_fit, double rel_rmse,
                        unsigned long long *visits, int light, CertPolicy pol, hipStream_t st)
{
    const int R = rows_per_wave();
    RowsArgs ra;
    ra.count = cnt;
    unsigned b0 = 0;
    for (int c = 0; c < kRowsBatchMax; ++c) {
        const int cc = c < cnt ? c : cnt - 1;
        const IcpProblem &Q = *probs[cc];
        RowsProblem &Rp = ra.p[c];
        const int64_t n64 = cdiv(Q.n, 64), n32 = cdiv(Q.n, 32);
        Rp.src_sorted = Q.src_sorted; Rp.idx_sorted = Q.idx_sorted; Rp.ptgt_sorted = Q.ptgt_sorted; Rp.cert = Q.cert; Rp.thist = Q.thist;
        Rp.light_key = Q.light_key + (R == 64 ? 0 : R == 32 ? n64 : n64 + n32); Rp.sbbox = Q.sbbox; Rp.state = Q.pair; Rp.ring = Q.ring; Rp.result = Q.result;
        Rp.progress = Q.progress; Rp.tag = Q.tag; Rp.tgt = Q.tgt; Rp.tn = Q.tn; Rp.Bs = Q.Bs; Rp.orig = Q.orig; Rp.tile_box = Q.tile_box; Rp.group_box = Q.group_box;
        Rp.tbbox = Q.tbbox; Rp.n = Q.n; Rp.n_groups = Q.n_groups; Rp.k = ks[cc];
        Rp.block0 = b0; Rp.blocks = (unsigned)cdiv(Q.n, R);
        if (c < cnt) b0 += Rp.blocks;
    }
#define KPX_ROWS_LAUNCH(M, RR) hipLaunchKernelGGL((icp_rows_kernel<M, RR>), dim3(b0), dim3(64), 0, st, ra, md2, max_iter, rel_fit, rel_rmse, visits, light, pol)
    if (mode == 1) { if (R == 64) KPX_ROWS_LAUNCH(1, 64); else if (R == 32) KPX_ROWS_LAUNCH(1, 32); else KPX_ROWS_LAUNCH(1, 16); }
    else { if (R == 64) KPX_ROWS_LAUNCH(0, 64); else if (R == 32) KPX_ROWS_LAUNCH(0, 32); else KPX_ROWS_LAUNCH(0, 16); }
#undef KPX_ROWS_LAUNCH
}

// ---- IcpEngine: the registrations of ALL frames in flight in one chain of launches (round 5) ------------------------------------
// With several frames in flight every frame used to drive its own chain of ~31 launches from its own host thread: four chains, four
// poll loops, ~120 launches per four frames competing for the command processor.  A kpx_stream attaches its worker threads to the
// device's engine instead: icp_batch_ordered prepares its registrations as before (curve order, boxes, operands, the init kernel: on
// the frame's own stream), records an event, and hands the group over; the engine thread -- ONE host thread, ONE stream -- carries the
// registrations of every frame that is in its registration phase in one launch per tick, each at the iteration it has reached
// (IcpProblem::k), the calm ones in icp_rows_kernel (<= 16 per launch), those still searching most rows in icp_iter_batch_kernel
// (<= 8 per launch).  A group joins at the first tick after its preparation has finished on the device (its event is polled: the
// engine's stream never waits for a frame that is still preparing), leaves when its last registration has converged or run out of
// iterations, and the frame's stream waits for the event recorded behind the group's last launch.  Results do not depend on who
// launches what when: partners are exact, the sums order-free (test_native_frame_stream_equals_serial_steps_and_oracle runs with and
// without the engine).  MEASURED (profiles/r05/exp_icp_engine_one_chain.txt, exp_icp_engine_two_chains.txt): 1550 Mpoints/s with one chain,
// 1720 with two (below), against 2740-2800 with a chain of launches per frame on the same box -- lockstep ticks advance every frame at the pace
// of the slowest launch, and a launch's tail (it lasts as long as its slowest block) is filled by nobody, where four independent chains fill
// each other's.  So the engine is OFF by default (KPX_STREAM_ENGINE=1 switches it on); results are identical either way.
struct EngJob;
struct EngProblem {
    IcpProblem P;
    unsigned long long gen;
    int next_k, seen, share;
    int lane;                              // the chain (0 / 1) its last launch ran in; -1: none yet
    bool done;
    EngJob *job;
};
constexpr int kEngEvents = 32;
struct EngJob {
    int count = 0;
    EngProblem p[kIcpBatchMax];
    double md2 = 0.0, rel_fit = 0.0, rel_rmse = 0.0;
    int mode = 0, max_iter = 0, light = 0, rows_mode = 0, rows_share = 0, window = 3;
    CertPolicy pol = { 0.0f, 0.0f, 0.0f, 0.0f };
    hipEvent_t ready = nullptr, done_ev[2] = { nullptr, nullptr };
    int used = 0;                          // bit f: the group ran launches in chain f (done_ev[f] is recorded when it leaves)
    std::atomic<int> state{ 0 };          // 0 handed over, 1 active, 2 finished (done events recorded, rc set)
    int rc = KPX_OK;
    bool same_launch(const EngJob &o) const
    {
        return md2 == o.md2 && rel_fit == o.rel_fit && rel_rmse == o.rel_rmse && mode == o.mode && max_iter == o.max_iter && light == o.light &&
               pol.calm == o.pol.calm && pol.factor == o.pol.factor && pol.smin == o.pol.smin && pol.smax == o.pol.smax;
    }
};
struct IcpEngine {
    int dev = 0, refs = 0;
    hipStream_t st[2] = { nullptr, nullptr };
    hipEvent_t mig[kEngEvents] = { nullptr };
    unsigned mig_next = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<EngJob *> inbox;
    std::atomic<int> inbox_n{ 0 };
    std::atomic<bool> quit{ false };
    std::atomic<unsigned long long> launches{ 0 }, ticks{ 0 };
};
static std::mutex g_engine_mu;
static IcpEngine *g_engine[16] = { nullptr };
static thread_local IcpEngine *t_engine = nullptr;

static void engine_fail(std::vector<EngJob *> &jobs, int rc)
{
    for (EngJob *j : jobs) { j->rc = rc; j->state.store(2, std::memory_order_release); }
    jobs.clear();
}
static void engine_run(IcpEngine *E)
{
    (void)hipSetDevice(E->dev);
    static const double stall_limit = [] { const char *e = getenv("KPX_ICP_STALL_SECONDS"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 60.0; }();
    std::vector<EngJob *> pending, active;
    auto t_last = std::chrono::steady_clock::now();
    for (;;) {
        if (E->inbox_n.load(std::memory_order_acquire) > 0) {
            std::lock_guard<std::mutex> lock(E->mu);
            for (EngJob *j : E->inbox) pending.push_back(j);
            E->inbox.clear();
            E->inbox_n.store(0, std::memory_order_release);
        }
        if (E->quit.load(std::memory_order_acquire)) {
            engine_fail(pending, KPX_ERR_HIP);
            engine_fail(active, KPX_ERR_HIP);
            return;
        }
        if (pending.empty() && active.empty()) {             // idle: a short spin, then sleep until a group arrives
            const auto t0 = std::chrono::steady_clock::now();
            bool woke = false;
            for (unsigned spins = 0; !woke; ++spins) {
                if (E->inbox_n.load(std::memory_order_acquire) > 0 || E->quit.load(std::memory_order_acquire)) woke = true;
                else if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) break;
                else __builtin_ia32_pause();
            }
            if (!woke) {
                std::unique_lock<std::mutex> lock(E->mu);
                E->cv.wait(lock, [&] { return E->inbox_n.load(std::memory_order_acquire) > 0 || E->quit.load(std::memory_order_acquire); });
            }
            t_last = std::chrono::steady_clock::now();
            continue;
        }
        bool advanced = false;
        // groups whose preparation has finished on the device join (the engine's stream never waits for one that has not)
        for (size_t i = 0; i < pending.size();) {
            const hipError_t q = hipEventQuery(pending[i]->ready);
            if (q == hipErrorNotReady) { ++i; continue; }
            EngJob *j = pending[i];
            pending.erase(pending.begin() + (long)i);
            if (q != hipSuccess || hipStreamWaitEvent(E->st[0], j->ready, 0) != hipSuccess || hipStreamWaitEvent(E->st[1], j->ready, 0) != hipSuccess) {
                j->rc = fail(KPX_ERR_HIP, "kpx_icp_batch (engine): the group's preparation failed: %s", hipGetErrorString(q));
                j->state.store(2, std::memory_order_release);
                continue;
            }
            j->state.store(1, std::memory_order_release);
            active.push_back(j);
            advanced = true;
        }
        // progress words -> what every registration has finished, whether it converged, how much of it is still searched
        for (EngJob *j : active)
            for (int c = 0; c < j->count; ++c) {
                EngProblem &q = j->p[c];
                if (q.done) continue;
                const unsigned long long w = __atomic_load_n(q.P.progress, __ATOMIC_ACQUIRE);
                if ((w >> 40) != q.gen) continue;
                q.seen = (int)(w & 0xFFFFFFFFull);
                q.share = q.seen >= 1 ? (int)((w >> 33) & 127ull) : 127;
                if ((w >> 32) & 1ull) q.done = true;
            }
        // Ticks.  Two chains of launches, each on its own stream: [0] the registrations that still search most of their rows
        // (icp_iter_batch_kernel: long, slot-bound launches), [1] the calm ones (icp_rows_kernel: short, latency-bound) -- in ONE chain the
        // calm registrations of three frames advanced at the pace of the fourth frame's first sweeps (measured: 1550 vs 2800 Mpoints/s,
        // profiles/r05/exp_icp_engine_one_chain.txt).  A chain ticks when every registration it would carry has room in its window; a
        // registration that changes chains takes an event along (its next launch waits for its previous one).
        std::vector<char> ticked(active.size(), 0);
        for (size_t a = 0; a < active.size(); ++a) {
            if (ticked[a]) continue;
            EngJob *lead = active[a];
            const int last_k = lead->max_iter;                      // (update in the last block: no closing launch)
            std::vector<EngProblem *> form[2];
            bool room[2] = { true, true };
            for (size_t b = a; b < active.size(); ++b) {
                if (ticked[b] || !active[b]->same_launch(*lead)) continue;
                ticked[b] = 1;
                for (int c = 0; c < active[b]->count; ++c) {
                    EngProblem &q = active[b]->p[c];
                    if (q.done || q.next_k > last_k) continue;
                    const int f = (lead->rows_mode == 2 || (lead->rows_mode == 1 && q.share <= lead->rows_share)) ? 1 : 0;
                    if (q.next_k - q.seen >= lead->window) room[f] = false;
                    form[f].push_back(&q);
                    q.job = active[b];
                }
            }
            unsigned long long *visits = prof_armed() ? nn_visits_ptr() : (unsigned long long *)nullptr;
            for (int f = 0; f < 2; ++f) {
                if (form[f].empty() || !room[f]) continue;
                advanced = true;
                E->ticks.fetch_add(1, std::memory_order_relaxed);
                hipStream_t fs = E->st[f];
                for (EngProblem *q : form[f]) {
                    if (q->lane >= 0 && q->lane != f) {                 // its previous launch ran in the other chain
                        hipEvent_t ev = E->mig[E->mig_next++ % kEngEvents];
                        (void)hipEventRecord(ev, E->st[q->lane]);
                        (void)hipStreamWaitEvent(fs, ev, 0);
                    }
                    q->lane = f;
                    q->job->used |= 1 << f;
                }
                if (f == 1) {
                    for (size_t i0 = 0; i0 < form[1].size(); i0 += kRowsBatchMax) {
                        const int cnt = (int)(form[1].size() - i0 < (size_t)kRowsBatchMax ? form[1].size() - i0 : (size_t)kRowsBatchMax);
                        const IcpProblem *pp[kRowsBatchMax];
                        int ks[kRowsBatchMax];
                        for (int c = 0; c < cnt; ++c) { pp[c] = &form[1][i0 + (size_t)c]->P; ks[c] = form[1][i0 + (size_t)c]->next_k; }
                        rows_launch(pp, ks, cnt, lead->md2, lead->mode, lead->max_iter, lead->rel_fit, lead->rel_rmse, visits, lead->light, lead->pol, fs);
                        E->launches.fetch_add(1, std::memory_order_relaxed);
                    }
                } else {
                    for (size_t i0 = 0; i0 < form[0].size(); i0 += kIcpBatchMax) {
                        IcpBatchArgs ba;
                        const int cnt = (int)(form[0].size() - i0 < (size_t)kIcpBatchMax ? form[0].size() - i0 : (size_t)kIcpBatchMax);
                        ba.count = cnt;
                        unsigned b0 = 0;
                        for (int c = 0; c < kIcpBatchMax; ++c) {
                            const EngProblem *q = form[0][i0 + (size_t)(c < cnt ? c : cnt - 1)];
                            ba.p[c] = q->P;
                            ba.p[c].k = q->next_k;
                            ba.p[c].block0 = b0;
                            if (c < cnt) b0 += q->P.blocks;
                        }
                        hipLaunchKernelGGL(icp_iter_batch_kernel, dim3(b0), dim3(kIThreads), 0, fs, ba, lead->md2, lead->mode, lead->max_iter, lead->rel_fit, lead->rel_rmse, visits,
                                           2, lead->light, lead->pol);
                        E->launches.fetch_add(1, std::memory_order_relaxed);
                    }
                }
                for (EngProblem *q : form[f]) ++q->next_k;
            }
            if (hipGetLastError() != hipSuccess) {
                const int rc = fail(KPX_ERR_HIP, "kpx_icp_batch (engine): launch failed");
                engine_fail(active, rc);
                break;
            }
        }
        // groups whose registrations have all converged or run out of iterations leave: the frame waits for the events behind their last
        // launches (one per chain the group used)
        for (size_t i = 0; i < active.size();) {
            EngJob *j = active[i];
            bool fin = true;
            for (int c = 0; c < j->count; ++c) fin = fin && (j->p[c].done || j->p[c].next_k > j->max_iter);
            if (!fin) { ++i; continue; }
            for (int f = 0; f < 2; ++f)
                if (((j->used >> f) & 1) && hipEventRecord(j->done_ev[f], E->st[f]) != hipSuccess) j->rc = fail(KPX_ERR_HIP, "kpx_icp_batch (engine): hipEventRecord failed");
            j->state.store(2, std::memory_order_release);
            active.erase(active.begin() + (long)i);
            advanced = true;
        }
        if (advanced) t_last = std::chrono::steady_clock::now();
        else {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_last).count() > stall_limit) {
                const int rc = fail(KPX_ERR_HIP, "kpx_icp_batch (engine): no progress for %.0f s (KPX_ICP_STALL_SECONDS)", stall_limit);
                engine_fail(active, rc);
                engine_fail(pending, rc);
                t_last = std::chrono::steady_clock::now();
            }
            __builtin_ia32_pause();
        }
    }
}

IcpEngine *icp_engine_acquire()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lock(g_engine_mu);
    IcpEngine *E = g_engine[dev];
    if (!E) {
        E = new IcpEngine();
        E->dev = dev;
        bool ok = hipStreamCreateWithFlags(&E->st[0], hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&E->st[1], hipStreamNonBlocking) == hipSuccess;
        for (int e = 0; e < kEngEvents && ok; ++e) ok = hipEventCreateWithFlags(&E->mig[e], hipEventDisableTiming) == hipSuccess;
        if (!ok) { delete E; return nullptr; }
        E->th = std::thread(engine_run, E);
        g_engine[dev] = E;
    }
    ++E->refs;
    return E;
}
void icp_engine_release(IcpEngine *E)
{
    if (!E) return;
    {
        std::lock_guard<std::mutex> lock(g_engine_mu);
        if (--E->refs > 0) return;
        g_engine[E->dev] = nullptr;
    }
    {
        std::lock_guard<std::mutex> lock(E->mu);
        E->quit.store(true, std::memory_order_release);
    }
    E->cv.notify_all();
    if (E->th.joinable()) E->th.join();
    for (int f = 0; f < 2; ++f) { (void)hipStreamSynchronize(E->st[f]); (void)hipStreamDestroy(E->st[f]); }
    for (int e = 0; e < kEngEvents; ++e) if (E->mig[e]) (void)hipEventDestroy(E->mig[e]);
    delete E;
}
void icp_engine_attach(IcpEngine *E) { t_engine = E; }
void icp_engine_counters(IcpEngine *E, unsigned long long *launches, unsigned long long *ticks)
{
    if (launches) *launches = E ? E->launches.load() : 0ull;
    if (ticks) *ticks = E ? E->ticks.load() : 0ull;
}
// hands one prepared group over and waits (host) until its last launch has been queued; `ls` then waits for that launch on the device
static int engine_run_group(IcpEngine *E, EngJob &job, hipStream_t ls)
{
    static thread_local hipEvent_t ev[kIcpBatchMax][3] = { { nullptr } };
    static thread_local int ev_next = 0;
    const int slot = ev_next++ % kIcpBatchMax;
    for (int e = 0; e < 3; ++e)
        if (!ev[slot][e]) KPX_HIP(hipEventCreateWithFlags(&ev[slot][e], hipEventDisableTiming));
    job.ready = ev[slot][0];
    job.done_ev[0] = ev[slot][1];
    job.done_ev[1] = ev[slot][2];
    KPX_HIP(hipEventRecord(job.ready, ls));
    {
        std::lock_guard<std::mutex> lock(E->mu);
        E->inbox.push_back(&job);
        E->inbox_n.fetch_add(1, std::memory_order_release);
    }
    E->cv.notify_all();
    while (job.state.load(std::memory_order_acquire) != 2) __builtin_ia32_pause();
    if (job.rc) return job.rc;
    for (int f = 0; f < 2; ++f)
        if ((job.used >> f) & 1) KPX_HIP(hipStreamWaitEvent(ls, job.done_ev[f], 0));
    return KPX_OK;
}
}  // namespace kpx

int kpx::icp_batch_ordered(int32_t count, const float *const *h_src, const int64_t *h_n_src, const float *tgt, const float *tgt_normals, int64_t n_tgt,
                           double max_dist, const double *h_init, int32_t mode, int32_t max_iteration, double relative_fitness, double relative_rmse,
                           double *d_results, void *ws, size_t ws_bytes, void *stream, bool presorted)
{
    KPX_REQUIRE(count >= 1 && count <= 64 && h_src && h_n_src, "kpx_icp_batch: bad batch");
    KPX_REQUIRE(mode == KPX_ICP_POINT_TO_POINT || mode == KPX_ICP_POINT_TO_PLANE, "kpx_icp: unknown estimation mode");
    KPX_REQUIRE(mode != KPX_ICP_POINT_TO_PLANE || tgt_normals,
                "TransformationEstimationPointToPlane and TransformationEstimationColoredICP require pre-computed normal vectors for target PointCloud.");
    KPX_REQUIRE(max_dist > 0.0, "Invalid max_correspondence_distance.");
    KPX_REQUIRE(n_tgt >= 1 && n_tgt < ((int64_t)1 << 31) - 65536 && max_iteration >= 0, "kpx_icp_batch: bad target size");
    KPX_REQUIRE(tgt && h_init && d_results && ws, "kpx_icp_batch: null pointer");
    for (int i = 0; i < count; ++i)
        KPX_REQUIRE(h_src[i] && h_n_src[i] >= 1 && h_n_src[i] < ((int64_t)1 << 31), "kpx_icp_batch: bad source cloud %d", i);
    hipStream_t st = (hipStream_t)stream;
    BusyScope busy;
    // (a one-launch chain that gave up its residency wait poisons ITS OWN results with NaN -- icp_chain_kernel -- and the callers that read
    // results report it for that call; an abort left over from an earlier call is not this call's error)
    (void)icp_chain_abort_take();
    // the problems of a batch are independent chains of short, latency-bound kernels: they run side by side on the
    // library's internal lanes (kpx_internal.h), forked from / joined to the caller's stream by events
    LaneSet *ln = nullptr;
    int lrc = lanes_get(&ln);
    if (lrc) return lrc;
    hipStream_t *lanes = ln->s;
    constexpr int kBatchLanes = kLaneCount;
    static thread_local IcpState *h_states = nullptr;   // pinned: two poll slots per problem (per calling thread)
    if (!h_states) KPX_HIP(hipHostMalloc((void **)&h_states, 64 * 2 * sizeof(IcpState), hipHostMallocDefault));
    Arena a(ws, ws_bytes);
    NnPlan plans[64];
    NnBuffers bufs[64] = {};
    NnPlan tplan = nn_plan(h_n_src[0], n_tgt);          // the shared operand is padded for the largest split plan
    for (int i = 0; i < count; ++i) {
        plans[i] = nn_plan(h_n_src[i], n_tgt);
        if (plans[i].tiles_pad > tplan.tiles_pad) tplan.tiles_pad = plans[i].tiles_pad;
        if (plans[i].f_tiles_pad > tplan.f_tiles_pad) tplan.f_tiles_pad = plans[i].f_tiles_pad;
    }
    nn_carve_target(a, tplan, &bufs[0]);
    for (int i = 0; i < count; ++i) {
        bufs[i].B = bufs[0].B; bufs[i].Bseed = bufs[0].Bseed; bufs[i].colB = bufs[0].colB; bufs[i].colSeed = bufs[0].colSeed; bufs[i].Bf = bufs[0].Bf; bufs[i].aux = bufs[0].aux;
        bufs[i].tbbox = bufs[0].tbbox;
        bufs[i].Bs = bufs[0].Bs; bufs[i].orig_t = bufs[0].orig_t; bufs[i].tile_box = bufs[0].tile_box; bufs[i].group_box = bufs[0].group_box;
        bufs[i].sort_t = bufs[0].sort_t;
        nn_carve_source(a, h_n_src[i], plans[i], &bufs[i]);
    }
    int64_t total_pts = n_tgt;
    for (int i = 0; i < count; ++i) total_pts += h_n_src[i];
    const bool can_batch_sort = count + 1 <= kMortonBatchMax && total_pts < ((int64_t)1 << 31);
    MortonBatchScratch ms;
    if (can_batch_sort) morton_batch_carve(a, total_pts, &ms);
    KPX_ARENA_CHECK(a);
    // the shared target and every source are ordered along their Morton curves by ONE sort (cloud number above the code)
    const bool ordered = can_batch_sort && local_engine();
    int rc = KPX_OK;
    if (ordered) {
        MortonBatch mb;
        mb.count = count + 1;
        mb.off[0] = 0;
        for (int c = 0; c < kMortonBatchMax; ++c) {
            const bool on = c < mb.count;
            mb.pts[c] = !on ? nullptr : (c == 0 ? tgt : h_src[c - 1]);
            mb.perm[c] = !on ? nullptr : (c == 0 ? bufs[0].orig_t : bufs[c - 1].row_of);
            mb.bbox[c] = !on ? nullptr : (c == 0 ? bufs[0].sort_t.bbox : bufs[c - 1].sort_s.bbox);
            mb.off[c + 1] = mb.off[c] + (!on ? 0 : (c == 0 ? n_tgt : h_n_src[c - 1]));
        }
        rc = morton_order_batch(mb, ms, st, presorted);
        if (rc) return rc;
    }
    rc = nn_prep(tgt, tplan, bufs[0], st, ordered);
    if (rc) return rc;
    hipEvent_t ev[64][2];
    const bool use_events = !local_engine();               // the all-pairs engine polls copies of the state through events
    if (use_events)
        for (int i = 0; i < count; ++i)
            for (int e = 0; e < 2; ++e) KPX_HIP(hipEventCreateWithFlags(&ev[i][e], hipEventDisableTiming));
    const double md2 = max_dist * max_dist;
    ScreenPolicy policy[64];
    // A chunk = `chunk` iterations of one problem followed by a copy of its state to a pinned slot and an event.  The
    // culled engine keeps two chunks per problem in flight (an iteration is two short kernels; kernels behind a raised
    // flag return at once), so a lane never waits for the host's round trip; the all-pairs engine polls every iteration
    // (its sweep choice follows the polled state).
    const int chunk = local_engine() ? 4 : 1, in_flight = local_engine() ? 2 : 1;
    int next_k[64], enq[64], polled[64];
    auto enqueue_chunk = [&](int i) -> int {
        hipStream_t ls = lanes[i % kBatchLanes];
        for (int c = 0; c < chunk && next_k[i] <= max_iteration; ++c, ++next_k[i]) {
            const int k = next_k[i];
            if (local_engine()) {
                icp_iter_launch(h_src[i], tgt, tgt_normals, plans[i], bufs[i], md2, mode, k, max_iteration, relative_fitness, relative_rmse,
                                d_results + 20 * i, ls);
            } else {
                int r = nn_search_launch(h_src[i], tgt, tgt_normals, plans[i], bufs[i], bufs[i].state->T, &bufs[i].state->done, k > 0,
                                         policy[i].allow(k), md2, mode, ls);
                if (r) return r;
                hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(kSolveThreads), 0, ls, bufs[i].part_acc, (int)cdiv(h_n_src[i], kMergeThreads),
                                   h_n_src[i], mode, k, max_iteration, relative_fitness, relative_rmse, bufs[i].state, d_results + 20 * i);
            }
        }
        const int slot = enq[i] & 1;
        KPX_HIP(hipMemcpyAsync(&h_states[2 * i + slot], bufs[i].state, sizeof(IcpState), hipMemcpyDeviceToHost, ls));
        KPX_HIP(hipEventRecord(ev[i][slot], ls));
        ++enq[i];
        return KPX_OK;
    };
    // One chain of launches for up to kIcpBatchMax problems (icp_iter_batch_kernel) when the clouds were ordered by the batch sort;
    // a single group runs on the caller's stream itself (no fork / join events).  KPX_ICP_BATCH_LAUNCH=0: one chain per problem.
    static const bool batch_launch = [] { const char *e = getenv("KPX_ICP_BATCH_LAUNCH"); return !(e && e[0] == '0'); }();
    static const bool fused_env = [] { const char *e = getenv("KPX_ICP_FUSE"); return !(e && e[0] == '0'); }();
    const bool grouped = local_engine() && ordered && batch_launch && fused_env;
    const int n_chain = grouped ? (int)cdiv(count, kIcpBatchMax) : count;
    const bool on_caller = grouped && n_chain == 1;
    const int used_lanes = on_caller ? 0 : (n_chain < kBatchLanes ? n_chain : kBatchLanes);
    if (used_lanes) rc = lanes_fork(ln, st, used_lanes);
    if (grouped) {
        static thread_local unsigned long long *h_progress = nullptr;
        static thread_local unsigned long long generation = 0;
        if (!h_progress) KPX_HIP(hipHostMalloc((void **)&h_progress, 64 * sizeof(unsigned long long), hipHostMallocDefault));
        generation = (generation + 1) & 0xFFFFFFull;
        const unsigned long long tag = generation << 40;
        // Launches kept queued ahead of the newest progress word the host has seen.  Every launch queued beyond the one that turns out to
        // be the last still runs (its blocks read "done" and return: ~4 us each) IN FRONT of whatever the caller queues next, and with
        // several frames in flight those empty launches take dispatch slots from the other frames' chains: 3 instead of round 2's 6,
        // same box, four frames in flight 2556-2600 vs 2441-2448 Mpoints/s, one frame at a time equal (a launch lasts >= 20 us, the
        // progress word reaches the host in a few; 2 starts to starve: profiles/r04/exp_icp_window.txt).  KPX_ICP_WINDOW: A/B switch.
        static const int window = [] { const char *e = getenv("KPX_ICP_WINDOW"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? v : 3; }();
        static const double stall_limit = [] { const char *e = getenv("KPX_ICP_STALL_SECONDS"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 60.0; }();
        auto t_last = std::chrono::steady_clock::now();
        // split (default): the update of every registration runs in icp_solve_batch_kernel between the sweeps; KPX_ICP_SPLIT=0: in the
        // prologue of the next sweep's blocks (one launch per iteration).  Same-run A/B with four frames in flight: 1830-1930 vs
        // 1730-1830 Mpoints/s -- the redundant prologue holds every block's wave slots 4 us longer, and slots are what frames compete for.
        // KPX_ICP_SPLIT=2: in the LAST block of the sweep itself (no update kernel, no redundant prologue).
        static const int split = [] { const char *e = getenv("KPX_ICP_SPLIT"); return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : KPX_ICP_SPLIT_DEFAULT; }();
        const int last_k = split ? max_iteration : max_iteration + 1;     // the fused chain ends with an update-only launch
        // A/B switches: bit 0 LightSkip, bit 1 row certificates (they rest on LightSkip's motion bookkeeping)
        static const int light = [] {
            const char *e = getenv("KPX_ICP_LIGHT_SKIP"), *c = getenv("KPX_ICP_CERT");
            const char *ck = getenv("KPX_ICP_CERT_CHECK");
            const int l = (e && e[0] == '0') ? 0 : 1;
            const char *cs = getenv("KPX_ICP_CHAIN_STAMPS");
            return l | ((l && !(c && c[0] == '0')) ? 2 : 0) | ((ck && ck[0] == '1') ? 4 : 0) | ((cs && cs[0] == '1') ? 8 : 0);
        }();
        static const CertPolicy cert_policy = [] {
            auto f = [](const char *name, float dflt) { const char *e = getenv(name); return e ? (float)atof(e) : dflt; };
            return CertPolicy{ f("KPX_CERT_CALM", 0.15f), f("KPX_CERT_FACTOR", 3.0f), f("KPX_CERT_SKIN_MIN", 0.02f), f("KPX_CERT_SKIN_MAX", 0.2f) };
        }();
        // The one-launch chain (icp_chain_kernel) when the update is the last block's (split 2), the records hold the iterations and the
        // group's blocks fit the device beside the chains already in flight; KPX_ICP_CHAIN=0: always a launch per iteration.
        static const unsigned long long chain_limit = [] {
            const char *e = getenv("KPX_ICP_CHAIN_WAIT_SECONDS");
            const double v = e ? atof(e) : 0.0;
            return (unsigned long long)((v > 0.0 ? v : 2.0) * 1e8);
        }();
        static const bool chain_alone = [] { const char *e = getenv("KPX_ICP_CHAIN_ALONE"); return !(e && e[0] == '0'); }();     // A/B: 0 = also with other frames in flight
        const bool chain_ok = chain_form_on() && split == 2 && max_iteration <= kChainRecords - 2 && chain_abort_word() != nullptr &&
                              (!chain_alone || busy_threads() <= 1);
        // KPX_ICP_ROWS=0: the iterations through icp_iter_batch_kernel (a wave per 16-row tile, blocks of four) -- the A/B switch of
        // icp_rows_kernel (a wave per 64 rows), which serves the update in the last block (split 2) with LightSkip's bookkeeping
        // (1, the default: per launch, by how much of the registrations the last reported iteration still searched -- the progress word's
        // bits 33..39: a wave per tile while most rows are searched, a wave per 64 rows once most are certified (KPX_ICP_ROWS_SHARE: the
        // largest searched share, in 1/127ths, at which the rows form is taken); 2: always the rows form.  The forms agree bit for bit,
        // so the choice -- which follows the host's timing -- never shows in a result.)
        static const int rows_env = [] { const char *e = getenv("KPX_ICP_ROWS"); return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1; }();
        static const int rows_share = [] { const char *e = getenv("KPX_ICP_ROWS_SHARE"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 127 ? v : 3; }();
        const int rows_mode = split == 2 ? rows_env : 0;
        IcpBatchArgs A[8], Ac[8];                              // count <= 64: at most 8 groups
        int gk[8];
        bool gfin[8];
        unsigned gblocks[8];
        for (int g = 0; g < n_chain && !rc; ++g) {
            hipStream_t ls = on_caller ? st : lanes[g % kBatchLanes];
            const int i0 = g * kIcpBatchMax, i1 = i0 + kIcpBatchMax < count ? i0 + kIcpBatchMax : count;
            Mat16x8 T0;
            unsigned b0 = 0;
            for (int c = 0; c < kIcpBatchMax; ++c) {
                const int i = i0 + c < i1 ? i0 + c : i1 - 1;      // unused slots repeat the last problem (never addressed: count bounds the search)
                IcpProblem &P = A[g].p[c];
                P.src = h_src[i]; P.row_of = bufs[i].row_of; P.src_sorted = bufs[i].src_sorted; P.idx_sorted = bufs[i].idx_sorted;
                P.ptgt_sorted = bufs[i].ptgt_sorted; P.pair = bufs[i].state;
                P.idx_cur = nullptr; P.d2_cur = nullptr;          // a batch reports transforms, not correspondence lists
                P.ring = bufs[i].acc_fixed; P.result = d_results + 20 * i; P.progress = &h_progress[i]; P.n = h_n_src[i];
                P.light_key = bufs[i].light_key; P.sbbox = bufs[i].sort_s.bbox; P.cert = bufs[i].cert_sorted; P.thist = bufs[i].thist;
                P.chain_rec = chain_ok ? bufs[i].chain_rec : nullptr;
                P.block0 = b0; P.blocks = (unsigned)cdiv(h_n_src[i], kIRows);
                P.tgt = tgt; P.tn = tgt_normals; P.Bs = bufs[0].Bs; P.orig = bufs[0].orig_t; P.tile_box = bufs[0].tile_box; P.group_box = bufs[0].group_box;
                P.tbbox = bufs[0].sort_t.bbox; P.n_groups = tplan.l_groups; P.k = 0; P.tag = tag;
                Ac[g].p[c] = P;
                Ac[g].p[c].block0 = (unsigned)c; Ac[g].p[c].blocks = 1u;
                for (int e = 0; e < 16; ++e) T0.m[c][e] = h_init[16 * i + e];
                if (i0 + c < i1) { b0 += P.blocks; __atomic_store_n(&h_progress[i], 0ull, __ATOMIC_RELAXED); }
            }
            A[g].count = Ac[g].count = i1 - i0;
            gblocks[g] = b0;
            gk[g] = 0; gfin[g] = false;
            hipLaunchKernelGGL(icp_batch_init_kernel, dim3(b0), dim3(256), 0, ls, A[g], T0);
            if (chain_ok) {
                const bool launched = chain_launch_if_fits(b0, ls, [&] {
                    ProfScope prof(KPX_PROF_NN_LOCAL, 0.0, ls);
                    hipLaunchKernelGGL(icp_chain_kernel, dim3(b0), dim3(kIThreads), 0, ls, A[g], tgt, tgt_normals, bufs[0].Bs, bufs[0].orig_t, bufs[0].tile_box,
                                       bufs[0].group_box, tplan.l_groups, bufs[0].sort_t.bbox, md2, mode, max_iteration, relative_fitness, relative_rmse, tag,
                                       prof_armed() ? nn_visits_ptr() : (unsigned long long *)nullptr, light, cert_policy, chain_limit, chain_abort_word());
                });
                if (launched) gfin[g] = true;
                static const bool chain_dump = [] { const char *e = getenv("KPX_ICP_CHAIN_DUMP"); return e && e[0] == '1'; }();
                if (launched && chain_dump) {                // development aid: the records of every registration of the group, per iteration
                    (void)hipStreamSynchronize(ls);
                    static double rec[kChainRecords * kChainRec];
                    for (int c = 0; c < A[g].count; ++c) {
                        (void)hipMemcpy(rec, A[g].p[c].chain_rec, sizeof(rec), hipMemcpyDeviceToHost);
                        for (int k = 0; k < kChainRecords; ++k) {
                            const IcpState *r = reinterpret_cast<const IcpState *>(rec + (size_t)kChainRec * k);
                            unsigned long long w0;
                            memcpy(&w0, r, 8);
                            if (w0 == kChainEmpty) { fprintf(stderr, "chain problem %d record %d: empty\n", c, k); break; }
                            fprintf(stderr, "chain problem %d record %d: iter %d done %d fitness %.9f rmse %.9f count %.0f T03 %.6f motion %.4f last %.4f reach %.3f\n", c, k, r->iter,
                                    r->done, r->fitness, r->rmse, r->count, r->T[3], r->motion, r->last_motion, r->reach);
                        }
                    }
                }
            }
        }
        // a kpx_stream worker: the device's engine carries this group's iterations together with the other frames' (IcpEngine above)
        if (t_engine && split == 2 && !rc) {
            std::vector<std::unique_ptr<EngJob>> jobs;
            for (int g = 0; g < n_chain && !rc; ++g) {
                if (gfin[g]) continue;                       // (ran as a one-launch chain)
                jobs.emplace_back(new EngJob());
                EngJob &J = *jobs.back();
                J.count = A[g].count;
                for (int c = 0; c < J.count; ++c) {
                    J.p[c].P = A[g].p[c];
                    J.p[c].gen = generation; J.p[c].next_k = 0; J.p[c].seen = 0; J.p[c].share = 127; J.p[c].done = false; J.p[c].lane = -1; J.p[c].job = nullptr;
                }
                J.md2 = md2; J.rel_fit = relative_fitness; J.rel_rmse = relative_rmse; J.mode = mode; J.max_iter = max_iteration; J.light = light;
                J.rows_mode = rows_mode; J.rows_share = rows_share; J.window = window; J.pol = cert_policy;
                rc = engine_run_group(t_engine, J, on_caller ? st : lanes[g % kBatchLanes]);
                gfin[g] = true;
            }
        }
        for (bool pending = true; pending && !rc;) {
            pending = false;
            bool advanced = false;
            for (int g = 0; g < n_chain; ++g) {
                if (gfin[g]) continue;
                int seen = INT_MAX, share = 0;
                bool all_done = true;
                // the launches still to be queued carry only the registrations that have not converged yet (as far as the host
                // knows: the progress words lag by up to `window` launches); a converged problem's blocks in an already queued
                // launch read its state and return
                IcpBatchArgs act, actc;
                act.count = actc.count = 0;
                unsigned ab = 0;
                for (int c = 0; c < A[g].count; ++c) {
                    const unsigned long long w = __atomic_load_n(&h_progress[g * kIcpBatchMax + c], __ATOMIC_ACQUIRE);
                    const bool mine = (w >> 40) == generation;
                    if (mine && ((w >> 32) & 1ull)) continue;                          // this problem has converged
                    all_done = false;
                    const int sc = mine ? (int)(w & 0xFFFFFFFFull) : 0;
                    seen = sc < seen ? sc : seen;
                    const int sh_c = mine && sc >= 1 ? (int)((w >> 33) & 127ull) : 127;
                    share = sh_c > share ? sh_c : share;
                    act.p[act.count] = A[g].p[c];
                    act.p[act.count].block0 = ab;
                    ab += A[g].p[c].blocks;
                    actc.p[actc.count] = Ac[g].p[c];
                    actc.p[actc.count].block0 = (unsigned)actc.count;
                    ++act.count; ++actc.count;
                }
                if (all_done) { gfin[g] = true; continue; }
                for (int c = act.count; c < kIcpBatchMax; ++c) { act.p[c] = act.p[act.count - 1]; actc.p[c] = actc.p[actc.count - 1]; }
                hipStream_t ls = on_caller ? st : lanes[g % kBatchLanes];
                while (gk[g] <= last_k && gk[g] - seen < window) {
                    advanced = true;
                    const bool closing = gk[g] > max_iteration;
                    ProfScope prof(KPX_PROF_NN_LOCAL, 0.0, ls);
                    if (rows_mode == 2 || (rows_mode == 1 && share <= rows_share)) {
                        // one wave per 64 rows (kpx_icprows.h): every problem with its own operands and iteration number
                        const IcpProblem *pp[kRowsBatchMax];
                        int ks[kRowsBatchMax];
                        for (int c = 0; c < act.count; ++c) { pp[c] = &act.p[c]; ks[c] = gk[g]; }
                        rows_launch(pp, ks, act.count, md2, mode, max_iteration, relative_fitness, relative_rmse,
                                    prof_armed() ? nn_visits_ptr() : (unsigned long long *)nullptr, light, cert_policy, ls);
                        ++gk[g];
                        continue;
                    }
                    for (int c = 0; c < kIcpBatchMax; ++c) { act.p[c].k = gk[g]; actc.p[c].k = gk[g]; }
                    hipLaunchKernelGGL(icp_iter_batch_kernel, dim3(closing ? (unsigned)act.count : ab), dim3(kIThreads), 0, ls, closing ? actc : act, md2,
                                       mode, max_iteration, relative_fitness, relative_rmse,
                                       prof_armed() ? nn_visits_ptr() : (unsigned long long *)nullptr, split, split == 2 ? light : 0, cert_policy);
                    if (split == 1)
                        hipLaunchKernelGGL(icp_solve_batch_kernel, dim3((unsigned)act.count), dim3(256), 0, ls, act, mode, gk[g], max_iteration,
                                           relative_fitness, relative_rmse, tag);
                    ++gk[g];
                }
                if (gk[g] > last_k) { gfin[g] = true; continue; }
                pending = true;
            }
            if (advanced) t_last = std::chrono::steady_clock::now();
            else if (pending) {
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_last).count();
                if (waited > stall_limit)
                    rc = fail(KPX_ERR_HIP, "kpx_icp_batch: no progress for %.0f s (KPX_ICP_STALL_SECONDS)", stall_limit);
                static const bool trace = [] { const char *e = getenv("KPX_ICP_TRACE_STALL"); return e && e[0] == '1'; }();
                static thread_local double told = 0.0;
                if (trace && waited > 0.003 && waited > told + 0.003) {
                    told = waited;
                    fprintf(stderr, "icp_batch waiting %.1f ms: generation %llu gk %d", waited * 1e3, generation, gk[0]);
                    for (int c = 0; c < A[0].count; ++c) {
                        const unsigned long long w = __atomic_load_n(&h_progress[c], __ATOMIC_ACQUIRE);
                        fprintf(stderr, " | p%d gen %llu done %llu k %llu", c, w >> 40, (w >> 32) & 1ull, w & 0xFFFFFFFFull);
                    }
                    fprintf(stderr, " query %d\n", (int)hipStreamQuery(on_caller ? st : lanes[0]));
                } else if (waited < 0.001) told = 0.0;
                __builtin_ia32_pause();
            }
        }
        if (hipGetLastError() != hipSuccess && !rc) rc = fail(KPX_ERR_HIP, "kpx_icp_batch: launch failed");
    } else if (local_engine()) {
        // Culled engine: every update kernel publishes "iterations finished | converged" in a pinned word of its problem.
        // The driver keeps a window of iterations queued per problem and tops it up as the words advance: no copies, no
        // events, and at most `window` launches wasted after a problem converges (they return at once on its flag).
        static thread_local unsigned long long *h_progress = nullptr;
        static thread_local unsigned long long generation = 0;
        if (!h_progress) KPX_HIP(hipHostMalloc((void **)&h_progress, 64 * sizeof(unsigned long long), hipHostMallocDefault));
        generation = (generation + 1) & 0xFFFFFFull;
        const unsigned long long tag = generation << 40;
        constexpr int window = 6;
        static const bool fused = [] { const char *e = getenv("KPX_ICP_FUSE"); return !(e && e[0] == '0'); }();   // A/B switch: 0 = update in its own kernel
        bool fin[64];
        for (int i = 0; i < count && !rc; ++i) {
            hipStream_t ls = lanes[i % kBatchLanes];
            __atomic_store_n(&h_progress[i], 0ull, __ATOMIC_RELAXED);
            hipLaunchKernelGGL(icp_init_kernel, dim3(1), dim3(1), 0, ls, bufs[i].state, mat16_from(h_init + 16 * i));
            rc = nn_prep_source(h_src[i], plans[i], bufs[i], ls, ordered);
            next_k[i] = 0;
            fin[i] = false;
        }
        // "no progress" = no progress word advanced and nothing could be queued for `stall_limit` seconds (the clock restarts
        // on every advance: the lanes first wait for whatever the caller already queued on `stream`, and a large batch runs long)
        static const double stall_limit = [] { const char *e = getenv("KPX_ICP_STALL_SECONDS"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 60.0; }();
        auto t_last = std::chrono::steady_clock::now();
        for (bool pending = true; pending && !rc;) {
            pending = false;
            bool advanced = false;
            for (int i = 0; i < count; ++i) {
                if (fin[i]) continue;
                const unsigned long long w = __atomic_load_n(&h_progress[i], __ATOMIC_ACQUIRE);
                const bool mine = (w >> 40) == generation;
                const int seen = mine ? (int)(w & 0xFFFFFFFFull) : 0;
                if (mine && ((w >> 32) & 1ull)) { fin[i] = true; continue; }               // converged
                hipStream_t ls = lanes[i % kBatchLanes];
                const int last_k = fused ? max_iteration + 1 : max_iteration;               // the fused chain ends with an update-only launch
                while (next_k[i] <= last_k && next_k[i] - seen < window) {
                    advanced = true;
                    if (fused)
                        icp_fused_launch(h_src[i], tgt, tgt_normals, plans[i], bufs[i], md2, mode, next_k[i], max_iteration, relative_fitness,
                                         relative_rmse, d_results + 20 * i, ls, &h_progress[i], tag);
                    else
                        icp_iter_launch(h_src[i], tgt, tgt_normals, plans[i], bufs[i], md2, mode, next_k[i], max_iteration, relative_fitness,
                                        relative_rmse, d_results + 20 * i, ls, &h_progress[i], tag);
                    ++next_k[i];
                }
                if (next_k[i] > last_k) { fin[i] = true; continue; }                        // everything is queued
                pending = true;
            }
            if (advanced) t_last = std::chrono::steady_clock::now();
            else if (pending) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_last).count() > stall_limit)
                    rc = fail(KPX_ERR_HIP, "kpx_icp_batch: no progress for %.0f s (KPX_ICP_STALL_SECONDS)", stall_limit);
                __builtin_ia32_pause();
            }
        }
        if (hipGetLastError() != hipSuccess && !rc) rc = fail(KPX_ERR_HIP, "kpx_icp_batch: launch failed");
    } else {
    for (int i = 0; i < count && !rc; ++i) {
        hipStream_t ls = lanes[i % kBatchLanes];
        hipLaunchKernelGGL(icp_init_kernel, dim3(1), dim3(1), 0, ls, bufs[i].state, mat16_from(h_init + 16 * i));
        rc = nn_prep_source(h_src[i], plans[i], bufs[i], ls);
        next_k[i] = 0; enq[i] = 0; polled[i] = 0;
        for (int c = 0; c < in_flight && !rc && next_k[i] <= max_iteration; ++c) rc = enqueue_chunk(i);
    }
    for (bool busy = true; busy && !rc;) {
        busy = false;
        for (int i = 0; i < count && !rc; ++i) {
            if (polled[i] >= enq[i]) continue;
            busy = true;
            const int slot = polled[i] & 1;
            if (hipEventSynchronize(ev[i][slot]) != hipSuccess) { rc = fail(KPX_ERR_HIP, "convergence poll failed"); break; }
            const IcpState hs = h_states[2 * i + slot];
            ++polled[i];
            if (hs.done || next_k[i] > max_iteration) continue;
            policy[i].observe(hs.fitness, hs.rmse);
            rc = enqueue_chunk(i);
        }
    }
    }
    if (used_lanes) {
        const int jrc = lanes_join(ln, st, used_lanes);
        rc = rc ? rc : jrc;
    }
    if (rc) for (int l = 0; l < used_lanes; ++l) (void)hipStreamSynchronize(lanes[l]);     // leave nothing in flight on an error
    if (use_events)
        for (int i = 0; i < count; ++i)
            for (int e = 0; e < 2; ++e) (void)hipEventDestroy(ev[i][e]);
    if (rc) return rc;
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}
