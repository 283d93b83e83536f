// kpx_voxel.hip -- a7: PointCloud.voxel_down_sample (preprocessing/filtering.py:23,
// preprocessing/registration.py:8,100,101).
//   origin = min_bound - v/2 ; index = floor((p - origin)/v)   (fp64, identical to the oracle)
//   63-bit key (21 bits per axis), stable radix sort of (key, point id), one thread per voxel sums
//   its points in ascending original index (fp64, sequential) -> bit-exact means.
// Output order: ascending (ix,iy,iz).
#include <hipcub/hipcub.hpp>

#include "kpx_internal.h"
#include "kpx_radix.h"

namespace kpx {

__global__ __launch_bounds__(256) void voxel_key_kernel(const float *__restrict__ pts, int64_t n, const double *__restrict__ bbox,
                                                        double voxel, uint64_t *__restrict__ keys, int32_t *__restrict__ vals,
                                                        int32_t *__restrict__ err)
{
    const double ox = bbox[0] - voxel * 0.5, oy = bbox[1] - voxel * 0.5, oz = bbox[2] - voxel * 0.5;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double fx = floor(((double)pts[3 * i] - ox) / voxel);
        double fy = floor(((double)pts[3 * i + 1] - oy) / voxel);
        double fz = floor(((double)pts[3 * i + 2] - oz) / voxel);
        bool bad = !(fx >= 0.0) || !(fy >= 0.0) || !(fz >= 0.0) || fx >= 2097152.0 || fy >= 2097152.0 || fz >= 2097152.0;
        if (bad) { *err = 1; fx = fy = fz = 0.0; }
        keys[i] = ((uint64_t)fx << 42) | ((uint64_t)fy << 21) | (uint64_t)fz;
        vals[i] = (int32_t)i;
    }
}

template <class Key> struct HeadPredT {
    const Key *keys;
    __device__ bool operator()(int64_t s, int) const { return s == 0 || keys[s] != keys[s - 1]; }
};
using HeadPred = HeadPredT<uint64_t>;
struct HeadEmit {
    int32_t *seg_start;
    __device__ void operator()(int64_t s, int, int32_t dst) const { seg_start[dst] = (int32_t)s; }
};

__global__ __launch_bounds__(256) void voxel_mean_kernel(const float *__restrict__ pts, const float *__restrict__ col,
                                                         const float *__restrict__ nrm, int64_t n,
                                                         const int32_t *__restrict__ vals, const int32_t *__restrict__ seg_start,
                                                         int32_t *__restrict__ d_count, const int32_t *__restrict__ err,
                                                         float *__restrict__ opts, float *__restrict__ ocol, float *__restrict__ onrm)
{
    const int32_t m_total = *d_count;
    // index overflow is reported through the count word (the host sees KPX_ERR_RANGE when it reads it); a block that reads
    // the count after this store sees a negative total and does nothing -- the output is invalid in that case anyway
    if (blockIdx.x == 0 && threadIdx.x == 0 && *err) *d_count = KPX_ERR_RANGE;
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < m_total; m += (int64_t)gridDim.x * blockDim.x) {
        int64_t s0 = seg_start[m], s1 = (m + 1 < m_total) ? seg_start[m + 1] : n;
        double sp[3] = { 0, 0, 0 }, sc[3] = { 0, 0, 0 }, sn[3] = { 0, 0, 0 };
        // The sums stay sequential in ascending point index (the contract); only the LOADS of 8 points are issued
        // together -- a dense voxel (a wall patch close to the camera holds 50+ points) was a chain of 2 dependent
        // global loads per point, and the longest voxel set the kernel's duration.
        for (int64_t s = s0; s < s1; s += 8) {
            const int cnt = (int)(s1 - s < 8 ? s1 - s : 8);
            int64_t p[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = k < cnt ? vals[s + k] : -1;
            float vp[8][3], vc[8][3], vn[8][3];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (p[k] < 0) continue;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    vp[k][a] = pts[3 * p[k] + a];
                    if (col) vc[k][a] = col[3 * p[k] + a];
                    if (nrm) vn[k][a] = nrm[3 * p[k] + a];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (p[k] < 0) continue;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    sp[a] += (double)vp[k][a];
                    if (col) sc[a] += (double)vc[k][a];
                    if (nrm) sn[a] += (double)vn[k][a];
                }
            }
        }
        double c = (double)(s1 - s0);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            opts[3 * m + a] = (float)(sp[a] / c);
            if (col && ocol) ocol[3 * m + a] = (float)(sc[a] / c);
        }
        if (nrm && onrm) {
            double nn = sqrt(fma(sn[2], sn[2], fma(sn[1], sn[1], sn[0] * sn[0])));
#pragma unroll
            for (int a = 0; a < 3; ++a) onrm[3 * m + a] = (float)(nn > 0 ? sn[a] / nn : sn[a]);
        }
    }
}
static int voxel_impl(const float *pts, const float *col, const float *nrm, int64_t n, double voxel, float *opts,
                      float *ocol, float *onrm, int32_t *d_count, Arena &a, hipStream_t st)
{
    const size_t nn = (size_t)(n > 0 ? n : 1);
    uint64_t *keys_in = a.get<uint64_t>(nn), *keys_out = a.get<uint64_t>(nn);
    int32_t *vals_in = a.get<int32_t>(nn), *vals_out = a.get<int32_t>(nn);
    int32_t *seg_start = a.get<int32_t>(nn);
    int32_t *counts = a.get<int32_t>((size_t)compact_ws_ints(n));
    double *part = a.get<double>((size_t)kBboxBlocks * 6 + 8);
    int32_t *err = a.get<int32_t>(1);
    size_t sort_bytes = memo_bytes(4, (int64_t)nn, [&] { size_t b = 0; (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, keys_in, keys_out, vals_in, vals_out, (int)nn, 0, 63, st); return b; });
    char *sort_tmp = a.get<char>(sort_bytes);
    if (a.dry) return KPX_OK;
    KPX_ARENA_CHECK(a);
    double *bbox = part + (size_t)kBboxBlocks * 6;
    int rc = bbox_f32(pts, n, bbox, part, st);
    if (rc) return rc;
    KPX_HIP(hipMemsetAsync(err, 0, sizeof(int32_t), st));
    int nb = (int)(cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256));
    hipLaunchKernelGGL(voxel_key_kernel, dim3(nb), dim3(256), 0, st, pts, n, bbox, voxel, keys_in, vals_in, err);
    KPX_HIP(hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 63, st));
    rc = compact(HeadPred{ keys_out }, HeadEmit{ seg_start }, n, 1, counts, d_count, st);
    if (rc) return rc;
    hipLaunchKernelGGL(voxel_mean_kernel, dim3(nb), dim3(256), 0, st, pts, col, nrm, n, vals_out, seg_start, d_count, err, opts,
                       ocol, onrm);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

// ---- several clouds in ONE pass ---------------------------------------------------------------------------------
// A voxel grid of an 85k-point cloud is ~28 short launches (bounding box, keys, the merge passes of the sort, head
// compaction, means), and the pipeline down-samples 4 clouds per frame: queued on four lanes the work overlaps on the
// GPU, but the host cannot issue 112 launches faster than ~0.6 ms.  Here the clouds are handled as one concatenated
// array: per-cloud bounding boxes, the cloud number as the most significant digit of the key -- mixed radix
// ((c DX + ix) DY + iy) DZ + iz with DX, DY, DZ the largest grid extents of the batch, so it fits 64 bits whenever the
// single-cloud key does -- ONE stable sort, ONE head compaction, ONE mean kernel.  Results are identical to the
// single-cloud path: same voxel indices (own origin per cloud), ascending (ix, iy, iz) per cloud, sums in ascending
// point index.
constexpr int kVoxelBatchMax = 8;
constexpr int kVoxelBatchBboxBlocks = 64;
struct VoxelBatch {
    const float *pts[kVoxelBatchMax];
    const float *col[kVoxelBatchMax];
    float *opts[kVoxelBatchMax];
    float *ocol[kVoxelBatchMax];
    int64_t off[kVoxelBatchMax + 1];      // cloud c owns [off[c], off[c+1]) of the concatenated index space
    int32_t count;
    int32_t morton;                       // keys = cloud | curve code of (ix, iy, iz) instead of cloud | (ix, iy, iz) row-major: 1 Z-curve, 2 Hilbert curve
};
// bits of the Z-curve code per axis: enough for the axis' largest index in the batch.  The code interleaves bit q of every axis that
// still has a bit q (x lowest), so its width is the SUM of the three widths -- a long axis costs its own extra bits only, not
// three times them (a frame's 26-bit cube code would be a fourth radix pass; 7 + 7 + 8 bits + 2 for the cloud stay within three)
__device__ __forceinline__ void voxel_batch_axis_bits(const double d[3], int bits[3])
{
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const unsigned long long m = (unsigned long long)d[a] - 1ull;          // largest index
        int n = 1;
        while (n < 21 && (m >> n) != 0ull) ++n;
        bits[a] = n;
    }
}
__device__ __forceinline__ unsigned long long voxel_zcode(unsigned long long x, unsigned long long y, unsigned long long z, const int bits[3])
{
    unsigned long long k = 0ull;
    int o = 0;
    const int top = bits[0] > bits[1] ? (bits[0] > bits[2] ? bits[0] : bits[2]) : (bits[1] > bits[2] ? bits[1] : bits[2]);
    for (int q = 0; q < top; ++q) {
        if (q < bits[0]) k |= ((x >> q) & 1ull) << o++;
        if (q < bits[1]) k |= ((y >> q) & 1ull) << o++;
        if (q < bits[2]) k |= ((z >> q) & 1ull) << o++;
    }
    return k;
}
// Hilbert index of (x, y, z), `bits` bits per axis (Skilling, "Programming the Hilbert curve", 2004: axes -> transpose, then the bits
// of the three transposed words interleaved from the top).  Consecutive cells of the curve are neighbours, which the Z-curve's
// are not: 16 consecutive points of a surface cloud -- a wave's rows, a target tile of the culled ICP sweep (kpx_nnlocal.h) -- span
// 177 instead of 246 mm (median; p99 677 instead of 1255) on the bench's 35 mm clouds, a wave multiplies 2.0 instead of 3.0 tiles
// on average (p99 9 instead of 13).  KPX_VOXEL_CURVE=z restores the Z-curve (A/B switch).  `bits` >= 1.
__device__ __forceinline__ unsigned long long voxel_hcode(unsigned long long x, unsigned long long y, unsigned long long z, int bits)
{
    unsigned long long X[3] = { x, y, z };
    const unsigned long long M = 1ull << (bits - 1);
    for (unsigned long long Q = M; Q > 1ull; Q >>= 1) {
        const unsigned long long P = Q - 1ull;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const unsigned long long t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    unsigned long long t = 0ull;
    for (unsigned long long Q = M; Q > 1ull; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1ull;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    unsigned long long k = 0ull;
    for (int b = bits - 1; b >= 0; --b) {
#pragma unroll
        for (int i = 0; i < 3; ++i) k = (k << 1) | ((X[i] >> b) & 1ull);
    }
    return k;
}
// Width m of the cube the Hilbert code covers when the axes need ab[0..2] bits: the largest m <= max(ab) whose key -- the axes' bits
// above m, row-major, then the 3 m bits of the cube's code -- takes no more 8-bit sort passes than the Z-code's ab[0] + ab[1] + ab[2]
// bits (cloud number included: cb bits).  m = min(ab) always qualifies; a frame's 7 + 7 + 8 bits give m = 7: two cubes stacked
// along z, one jump of the curve between them, three passes as before.
__device__ __forceinline__ int voxel_hilbert_cube_bits(const int ab[3], int cb)
{
    const int top = ab[0] > ab[1] ? (ab[0] > ab[2] ? ab[0] : ab[2]) : (ab[1] > ab[2] ? ab[1] : ab[2]);
    const int low = ab[0] < ab[1] ? (ab[0] < ab[2] ? ab[0] : ab[2]) : (ab[1] < ab[2] ? ab[1] : ab[2]);
    const int passes = (ab[0] + ab[1] + ab[2] + cb + 7) / 8;
    for (int m = top; m > low; --m) {
        int n = 3 * m + cb;
#pragma unroll
        for (int a = 0; a < 3; ++a) n += ab[a] > m ? ab[a] - m : 0;
        if (n <= 64 && (n + 7) / 8 <= passes) return m;
    }
    return low;
}
__device__ __forceinline__ int voxel_hilbert_key_bits(const int ab[3], int m)
{
    int n = 3 * m;
#pragma unroll
    for (int a = 0; a < 3; ++a) n += ab[a] > m ? ab[a] - m : 0;
    return n;
}
// key of one voxel: [x >> m | y >> m | z >> m] (row-major over the cubes) above the cube's Hilbert code of the low m bits
__device__ __forceinline__ unsigned long long voxel_hilbert_key(unsigned long long x, unsigned long long y, unsigned long long z, const int ab[3], int m)
{
    const unsigned long long mask = (1ull << m) - 1ull;
    const int ex = ab[0] > m ? ab[0] - m : 0, ey = ab[1] > m ? ab[1] - m : 0, ez = ab[2] > m ? ab[2] - m : 0;
    (void)ex;
    const unsigned long long hi = (((x >> m) << ey | (y >> m)) << ez) | (z >> m);
    return (hi << (3 * m)) | voxel_hcode(x & mask, y & mask, z & mask, m);
}
__device__ __forceinline__ int voxel_batch_cloud(const VoxelBatch &b, int64_t i)
{
    int c = 0;
#pragma unroll
    for (int k = 1; k < kVoxelBatchMax; ++k) c += (k < b.count && i >= b.off[k]) ? 1 : 0;
    return c;
}
__global__ __launch_bounds__(256) void voxel_batch_bbox_partial_kernel(VoxelBatch b, double *__restrict__ part)
{
    __shared__ float sh[6][4];
    const int c = blockIdx.y;
    const float *pts = b.pts[c];
    const int64_t n = b.off[c + 1] - b.off[c];
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { float v = pts[3 * i + a]; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]); }
    if (lane_id() == 0)
        for (int a = 0; a < 3; ++a) { sh[a][wave_id()] = mn[a]; sh[3 + a][wave_id()] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sh[threadIdx.x][0];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, sh[threadIdx.x][w]) : fmaxf(v, sh[threadIdx.x][w]);
        part[((int64_t)c * kVoxelBatchBboxBlocks + blockIdx.x) * 6 + threadIdx.x] = (double)v;
    }
}
// one wave per cloud: folds the partial boxes, bbox[c][0..5]; zeroes the cloud's error word
__global__ __launch_bounds__(64) void voxel_batch_bbox_final_kernel(const double *__restrict__ part, double *__restrict__ bbox, int32_t *__restrict__ err)
{
    const int c = blockIdx.x, lane = lane_id();
    double v[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) v[a] = part[((int64_t)c * kVoxelBatchBboxBlocks + lane) * 6 + a];
#pragma unroll
    for (int a = 0; a < 3; ++a) { v[a] = wave_min(v[a]); v[3 + a] = wave_max(v[3 + a]); }
    if (lane == 0) {
        for (int a = 0; a < 6; ++a) bbox[8 * c + a] = v[a];
        err[c] = 0;
    }
}
// largest grid extents of the batch (index < floor((max - origin) / v) + 1) and whether count x DX x DY x DZ fits 64 bits
__device__ __forceinline__ void voxel_batch_dims(const VoxelBatch &b, const double *__restrict__ bbox, double voxel, double d[3], int *overflow)
{
    d[0] = d[1] = d[2] = 1.0;
    for (int c = 0; c < b.count; ++c) {
        if (b.off[c + 1] == b.off[c]) continue;
        for (int a = 0; a < 3; ++a) {
            const double e = floor((bbox[8 * c + 3 + a] - (bbox[8 * c + a] - voxel * 0.5)) / voxel) + 1.0;
            if (e > d[a] && e < 2097152.0) d[a] = e;              // out-of-range clouds are flagged per point by the key kernel
        }
    }
    *overflow = ((double)b.count * d[0]) * (d[1] * d[2]) >= 18446744073709551616.0 ? 1 : 0;
}
// bits the keys of this batch occupy: batches above rocPRIM's merge-sort limit are sorted by Onesweep, one pass per 8 bits --
// a 4-sensor frame needs ~24 of the 64
__device__ __forceinline__ int voxel_batch_key_bits(const VoxelBatch &b, const double d[3], int overflow)
{
    int n = 64;
    if (!overflow && b.morton) {
        int cb = 0, ab[3];
        while ((1 << cb) < b.count) ++cb;
        voxel_batch_axis_bits(d, ab);
        n = (b.morton == 2 ? voxel_hilbert_key_bits(ab, voxel_hilbert_cube_bits(ab, cb)) : ab[0] + ab[1] + ab[2]) + cb;
        n = n < 1 ? 1 : (n > 64 ? 64 : n);
    } else if (!overflow) {
        const unsigned long long range = (((unsigned long long)b.count * (unsigned long long)d[0]) * (unsigned long long)d[1]) * (unsigned long long)d[2];
        n = 1;
        while (n < 64 && (range >> n) != 0ull) ++n;
    }
    return n;
}
__global__ void voxel_batch_bits_kernel(VoxelBatch b, const double *__restrict__ bbox, double voxel, int32_t *__restrict__ bits)
{
    double d[3];
    int overflow;
    voxel_batch_dims(b, bbox, voxel, d, &overflow);
    *bits = voxel_batch_key_bits(b, d, overflow);
}
template <class Key>
__global__ __launch_bounds__(256) void voxel_batch_key_kernel(VoxelBatch b, const double *__restrict__ bbox, double voxel,
                                                              Key *__restrict__ keys, int32_t *__restrict__ vals, int32_t *__restrict__ err,
                                                              int32_t *__restrict__ bits_out)
{
    __shared__ double dims[3];
    __shared__ int overflow, axis_bits[3], cube_bits;
    if (threadIdx.x == 0) {
        double d[3];
        int ov;
        voxel_batch_dims(b, bbox, voxel, d, &ov);
        if (bits_out && blockIdx.x == 0) *bits_out = voxel_batch_key_bits(b, d, ov);      // speculated width: the caller checks it afterwards
        dims[0] = d[0]; dims[1] = d[1]; dims[2] = d[2];
        int ab[3], cb = 0;
        voxel_batch_axis_bits(d, ab);
        axis_bits[0] = ab[0]; axis_bits[1] = ab[1]; axis_bits[2] = ab[2];
        while ((1 << cb) < b.count) ++cb;
        cube_bits = voxel_hilbert_cube_bits(ab, cb);
        overflow = (ov || (b.morton == 1 && ab[0] + ab[1] + ab[2] + cb > 64) || (b.morton == 2 && voxel_hilbert_key_bits(ab, cube_bits) + cb > 64)) ? 1 : 0;
    }
    __syncthreads();
    const uint64_t DX = (uint64_t)dims[0], DY = (uint64_t)dims[1], DZ = (uint64_t)dims[2];
    const int64_t total = b.off[b.count];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = voxel_batch_cloud(b, i);
        const float *pts = b.pts[c];
        const int64_t j = i - b.off[c];
        const double ox = bbox[8 * c] - voxel * 0.5, oy = bbox[8 * c + 1] - voxel * 0.5, oz = bbox[8 * c + 2] - voxel * 0.5;
        double fx = floor(((double)pts[3 * j] - ox) / voxel);
        double fy = floor(((double)pts[3 * j + 1] - oy) / voxel);
        double fz = floor(((double)pts[3 * j + 2] - oz) / voxel);
        const bool bad = overflow || !(fx >= 0.0) || !(fy >= 0.0) || !(fz >= 0.0) || fx >= 2097152.0 || fy >= 2097152.0 || fz >= 2097152.0;
        if (bad) { err[c] = 1; fx = fy = fz = 0.0; }
        if (b.morton == 2) {
            const int ab[3] = { axis_bits[0], axis_bits[1], axis_bits[2] };
            const int m = cube_bits;
            keys[i] = (Key)(((uint64_t)c << voxel_hilbert_key_bits(ab, m)) | voxel_hilbert_key((uint64_t)fx, (uint64_t)fy, (uint64_t)fz, ab, m));
        } else if (b.morton) {
            const int ab[3] = { axis_bits[0], axis_bits[1], axis_bits[2] };
            keys[i] = (Key)(((uint64_t)c << (ab[0] + ab[1] + ab[2])) | voxel_zcode((uint64_t)fx, (uint64_t)fy, (uint64_t)fz, ab));
        }
        else keys[i] = (Key)((((uint64_t)c * DX + (uint64_t)fx) * DY + (uint64_t)fy) * DZ + (uint64_t)fz);
        vals[i] = (int32_t)i;
    }
}
// head[c] = number of voxels (segment heads) before cloud c in the sorted array; d_counts[c] = voxels of cloud c
__global__ __launch_bounds__(64) void voxel_batch_locate_kernel(VoxelBatch b, const int32_t *__restrict__ seg_start, const int32_t *__restrict__ d_total,
                                                                const int32_t *__restrict__ err, int32_t *__restrict__ head, int32_t *__restrict__ d_counts)
{
    __shared__ int32_t h[kVoxelBatchMax + 1];
    const int c = threadIdx.x;
    const int32_t m = *d_total;
    if (c <= b.count) {
        int32_t lo = 0, hi = m;                      // first head position >= off[c]
        const int64_t want = b.off[c];
        while (lo < hi) {
            const int32_t mid = lo + (hi - lo) / 2;
            if (seg_start[mid] < want) lo = mid + 1; else hi = mid;
        }
        h[c] = c == b.count ? m : lo;
        head[c] = h[c];
    }
    __syncthreads();
    if (c < b.count) d_counts[c] = err[c] ? KPX_ERR_RANGE : h[c + 1] - h[c];
}
__global__ __launch_bounds__(256) void voxel_batch_mean_kernel(VoxelBatch b, const int32_t *__restrict__ vals, const int32_t *__restrict__ seg_start,
                                                               const int32_t *__restrict__ d_total, const int32_t *__restrict__ head)
{
    const int32_t m_total = *d_total;
    const int64_t total = b.off[b.count];
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < m_total; m += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s0 = seg_start[m], s1 = (m + 1 < m_total) ? seg_start[m + 1] : total;
        const int c = voxel_batch_cloud(b, s0);      // the sorted array keeps the clouds' ranges: position s0 tells the cloud
        const float *pts = b.pts[c], *col = b.col[c];
        const int64_t base = b.off[c];
        double sp[3] = { 0, 0, 0 }, sc[3] = { 0, 0, 0 };
        for (int64_t s = s0; s < s1; s += 8) {       // loads of 8 points issued together, sums sequential (as voxel_mean_kernel)
            const int cnt = (int)(s1 - s < 8 ? s1 - s : 8);
            int64_t p[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = k < cnt ? (int64_t)vals[s + k] - base : -1;
            float vp[8][3], vc[8][3];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (p[k] < 0) continue;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    vp[k][a] = pts[3 * p[k] + a];
                    if (col) vc[k][a] = col[3 * p[k] + a];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (p[k] < 0) continue;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    sp[a] += (double)vp[k][a];
                    if (col) sc[a] += (double)vc[k][a];
                }
            }
        }
        const double cn = (double)(s1 - s0);
        const int64_t o = m - head[c];
        float *opts = b.opts[c], *ocol = b.ocol[c];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            opts[3 * o + a] = (float)(sp[a] / cn);
            if (col && ocol) ocol[3 * o + a] = (float)(sc[a] / cn);
        }
    }
}

struct VoxelBatchScratch {
    uint64_t *keys_in, *keys_out;
    int32_t *vals_in, *vals_out, *seg_start, *counts, *err, *head, *d_total;
    double *part, *bbox;
    char *sort_tmp;
    size_t sort_bytes;
    RadixScratch rx;            // the hand-written sort (keys of at most 32 bits, total <= kRadixMaxPairs)
    char *counts_end;
};
static void voxel_batch_carve(Arena &a, int64_t total, VoxelBatchScratch *s)
{
    const size_t nn = (size_t)(total > 0 ? total : 1);
    s->keys_in = a.get<uint64_t>(nn); s->keys_out = a.get<uint64_t>(nn);
    s->vals_in = a.get<int32_t>(nn); s->vals_out = a.get<int32_t>(nn);
    s->seg_start = a.get<int32_t>(nn);
    s->err = a.get<int32_t>(kVoxelBatchMax);
    s->head = a.get<int32_t>(kVoxelBatchMax + 1);
    s->d_total = a.get<int32_t>(1);
    s->part = a.get<double>((size_t)kVoxelBatchMax * kVoxelBatchBboxBlocks * 6);
    s->bbox = a.get<double>((size_t)kVoxelBatchMax * 8);
    s->sort_bytes = memo_bytes(5, (int64_t)nn, [&] { size_t b = 0; (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, s->keys_in, s->keys_out, s->vals_in, s->vals_out, (int)nn, 0, 64, (hipStream_t) nullptr); return b; });
    s->sort_tmp = a.get<char>(s->sort_bytes);
    radix_carve(a, total <= kRadixMaxPairs ? total : kRadixMaxPairs, &s->rx);      // unconditional: the workspace size stays monotonic in the point count
    s->counts = a.get<int32_t>((size_t)compact_ws_ints(total));          // right behind the sort's cleared histograms: one memset for both
    s->counts_end = reinterpret_cast<char *>(s->counts + (size_t)compact_ws_ints(total));
}
// spec_bits > 0 (frame loop): the key width is NOT read back -- the sort covers spec_bits bits and *d_bits receives the width the
// batch really needs; the caller compares the two after its own read-back of the counts and repeats the call with spec_bits = 0
// when the speculation was too narrow (the outputs of that call are garbage).
static int voxel_batch_impl(const VoxelBatch &b, double voxel, int32_t *d_counts, Arena &a, hipStream_t st, int spec_bits = 0,
                            int32_t *d_bits = nullptr)
{
    const int64_t total = b.off[b.count];
    VoxelBatchScratch s;
    voxel_batch_carve(a, total, &s);
    KPX_ARENA_CHECK(a);
    hipLaunchKernelGGL(voxel_batch_bbox_partial_kernel, dim3(kVoxelBatchBboxBlocks, b.count), dim3(256), 0, st, b, s.part);
    hipLaunchKernelGGL(voxel_batch_bbox_final_kernel, dim3(b.count), dim3(64), 0, st, s.part, s.bbox, s.err);
    const int nb = (int)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256));
    int end_bit = 64;
    int32_t *bits_out = nullptr;
    if (spec_bits > 0 && d_bits) {
        bits_out = d_bits;                               // written by the key kernel itself
        end_bit = spec_bits > 64 ? 64 : spec_bits;
    } else if (total > (int64_t)128 * 1024) {
        // reading the key width back (one small round trip; the caller waits for the counts anyway) lets keys of at most 32 bits
        // (a 4-sensor frame needs ~25, a 1M-point room at 10 mm 25) be written, sorted and compared as 32-bit words by the library's
        // own radix sort -- three 8-bit passes instead of the vendor sort's eight over 64-bit keys.  Below ~128k points the vendor's
        // merge sort of a handful of launches is as fast as the round trip.
        static thread_local int32_t *h_bits = nullptr;
        if (!h_bits) KPX_HIP(hipHostMalloc((void **)&h_bits, sizeof(int32_t), hipHostMallocDefault));
        hipLaunchKernelGGL(voxel_batch_bits_kernel, dim3(1), dim3(1), 0, st, b, s.bbox, voxel, s.head);
        KPX_HIP(hipMemcpyAsync(h_bits, s.head, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (d_bits) KPX_HIP(hipMemcpyAsync(d_bits, s.head, sizeof(int32_t), hipMemcpyDefault, st));      // d_bits may be pinned host memory (kpx_frame_step)
        KPX_HIP(hipStreamSynchronize(st));
        end_bit = *h_bits < 1 ? 1 : (*h_bits > 64 ? 64 : *h_bits);
    } else if (d_bits) {
        hipLaunchKernelGGL(voxel_batch_bits_kernel, dim3(1), dim3(1), 0, st, b, s.bbox, voxel, d_bits);      // for the caller's next speculation
    }
    size_t bytes = s.sort_bytes;
    int rc;
    if (end_bit <= 32) {
        uint32_t *k_in = reinterpret_cast<uint32_t *>(s.keys_in), *k_out = reinterpret_cast<uint32_t *>(s.keys_out);
        hipLaunchKernelGGL(voxel_batch_key_kernel<uint32_t>, dim3(nb), dim3(256), 0, st, b, s.bbox, voxel, k_in, s.vals_in, s.err, bits_out);
        static const bool vendor_sort = [] { const char *e = getenv("KPX_RADIX"); return e && e[0] == '0'; }();       // A/B switch
        bool cleared = false;
        if (total <= kRadixMaxPairs && !vendor_sort) {
            rc = radix_sort_pairs_u32(s.rx, k_in, k_out, s.vals_in, s.vals_out, total, end_bit, st, s.counts_end);
            if (rc) return rc;
            cleared = true;
        } else {
            KPX_HIP(hipcub::DeviceRadixSort::SortPairs(s.sort_tmp, bytes, k_in, k_out, s.vals_in, s.vals_out, (int)total, 0, end_bit, st));
        }
        rc = compact(HeadPredT<uint32_t>{ k_out }, HeadEmit{ s.seg_start }, total, 1, s.counts, s.d_total, st, cleared);
    } else {
        hipLaunchKernelGGL(voxel_batch_key_kernel<uint64_t>, dim3(nb), dim3(256), 0, st, b, s.bbox, voxel, s.keys_in, s.vals_in, s.err, bits_out);
        KPX_HIP(hipcub::DeviceRadixSort::SortPairs(s.sort_tmp, bytes, s.keys_in, s.keys_out, s.vals_in, s.vals_out, (int)total, 0, end_bit, st));
        rc = compact(HeadPred{ s.keys_out }, HeadEmit{ s.seg_start }, total, 1, s.counts, s.d_total, st);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(voxel_batch_locate_kernel, dim3(1), dim3(64), 0, st, b, s.seg_start, s.d_total, s.err, s.head, d_counts);
    hipLaunchKernelGGL(voxel_batch_mean_kernel, dim3(nb), dim3(256), 0, st, b, s.vals_out, s.seg_start, s.d_total, s.head);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

// ---- transform + fuse + voxel grid in one pass (preprocessing/data.py:44-61) ----------------------------------------
// Cloud c of the frame is moved by its registration T[c] (identity for the master), the clouds are stacked and the stack is
// down-sampled.  The reference keeps the moved points as float64 arrays; storing them as float32 first would move points
// that lie within 2^-24 |x| of a voxel face into the neighbouring voxel (measured: ~1e-5 of the points,
// oracle/storage_deviation.py).  So the stack is never materialised: the bounding box, the voxel index and the per-voxel
// sums all use the fp64 value p' = AC1(T[c], p), recomputed from the float32 sensor point wherever it is needed (9 fma).
// That is also one pass less over HBM and three launches less per sensor than transform -> concat -> voxel.
constexpr int kFuseMax = 16;
constexpr int kFuseBboxBlocks = 32;
struct FuseBatch {
    const float *pts[kFuseMax];
    const float *col[kFuseMax];
    int64_t off[kFuseMax + 1];
    double T[kFuseMax][12];               // rows of [R | t]
    const double *dT[kFuseMax];           // non-null: the cloud's 4x4 (row-major, first 12 entries used) is read from device memory instead --
                                          // the frame loop hands over the registrations' results without a host round trip
    int32_t count;
};
__device__ __forceinline__ int fuse_cloud(const FuseBatch &b, int64_t i)
{
    int c = 0;
#pragma unroll
    for (int k = 1; k < kFuseMax; ++k) c += (k < b.count && i >= b.off[k]) ? 1 : 0;
    return c;
}
__device__ __forceinline__ void fuse_point(const FuseBatch &b, int c, int64_t j, double o[3])
{
    const float *p = b.pts[c] + 3 * j;
    const double x = p[0], y = p[1], z = p[2];
    const double *T = b.dT[c] ? b.dT[c] : b.T[c];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = fma(T[4 * k], x, fma(T[4 * k + 1], y, fma(T[4 * k + 2], z, T[4 * k + 3])));
}
__global__ __launch_bounds__(256) void fuse_bbox_partial_kernel(FuseBatch b, double *__restrict__ part)
{
    __shared__ double sh[6][4];
    const int c = blockIdx.y;
    const int64_t n = b.off[c + 1] - b.off[c];
    double mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double q[3];
        fuse_point(b, c, i, q);
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fmin(mn[a], q[a]); mx[a] = fmax(mx[a], q[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]); }
    if (lane_id() == 0)
        for (int a = 0; a < 3; ++a) { sh[a][wave_id()] = mn[a]; sh[3 + a][wave_id()] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = sh[threadIdx.x][0];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fmin(v, sh[threadIdx.x][w]) : fmax(v, sh[threadIdx.x][w]);
        part[((int64_t)c * kFuseBboxBlocks + blockIdx.x) * 6 + threadIdx.x] = v;
    }
}
// folds the count x kFuseBboxBlocks partial boxes (min / max: exact, order-free) -> bbox[0..5]; zeroes the error word
__global__ __launch_bounds__(64) void fuse_bbox_final_kernel(const double *__restrict__ part, int rows, double *__restrict__ bbox, int32_t *__restrict__ err)
{
    const int lane = lane_id();
    double v[6] = { INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY };
    for (int r = lane; r < rows; r += 64) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { v[a] = fmin(v[a], part[(int64_t)r * 6 + a]); v[3 + a] = fmax(v[3 + a], part[(int64_t)r * 6 + 3 + a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { v[a] = wave_min(v[a]); v[3 + a] = wave_max(v[3 + a]); }
    if (lane == 0) {
        for (int a = 0; a < 6; ++a) bbox[a] = v[a];
        *err = 0;
    }
}
__global__ __launch_bounds__(256) void fuse_key_kernel(FuseBatch b, const double *__restrict__ bbox, double voxel, uint64_t *__restrict__ keys,
                                                       int32_t *__restrict__ vals, int32_t *__restrict__ err)
{
    const double ox = bbox[0] - voxel * 0.5, oy = bbox[1] - voxel * 0.5, oz = bbox[2] - voxel * 0.5;
    const int64_t total = b.off[b.count];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = fuse_cloud(b, i);
        double q[3];
        fuse_point(b, c, i - b.off[c], q);
        double fx = floor((q[0] - ox) / voxel), fy = floor((q[1] - oy) / voxel), fz = floor((q[2] - oz) / voxel);
        const bool bad = !(fx >= 0.0) || !(fy >= 0.0) || !(fz >= 0.0) || fx >= 2097152.0 || fy >= 2097152.0 || fz >= 2097152.0;
        if (bad) { *err = 1; fx = fy = fz = 0.0; }
        keys[i] = ((uint64_t)fx << 42) | ((uint64_t)fy << 21) | (uint64_t)fz;
        vals[i] = (int32_t)i;
    }
}
// The same keys as 32-bit mixed-radix words (ix DY + iy) DZ + iz over the grid's own extent -- the SAME order (ascending ix, iy, iz)
// in as few bits as the fused cloud needs (a person at 10 mm voxels: ~23), for the library's own radix sort: three 8-bit passes of two
// launches each where the vendor's merge sort of the 63-bit keys took seven launches with host work between them (~150 us of a frame
// under load, profiles/r05/overlap_timeline_native_stream.txt).  bits_out: the width this cloud needs (> 32: the keys written here
// are useless and the caller takes the 63-bit path).
__device__ __forceinline__ int fuse_key_bits(const double *__restrict__ bbox, double voxel, double d[3], bool *sane_out)
{
    const double o[3] = { bbox[0] - voxel * 0.5, bbox[1] - voxel * 0.5, bbox[2] - voxel * 0.5 };
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = floor((bbox[3 + a] - o[a]) / voxel) + 1.0;
    const bool sane = d[0] >= 1.0 && d[1] >= 1.0 && d[2] >= 1.0 && d[0] < 2097152.0 && d[1] < 2097152.0 && d[2] < 2097152.0;      // (NaN / empty boxes: not sane)
    int bits = 64;
    if (sane && d[0] * d[1] * d[2] < 18446744073709551616.0) {
        const unsigned long long range = (unsigned long long)d[0] * (unsigned long long)d[1] * (unsigned long long)d[2];
        bits = 1;
        while (bits < 64 && (range >> bits) != 0ull) ++bits;
    }
    *sane_out = sane;
    return bits;
}
__global__ void fuse_bits_kernel(const double *__restrict__ bbox, double voxel, int32_t *__restrict__ bits_out)
{
    double d[3];
    bool sane;
    *bits_out = fuse_key_bits(bbox, voxel, d, &sane);
}
__global__ __launch_bounds__(256) void fuse_key32_kernel(FuseBatch b, const double *__restrict__ bbox, double voxel, uint32_t *__restrict__ keys,
                                                         int32_t *__restrict__ vals, int32_t *__restrict__ bits_out)
{
    const double ox = bbox[0] - voxel * 0.5, oy = bbox[1] - voxel * 0.5, oz = bbox[2] - voxel * 0.5;
    double dd[3];
    bool sane;
    const int bits = fuse_key_bits(bbox, voxel, dd, &sane);
    const double dx = dd[0], dy = dd[1], dz = dd[2];
    if (blockIdx.x == 0 && threadIdx.x == 0) *bits_out = bits;
    const unsigned long long DY = sane ? (unsigned long long)dy : 1ull, DZ = sane ? (unsigned long long)dz : 1ull;
    const int64_t total = b.off[b.count];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = fuse_cloud(b, i);
        double q[3];
        fuse_point(b, c, i - b.off[c], q);
        double fx = floor((q[0] - ox) / voxel), fy = floor((q[1] - oy) / voxel), fz = floor((q[2] - oz) / voxel);
        if (!(fx >= 0.0) || !(fy >= 0.0) || !(fz >= 0.0) || !(fx < dx) || !(fy < dy) || !(fz < dz)) fx = fy = fz = 0.0;      // (only with bits = 64: caller redoes)
        keys[i] = (uint32_t)(((unsigned long long)fx * DY + (unsigned long long)fy) * DZ + (unsigned long long)fz);
        vals[i] = (int32_t)i;
    }
}
__global__ __launch_bounds__(256) void fuse_mean_kernel(FuseBatch b, const int32_t *__restrict__ vals, const int32_t *__restrict__ seg_start,
                                                        int32_t *__restrict__ d_count, const int32_t *__restrict__ err, float *__restrict__ opts,
                                                        float *__restrict__ ocol)
{
    const int32_t m_total = *d_count;
    if (blockIdx.x == 0 && threadIdx.x == 0 && *err) *d_count = KPX_ERR_RANGE;      // as voxel_mean_kernel
    const int64_t total = b.off[b.count];
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < m_total; m += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s0 = seg_start[m], s1 = (m + 1 < m_total) ? seg_start[m + 1] : total;
        double sp[3] = { 0, 0, 0 }, sc[3] = { 0, 0, 0 };
        for (int64_t s = s0; s < s1; s += 8) {       // loads of 8 points issued together, sums sequential in ascending stacked index
            const int cnt = (int)(s1 - s < 8 ? s1 - s : 8);
            int64_t p[8];
            int cl[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                p[k] = k < cnt ? (int64_t)vals[s + k] : -1;
                cl[k] = p[k] >= 0 ? fuse_cloud(b, p[k]) : 0;
            }
            double vq[8][3];
            float vc[8][3];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (p[k] < 0) continue;
                const int64_t j = p[k] - b.off[cl[k]];
                fuse_point(b, cl[k], j, vq[k]);
                const float *col = b.col[cl[k]];
#pragma unroll
                for (int a = 0; a < 3; ++a) vc[k][a] = col ? col[3 * j + a] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (p[k] < 0) continue;
#pragma unroll
                for (int a = 0; a < 3; ++a) { sp[a] += vq[k][a]; sc[a] += (double)vc[k][a]; }
            }
        }
        const double cn = (double)(s1 - s0);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            opts[3 * m + a] = (float)(sp[a] / cn);
            if (ocol) ocol[3 * m + a] = (float)(sc[a] / cn);
        }
    }
}
struct FuseScratch {
    uint64_t *keys_in, *keys_out;
    int32_t *vals_in, *vals_out, *seg_start, *counts, *err;
    double *part, *bbox;
    char *sort_tmp;
    size_t sort_bytes;
    RadixScratch rx;            // the library's own sort (32-bit keys, total <= kRadixMaxPairs): the speculative path of the frame loop
    char *counts_end;
};
static void fuse_carve(Arena &a, int64_t total, FuseScratch *s)
{
    const size_t nn = (size_t)(total > 0 ? total : 1);
    s->keys_in = a.get<uint64_t>(nn); s->keys_out = a.get<uint64_t>(nn);
    s->vals_in = a.get<int32_t>(nn); s->vals_out = a.get<int32_t>(nn);
    s->seg_start = a.get<int32_t>(nn);
    radix_carve(a, total <= kRadixMaxPairs ? total : kRadixMaxPairs, &s->rx);
    s->counts = a.get<int32_t>((size_t)compact_ws_ints(total));          // right behind the sort's cleared histograms: one memset for both
    s->counts_end = reinterpret_cast<char *>(s->counts + (size_t)compact_ws_ints(total));
    s->err = a.get<int32_t>(1);
    s->part = a.get<double>((size_t)kFuseMax * kFuseBboxBlocks * 6);
    s->bbox = a.get<double>(8);
    s->sort_bytes = memo_bytes(6, (int64_t)nn, [&] { size_t b = 0; (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, s->keys_in, s->keys_out, s->vals_in, s->vals_out, (int)nn, 0, 63, (hipStream_t) nullptr); return b; });
    s->sort_tmp = a.get<char>(s->sort_bytes);
}
// spec_bits > 0 (frame loop): the fused cloud's keys are taken to fit `spec_bits` <= 32 bits -- the width its slot's previous frame
// needed -- and sorted as 32-bit words by the library's own radix sort; *d_bits (pinned host memory, written by the key kernel) receives
// the width this frame really needs, and the caller repeats the call with spec_bits = 0 when that is larger (the outputs of the
// speculative call are then garbage).  spec_bits = 0: 63-bit keys and the vendor sort, as exported.
static int fuse_voxel_impl(const FuseBatch &b, double voxel, float *opts, float *ocol, int32_t *d_count, Arena &a, hipStream_t st, int spec_bits = 0,
                           int32_t *d_bits = nullptr)
{
    const int64_t total = b.off[b.count];
    FuseScratch s;
    fuse_carve(a, total, &s);
    KPX_ARENA_CHECK(a);
    hipLaunchKernelGGL(fuse_bbox_partial_kernel, dim3(kFuseBboxBlocks, b.count), dim3(256), 0, st, b, s.part);
    hipLaunchKernelGGL(fuse_bbox_final_kernel, dim3(1), dim3(64), 0, st, s.part, b.count * kFuseBboxBlocks, s.bbox, s.err);
    const int nb = (int)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256));
    if (spec_bits > 0 && spec_bits <= 32 && d_bits && total <= kRadixMaxPairs) {
        uint32_t *k_in = reinterpret_cast<uint32_t *>(s.keys_in), *k_out = reinterpret_cast<uint32_t *>(s.keys_out);
        hipLaunchKernelGGL(fuse_key32_kernel, dim3(nb), dim3(256), 0, st, b, s.bbox, voxel, k_in, s.vals_in, d_bits);
        int rc = radix_sort_pairs_u32(s.rx, k_in, k_out, s.vals_in, s.vals_out, total, spec_bits, st, s.counts_end);
        if (rc) return rc;
        rc = compact(HeadPredT<uint32_t>{ k_out }, HeadEmit{ s.seg_start }, total, 1, s.counts, d_count, st, true);
        if (rc) return rc;
        hipLaunchKernelGGL(fuse_mean_kernel, dim3(nb), dim3(256), 0, st, b, s.vals_out, s.seg_start, d_count, s.err, opts, ocol);
        KPX_LAUNCH_CHECK();
        return KPX_OK;
    }
    if (d_bits) hipLaunchKernelGGL(fuse_bits_kernel, dim3(1), dim3(1), 0, st, s.bbox, voxel, d_bits);      // for the caller's next speculation
    hipLaunchKernelGGL(fuse_key_kernel, dim3(nb), dim3(256), 0, st, b, s.bbox, voxel, s.keys_in, s.vals_in, s.err);
    size_t bytes = s.sort_bytes;
    KPX_HIP(hipcub::DeviceRadixSort::SortPairs(s.sort_tmp, bytes, s.keys_in, s.keys_out, s.vals_in, s.vals_out, (int)total, 0, 63, st));
    int rc = compact(HeadPred{ s.keys_out }, HeadEmit{ s.seg_start }, total, 1, s.counts, d_count, st);
    if (rc) return rc;
    hipLaunchKernelGGL(fuse_mean_kernel, dim3(nb), dim3(256), 0, st, b, s.vals_out, s.seg_start, d_count, s.err, opts, ocol);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

}  // namespace kpx

using namespace kpx;

KPX_EXPORT size_t kpx_fuse_voxel_workspace_bytes(int64_t total)
{
    Arena a(nullptr, 0);
    FuseScratch s;
    fuse_carve(a, total, &s);
    return a.off;
}
KPX_EXPORT int kpx_fuse_voxel_downsample(int32_t count, const float *const *h_pts, const float *const *h_col, const int64_t *h_n,
                                         const double *h_T, double voxel, float *opts, float *ocol, int32_t *d_count, void *ws,
                                         size_t ws_bytes, void *stream)
{
    return kpx::fuse_voxel_downsample_dev(count, h_pts, h_col, h_n, h_T, nullptr, voxel, opts, ocol, d_count, ws, ws_bytes, stream, 0, nullptr);
}
int kpx::fuse_voxel_downsample_dev(int32_t count, const float *const *h_pts, const float *const *h_col, const int64_t *h_n, const double *h_T,
                                   const double *const *h_dT, double voxel, float *opts, float *ocol, int32_t *d_count, void *ws, size_t ws_bytes,
                                   void *stream, int spec_bits, int32_t *d_bits)
{
    KPX_REQUIRE(voxel > 0.0, "voxel_size <= 0");
    KPX_REQUIRE(count >= 1 && count <= kFuseMax, "kpx_fuse_voxel_downsample: 1 .. %d clouds", kFuseMax);
    KPX_REQUIRE(h_pts && h_n && h_T && d_count && ws, "kpx_fuse_voxel_downsample: null pointer");
    hipStream_t st = (hipStream_t)stream;
    FuseBatch b;
    b.count = count;
    b.off[0] = 0;
    bool any_col = false, all_col = true;
    for (int i = 0; i < kFuseMax; ++i) {
        const bool on = i < count;
        if (on) {
            KPX_REQUIRE(h_n[i] >= 0 && (h_n[i] == 0 || h_pts[i]), "kpx_fuse_voxel_downsample: bad cloud %d", i);
            if (h_n[i] > 0) { const bool hc = h_col && h_col[i]; any_col |= hc; all_col &= hc; }
        }
        b.pts[i] = on ? h_pts[i] : nullptr;
        b.col[i] = (on && h_col) ? h_col[i] : nullptr;
        b.off[i + 1] = b.off[i] + (on ? h_n[i] : 0);
        for (int k = 0; k < 12; ++k) b.T[i][k] = on ? h_T[16 * i + k] : 0.0;
        b.dT[i] = (on && h_dT) ? h_dT[i] : nullptr;
    }
    const int64_t total = b.off[count];
    KPX_REQUIRE(total < ((int64_t)1 << 31), "kpx_fuse_voxel_downsample: bad size");
    // [O3D] operator+= keeps colours only when both clouds have them
    KPX_REQUIRE(!ocol || !any_col || all_col, "kpx_fuse_voxel_downsample: colours on some clouds only");
    if (total == 0) { KPX_HIP(hipMemsetAsync(d_count, 0, sizeof(int32_t), st)); return KPX_OK; }
    KPX_REQUIRE(opts, "kpx_fuse_voxel_downsample: null pointer");
    Arena a(ws, ws_bytes);
    return fuse_voxel_impl(b, voxel, opts, (any_col && all_col) ? ocol : nullptr, d_count, a, st, spec_bits, d_bits);
}

KPX_EXPORT size_t kpx_voxel_workspace_bytes(int64_t n)
{
    Arena a(nullptr, 0);
    voxel_impl(nullptr, nullptr, nullptr, n, 1.0, nullptr, nullptr, nullptr, nullptr, a, nullptr);
    Arena one(nullptr, 0);
    VoxelBatchScratch s;
    voxel_batch_carve(one, n, &s);
    return a.off > one.off ? a.off : one.off;
}
KPX_EXPORT int kpx_voxel_downsample(const float *pts, const float *col, const float *nrm, int64_t n, double voxel,
                                    float *opts, float *ocol, float *onrm, int32_t *d_count, void *ws, size_t ws_bytes,
                                    void *stream)
{
    KPX_REQUIRE(voxel > 0.0, "voxel_size <= 0");                       // [O3D] raises here
    KPX_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "kpx_voxel_downsample: bad size");
    KPX_REQUIRE(d_count && ws, "kpx_voxel_downsample: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { KPX_HIP(hipMemsetAsync(d_count, 0, sizeof(int32_t), st)); return KPX_OK; }
    KPX_REQUIRE(pts && opts, "kpx_voxel_downsample: null pointer");
    Arena a(ws, ws_bytes);
    // Without normals a single cloud takes the one-pass batch form too (a batch of one): keys of <= 32 bits whenever the grid allows
    // (a 1M-point room at 10 mm needs 25), sorted by the library's own radix sort and the staged mean kernel -- the 63-bit key +
    // vendor-sort path below remains for clouds with normals.  KPX_VOXEL_SINGLE=0: A/B switch.
    static const bool single_batch = [] { const char *e = getenv("KPX_VOXEL_SINGLE"); return !(e && e[0] == '0'); }();
    if (!nrm && single_batch) {
        VoxelBatch b;
        b.count = 1;
        b.morton = 0;
        b.off[0] = 0;
        for (int i = 0; i < kVoxelBatchMax; ++i) {
            b.pts[i] = i == 0 ? pts : nullptr;
            b.col[i] = i == 0 ? col : nullptr;
            b.opts[i] = i == 0 ? opts : nullptr;
            b.ocol[i] = (i == 0 && col) ? ocol : nullptr;
            b.off[i + 1] = n;
        }
        return voxel_batch_impl(b, voxel, d_count, a, st);
    }
    return voxel_impl(pts, col, nrm, n, voxel, opts, ocol, onrm, d_count, a, st);
}

KPX_EXPORT size_t kpx_voxel_batch_workspace_bytes(int32_t count, const int64_t *h_n)
{
    if (count < 1 || !h_n) return 0;
    Arena a(nullptr, 0);
    int64_t total = 0;
    for (int i = 0; i < count; ++i) {
        voxel_impl(nullptr, nullptr, nullptr, h_n[i], 1.0, nullptr, nullptr, nullptr, nullptr, a, nullptr);
        total += h_n[i] > 0 ? h_n[i] : 0;
    }
    Arena one(nullptr, 0);
    VoxelBatchScratch s;
    voxel_batch_carve(one, total, &s);
    return a.off > one.off ? a.off : one.off;
}
KPX_EXPORT int kpx_voxel_downsample_batch(int32_t count, const float *const *h_pts, const float *const *h_col, const int64_t *h_n,
                                          double voxel, float *const *h_opts, float *const *h_ocol, int32_t *d_counts, void *ws,
                                          size_t ws_bytes, void *stream)
{
    return kpx::voxel_downsample_batch_spec(count, h_pts, h_col, h_n, voxel, h_opts, h_ocol, d_counts, ws, ws_bytes, stream, 0, nullptr, false);
}
int kpx::voxel_downsample_batch_spec(int32_t count, const float *const *h_pts, const float *const *h_col, const int64_t *h_n, double voxel,
                                     float *const *h_opts, float *const *h_ocol, int32_t *d_counts, void *ws, size_t ws_bytes, void *stream,
                                     int spec_bits, int32_t *d_bits, bool morton)
{
    KPX_REQUIRE(voxel > 0.0, "voxel_size <= 0");
    KPX_REQUIRE(count >= 1 && count <= 64 && h_pts && h_n && h_opts && d_counts && ws, "kpx_voxel_downsample_batch: bad arguments");
    for (int i = 0; i < count; ++i)
        KPX_REQUIRE(h_n[i] >= 0 && h_n[i] < ((int64_t)1 << 31) && (h_n[i] == 0 || (h_pts[i] && h_opts[i])),
                    "kpx_voxel_downsample_batch: bad cloud %d", i);
    hipStream_t st = (hipStream_t)stream;
    int64_t total = 0;
    for (int i = 0; i < count; ++i) total += h_n[i];
    // (the Z-curve order exists in the one-pass form only; the other forms keep the row-major order -- callers that asked for it only
    // lose locality, never correctness)
    if (count <= kVoxelBatchMax && total > 0 && total < ((int64_t)1 << 31)) {         // one pass over the concatenated clouds
        VoxelBatch b;
        b.count = count;
        static const int curve = [] { const char *e = getenv("KPX_VOXEL_CURVE"); return (e && e[0] == 'z') ? 1 : 2; }();      // A/B switch
        b.morton = morton ? curve : 0;
        b.off[0] = 0;
        for (int i = 0; i < kVoxelBatchMax; ++i) {
            const bool on = i < count;
            b.pts[i] = on ? h_pts[i] : nullptr;
            b.col[i] = (on && h_col) ? h_col[i] : nullptr;
            b.opts[i] = on ? h_opts[i] : nullptr;
            b.ocol[i] = (on && h_col && h_ocol) ? h_ocol[i] : nullptr;
            b.off[i + 1] = b.off[i] + (on ? h_n[i] : 0);
        }
        Arena one(ws, ws_bytes);
        return voxel_batch_impl(b, voxel, d_counts, one, st, spec_bits, d_bits);
    }
    if (d_bits) KPX_HIP(hipMemsetAsync(d_bits, 0, sizeof(int32_t), st));         // the other forms do not speculate: width 0 = "fine"
    if (count > kVoxelBatchMax && total > 0) {
        // more clouds than one pass takes: groups of kVoxelBatchMax, one concatenated pass each, one after the other on `stream`
        // (64 clouds of 1M points: 11 ms cloud by cloud on the lanes -- every cloud its own 8-pass sort -- against eight 8M-key sorts)
        // groups of kVoxelBatchMax consecutive clouds.  (Groups capped at the library's own radix sort -- 4M pairs, four 1M-point
        // clouds -- were measured SLOWER than eight clouds per group on the vendor's Onesweep: 8.4 vs 7.1 ms for 64 x 1M points; at
        // 4M pairs the own sort's four 8-bit passes cost 31 + 9 us each, and half as many groups halve the per-group launches.)
        int gstart[65], ng = 0;
        bool ok = true;
        for (int i = 0; i < count;) {
            gstart[ng++] = i;
            int64_t gt = 0;
            int gc = 0;
            while (i < count && gc < kVoxelBatchMax) { gt += h_n[i]; ++gc; ++i; }
            ok = ok && gt < ((int64_t)1 << 31);
        }
        gstart[ng] = count;
        if (ok) {
            for (int g = 0; g < ng; ++g) {
                const int g0 = gstart[g], gc = gstart[g + 1] - g0;
                int64_t gt = 0;
                for (int i = 0; i < gc; ++i) gt += h_n[g0 + i];
                if (gt == 0) { KPX_HIP(hipMemsetAsync(d_counts + g0, 0, (size_t)gc * sizeof(int32_t), st)); continue; }
                VoxelBatch b;
                b.count = gc;
                b.morton = 0;
                b.off[0] = 0;
                for (int i = 0; i < kVoxelBatchMax; ++i) {
                    const bool on = i < gc;
                    b.pts[i] = on ? h_pts[g0 + i] : nullptr;
                    b.col[i] = (on && h_col) ? h_col[g0 + i] : nullptr;
                    b.opts[i] = on ? h_opts[g0 + i] : nullptr;
                    b.ocol[i] = (on && h_col && h_ocol) ? h_ocol[g0 + i] : nullptr;
                    b.off[i + 1] = b.off[i] + (on ? h_n[g0 + i] : 0);
                }
                Arena one(ws, ws_bytes);
                const int grc = voxel_batch_impl(b, voxel, d_counts + g0, one, st);
                if (grc) return grc;
            }
            return KPX_OK;
        }
    }
    LaneSet *ln = nullptr;
    int rc = lanes_get(&ln);
    if (rc) return rc;
    const int used = count < kLaneCount ? count : kLaneCount;
    rc = lanes_fork(ln, st, used);
    if (rc) return rc;
    Arena a(ws, ws_bytes);
    for (int i = 0; i < count && !rc; ++i) {
        hipStream_t ls = ln->s[i % kLaneCount];
        if (h_n[i] == 0) { rc = hipMemsetAsync(d_counts + i, 0, sizeof(int32_t), ls) == hipSuccess ? KPX_OK : fail(KPX_ERR_HIP, "memset failed"); continue; }
        rc = voxel_impl(h_pts[i], h_col ? h_col[i] : nullptr, nullptr, h_n[i], voxel, h_opts[i], (h_col && h_ocol) ? h_ocol[i] : nullptr, nullptr,
                        d_counts + i, a, ls);
    }
    const int jrc = lanes_join(ln, st, used);
    return rc ? rc : jrc;
}
