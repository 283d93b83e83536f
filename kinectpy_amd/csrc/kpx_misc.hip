// kpx_misc.hip -- container operations of the Open3D surface the path touches:
// transform, select_by_index, half-space select, slab split, bounding box.  HBM-streaming.
#include <mutex>
#include <unordered_map>
#include "kpx_internal.h"
#include "kpx_morton.h"
#include "kpx_radix.h"

namespace kpx {

namespace {
std::mutex g_memo_mu;
std::unordered_map<unsigned long long, size_t> g_memo;
}
bool memo_bytes_lookup(unsigned site, int64_t n, size_t *bytes)
{
    const unsigned long long key = ((unsigned long long)site << 48) ^ (unsigned long long)n;
    std::lock_guard<std::mutex> lock(g_memo_mu);
    auto it = g_memo.find(key);
    if (it == g_memo.end()) return false;
    *bytes = it->second;
    return true;
}
void memo_bytes_store(unsigned site, int64_t n, size_t bytes)
{
    const unsigned long long key = ((unsigned long long)site << 48) ^ (unsigned long long)n;
    std::lock_guard<std::mutex> lock(g_memo_mu);
    g_memo[key] = bytes;
}


struct Affine { double m[12]; };   // rows of [R | t]

// AC1: p'_k = fma(R_k0, x, fma(R_k1, y, fma(R_k2, z, t_k)))
__device__ __forceinline__ void xform3(const Affine &A, double x, double y, double z, double o[3])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = fma(A.m[4 * k], x, fma(A.m[4 * k + 1], y, fma(A.m[4 * k + 2], z, A.m[4 * k + 3])));
}

// 4 points (48 B) per lane with every global access a contiguous KiB per wave: the three 16-byte pieces a lane owns are
// 48 bytes apart, so loads and stores go through LDS (round r, lane l <-> piece 64 r + l of the wave's 3 KiB).  Covers
// the first n - n % 256 points; the tail runs in transform_kernel.
template <bool ROT_ONLY>
__global__ __launch_bounds__(256) void transform_lds_kernel(const float *__restrict__ in, int64_t n, Affine A, float *__restrict__ out)
{
    __shared__ float4 stage[4][192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t chunks = n >> 8;                                    // 256 points = 3 KiB per wave-trip
    for (int64_t c = (int64_t)blockIdx.x * 4 + wave; c < chunks; c += (int64_t)gridDim.x * 4) {
        const float4 *src = reinterpret_cast<const float4 *>(in + c * 768);
#pragma unroll
        for (int r = 0; r < 3; ++r) stage[wave][64 * r + lane] = KPX_STREAM_LOAD(src + 64 * r + lane);
        wave_lds_fence();
        const float4 a = stage[wave][3 * lane], b = stage[wave][3 * lane + 1], cc = stage[wave][3 * lane + 2];
        const float v[12] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, cc.x, cc.y, cc.z, cc.w };
        float r12[12];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const double x = v[3 * p], y = v[3 * p + 1], z = v[3 * p + 2];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double o = ROT_ONLY ? fma(A.m[4 * k], x, fma(A.m[4 * k + 1], y, A.m[4 * k + 2] * z))
                                          : fma(A.m[4 * k], x, fma(A.m[4 * k + 1], y, fma(A.m[4 * k + 2], z, A.m[4 * k + 3])));
                r12[3 * p + k] = (float)o;
            }
        }
        wave_lds_fence();
        stage[wave][3 * lane] = make_float4(r12[0], r12[1], r12[2], r12[3]);
        stage[wave][3 * lane + 1] = make_float4(r12[4], r12[5], r12[6], r12[7]);
        stage[wave][3 * lane + 2] = make_float4(r12[8], r12[9], r12[10], r12[11]);
        wave_lds_fence();
        float4 *dst = reinterpret_cast<float4 *>(out + c * 768);
#pragma unroll
        for (int r = 0; r < 3; ++r) KPX_STREAM_STORE(stage[wave][64 * r + lane], dst + 64 * r + lane);
        wave_lds_fence();
    }
}

// 4 points (48 B) per thread when aligned; rotate_only drops t and uses R_k2*z as the chain seed
template <bool ROT_ONLY>
__global__ __launch_bounds__(256) void transform_kernel(const float *__restrict__ in, int64_t n, Affine A,
                                                        float *__restrict__ out, int vec)
{
    if (vec) {
        const int64_t groups = n >> 2;
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
            const float4 *src = reinterpret_cast<const float4 *>(in + g * 12);
            float4 a = src[0], b = src[1], c = src[2];
            float v[12] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w };
            float r[12];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                double x = v[3 * p], y = v[3 * p + 1], z = v[3 * p + 2];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double o = ROT_ONLY ? fma(A.m[4 * k], x, fma(A.m[4 * k + 1], y, A.m[4 * k + 2] * z))
                                        : fma(A.m[4 * k], x, fma(A.m[4 * k + 1], y, fma(A.m[4 * k + 2], z, A.m[4 * k + 3])));
                    r[3 * p + k] = (float)o;
                }
            }
            float4 *dst = reinterpret_cast<float4 *>(out + g * 12);
            dst[0] = make_float4(r[0], r[1], r[2], r[3]);
            dst[1] = make_float4(r[4], r[5], r[6], r[7]);
            dst[2] = make_float4(r[8], r[9], r[10], r[11]);
        }
    }
    const int64_t first = vec ? (n & ~(int64_t)3) : 0;
    for (int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double o = ROT_ONLY ? fma(A.m[4 * k], x, fma(A.m[4 * k + 1], y, A.m[4 * k + 2] * z))
                                : fma(A.m[4 * k], x, fma(A.m[4 * k + 1], y, fma(A.m[4 * k + 2], z, A.m[4 * k + 3])));
            out[3 * i + k] = (float)o;
        }
    }
}

// out = x @ A + t on f64 rows of 3 (align_skeletons): o_c = (x0*A0c + x1*A1c) + x2*A2c, then + t_c
__global__ __launch_bounds__(256) void joints_affine_kernel(const double *__restrict__ x, int64_t rows, Affine A,
                                                            double *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
        double a = x[3 * i], b = x[3 * i + 1], c = x[3 * i + 2];
#pragma unroll
        for (int k = 0; k < 3; ++k) out[3 * i + k] = ((a * A.m[k] + b * A.m[3 + k]) + c * A.m[6 + k]) + A.m[9 + k];
    }
}


// ---- skeleton fusion (utils/skeleton_fusion.py:21-74) ---------------------------------------------------------------
// fused[f][j] = sum_c w_c p_c / sum_c w_c over the first three cameras, w_c = 1 / (|p_c - fused[f-1][j]|^alpha |p_c - centroid|^beta);
// the first `initial` frames are the plain mean over all cameras.  A recurrence in f: one thread per joint walks the
// frames (tiny data; the reference loops the same way).
__global__ __launch_bounds__(64) void fuse_skeletons_kernel(const double *__restrict__ sk, int32_t cams, int64_t frames, int32_t joints,
                                                            double alpha, double beta, int32_t initial, double *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= joints) return;
    const int64_t cam_stride = frames * joints * 3;
    const int64_t first = initial < frames ? initial : frames;
    for (int64_t f = 0; f < first; ++f) {
        for (int a = 0; a < 3; ++a) {
            double sum = 0.0;
            for (int c = 0; c < cams; ++c) sum += sk[c * cam_stride + (f * joints + j) * 3 + a];
            out[(f * joints + j) * 3 + a] = sum / (double)cams;
        }
    }
    for (int64_t f = first; f < frames; ++f) {
        double last[3], p[3][3], cen[3];
        for (int a = 0; a < 3; ++a) last[a] = out[((f - 1) * joints + j) * 3 + a];
        for (int c = 0; c < 3; ++c)
            for (int a = 0; a < 3; ++a) p[c][a] = sk[c * cam_stride + (f * joints + j) * 3 + a];
        for (int a = 0; a < 3; ++a) cen[a] = ((p[0][a] + p[1][a]) + p[2][a]) / 3.0;
        double w[3];
        for (int c = 0; c < 3; ++c) {
            double g2 = 0.0, d2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                const double dg = p[c][a] - last[a], dc = p[c][a] - cen[a];
                g2 += dg * dg;
                d2 += dc * dc;
            }
            w[c] = 1.0 / (pow(sqrt(g2), alpha) * pow(sqrt(d2), beta));
        }
        const double ws = (w[0] + w[1]) + w[2];
        for (int a = 0; a < 3; ++a) out[(f * joints + j) * 3 + a] = ((w[0] * p[0][a] + w[1] * p[1][a]) + w[2] * p[2][a]) / ws;
    }
}

// ---- bounding box --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bbox_partial_kernel(const float *__restrict__ pts, int64_t n, double *__restrict__ part)
{
    __shared__ float sh[6][4];
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
        part[(int64_t)blockIdx.x * 6 + threadIdx.x] = (double)v;
    }
}
// The same partial boxes from lane-contiguous 16-byte loads (pts 16-byte aligned): a wave reads 3 KiB = 256 points per chunk as three
// loads of one contiguous KiB each, two chunks in flight.  Float e of the cloud belongs to axis e mod 3, and float 4 (192 c + 64 r + lane) + j
// has e mod 3 = (r + lane + j) mod 3 -- so a lane folds into SLOT (r + j) mod 3 (known at compile time) and turns its three slots by
// lane mod 3 once at the end.  (The one-point-per-trip kernel above read 64M points at 2.9 TB/s.)
__global__ __launch_bounds__(256) void bbox_partial_vec_kernel(const float *__restrict__ pts, int64_t n, double *__restrict__ part)
{
    __shared__ float sh[6][4];
    const int lane = lane_id(), wave = wave_id();
    float smn[3] = { INFINITY, INFINITY, INFINITY }, smx[3] = { -INFINITY, -INFINITY, -INFINITY };
    const int64_t chunks = n / 256, wstride = (int64_t)gridDim.x * 4;
    const float4 *src = reinterpret_cast<const float4 *>(pts);
    for (int64_t c = (int64_t)blockIdx.x * 4 + wave; c < chunks; c += 2 * wstride) {
        const int64_t c2 = c + wstride < chunks ? c + wstride : c;          // the odd last trip folds its chunk twice
        float4 v[6];
#pragma unroll
        for (int r = 0; r < 3; ++r) { v[r] = src[c * 192 + 64 * r + lane]; v[3 + r] = src[c2 * 192 + 64 * r + lane]; }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const float e[4] = { v[r].x, v[r].y, v[r].z, v[r].w };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sl = (r % 3 + j) % 3;
                smn[sl] = fminf(smn[sl], e[j]); smx[sl] = fmaxf(smx[sl], e[j]);
            }
        }
    }
    // slot c of this lane is axis (c + lane) mod 3
    const int t = lane % 3;
    float mn[3], mx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        mn[a] = t == 0 ? smn[a] : (t == 1 ? smn[(a + 2) % 3] : smn[(a + 1) % 3]);
        mx[a] = t == 0 ? smx[a] : (t == 1 ? smx[(a + 2) % 3] : smx[(a + 1) % 3]);
    }
    if (blockIdx.x == 0) {                                                   // the points behind the last whole chunk
        const int64_t i = chunks * 256 + threadIdx.x;
        if (i < n) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { const float v = pts[3 * i + a]; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]); }
    if (lane == 0)
        for (int a = 0; a < 3; ++a) { sh[a][wave] = mn[a]; sh[3 + a][wave] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sh[threadIdx.x][0];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, sh[threadIdx.x][w]) : fmaxf(v, sh[threadIdx.x][w]);
        part[(int64_t)blockIdx.x * 6 + threadIdx.x] = (double)v;
    }
}
__global__ __launch_bounds__(256) void bbox_final_kernel(const double *__restrict__ part, int nb, double *__restrict__ bbox)
{
    // min / max are exact and order-independent: thread t folds the partial rows t, t+256, ..., then waves, then LDS
    __shared__ double sh[6][4];
    double v[6] = { INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY };
    for (int b = threadIdx.x; b < nb; b += 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { v[a] = fmin(v[a], part[(int64_t)b * 6 + a]); v[3 + a] = fmax(v[3 + a], part[(int64_t)b * 6 + 3 + a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { v[a] = wave_min(v[a]); v[3 + a] = wave_max(v[3 + a]); }
    if (lane_id() == 0)
        for (int a = 0; a < 6; ++a) sh[a][wave_id()] = v[a];
    __syncthreads();
    if (threadIdx.x < 6) {
        double r = sh[threadIdx.x][0];
        for (int w = 1; w < 4; ++w) r = threadIdx.x < 3 ? fmin(r, sh[threadIdx.x][w]) : fmax(r, sh[threadIdx.x][w]);
        bbox[threadIdx.x] = r;
    }
}
int lanes_get(LaneSet **out)
{
    // one set per host thread and device: the fork / join events belong to one call at a time, and streams belong to
    // the device that was current when they were created
    constexpr int kMaxDevices = 16;
    static thread_local LaneSet sets[kMaxDevices];
    static thread_local bool ready[kMaxDevices] = {};
    int dev = 0;
    KPX_HIP(hipGetDevice(&dev));
    KPX_REQUIRE(dev >= 0 && dev < kMaxDevices, "lanes_get: device ordinal %d out of range", dev);
    LaneSet &set = sets[dev];
    if (!ready[dev]) {
        for (int l = 0; l < kLaneCount; ++l) {
            KPX_HIP(hipStreamCreateWithFlags(&set.s[l], hipStreamNonBlocking));
            KPX_HIP(hipEventCreateWithFlags(&set.join[l], hipEventDisableTiming));
        }
        KPX_HIP(hipEventCreateWithFlags(&set.fork, hipEventDisableTiming));
        ready[dev] = true;
    }
    *out = &set;
    return KPX_OK;
}
int lanes_fork(LaneSet *l, hipStream_t caller, int used)
{
    KPX_HIP(hipEventRecord(l->fork, caller));
    for (int i = 0; i < used && i < kLaneCount; ++i) KPX_HIP(hipStreamWaitEvent(l->s[i], l->fork, 0));
    return KPX_OK;
}
int lanes_join(LaneSet *l, hipStream_t caller, int used)
{
    for (int i = 0; i < used && i < kLaneCount; ++i) {
        KPX_HIP(hipEventRecord(l->join[i], l->s[i]));
        KPX_HIP(hipStreamWaitEvent(caller, l->join[i], 0));
    }
    return KPX_OK;
}

// small clouds: one block does the whole reduction (one launch instead of two; the chains these boxes sit in are bound by
// the host's launch rate)
__global__ __launch_bounds__(1024) void bbox_small_kernel(const float *__restrict__ pts, int64_t n, double *__restrict__ bbox)
{
    __shared__ float sh[6][16];
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    // eight points per trip, their loads issued together: one block streams the cloud, so the loop is as fast as its loads in flight
    // (one point per trip took 14 us for 30k points, the latency of 30 dependent trips)
    for (int64_t i0 = threadIdx.x; i0 < n; i0 += 8 * (int64_t)blockDim.x) {
        float v[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = i0 + (int64_t)u * blockDim.x;
            const bool on = i < n;
#pragma unroll
            for (int a = 0; a < 3; ++a) v[u][a] = on ? pts[3 * i + a] : pts[3 * i0 + a];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], v[u][a]); mx[a] = fmaxf(mx[a], v[u][a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]); }
    if (lane_id() == 0)
        for (int a = 0; a < 3; ++a) { sh[a][wave_id()] = mn[a]; sh[3 + a][wave_id()] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sh[threadIdx.x][0];
        for (int w = 1; w < 16; ++w) v = threadIdx.x < 3 ? fminf(v, sh[threadIdx.x][w]) : fmaxf(v, sh[threadIdx.x][w]);
        bbox[threadIdx.x] = (double)v;
    }
}
int bbox_f32(const float *pts, int64_t n, double *d_bbox6, double *ws_partials, hipStream_t st)
{
    if (n <= 65536) {
        hipLaunchKernelGGL(bbox_small_kernel, dim3(1), dim3(1024), 0, st, pts, n, d_bbox6);
        KPX_LAUNCH_CHECK();
        return KPX_OK;
    }
    int nb = (int)(cdiv(n, 256 * 8) < 1 ? 1 : (cdiv(n, 256 * 8) > kBboxBlocks ? kBboxBlocks : cdiv(n, 256 * 8)));
    static const bool scalar_only = [] { const char *e = getenv("KPX_BBOX_VEC"); return e && e[0] == '0'; }();         // A/B switch
    if (((uintptr_t)pts % 16) == 0 && !scalar_only)
        hipLaunchKernelGGL(bbox_partial_vec_kernel, dim3(nb), dim3(256), 0, st, pts, n, ws_partials);
    else
        hipLaunchKernelGGL(bbox_partial_kernel, dim3(nb), dim3(256), 0, st, pts, n, ws_partials);
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(256), 0, st, ws_partials, nb, d_bbox6);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

// ---- select_by_index -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather3_kernel(const float *__restrict__ a0, const float *__restrict__ a1,
                                                      const float *__restrict__ a2, const int32_t *__restrict__ idx,
                                                      int64_t n_idx, float *__restrict__ o0, float *__restrict__ o1,
                                                      float *__restrict__ o2)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_idx; k += (int64_t)gridDim.x * blockDim.x) {
        int64_t s = idx[k];
        if (a0) { o0[3 * k] = a0[3 * s]; o0[3 * k + 1] = a0[3 * s + 1]; o0[3 * k + 2] = a0[3 * s + 2]; }
        if (a1) { o1[3 * k] = a1[3 * s]; o1[3 * k + 1] = a1[3 * s + 1]; o1[3 * k + 2] = a1[3 * s + 2]; }
        if (a2) { o2[3 * k] = a2[3 * s]; o2[3 * k + 1] = a2[3 * s + 1]; o2[3 * k + 2] = a2[3 * s + 2]; }
    }
}
// the gather of the first attribute (the points) that also leaves the partial boxes of what it wrote (rows of six doubles, one per block,
// folded by bbox_final_kernel): the selected cloud's bounds cost no pass of their own (kpx_select_by_index_bounds)
constexpr int kGatherBoundsBlocks = 4096;
__global__ __launch_bounds__(256) void gather3_bounds_kernel(const float *__restrict__ a0, const float *__restrict__ a1,
                                                             const float *__restrict__ a2, const int32_t *__restrict__ idx,
                                                             int64_t n_idx, float *__restrict__ o0, float *__restrict__ o1,
                                                             float *__restrict__ o2, double *__restrict__ part)
{
    __shared__ float sh[6][4];
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_idx; k += (int64_t)gridDim.x * blockDim.x) {
        int64_t s = idx[k];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = a0[3 * s + a];
            o0[3 * k + a] = v; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v);
        }
        if (a1) { o1[3 * k] = a1[3 * s]; o1[3 * k + 1] = a1[3 * s + 1]; o1[3 * k + 2] = a1[3 * s + 2]; }
        if (a2) { o2[3 * k] = a2[3 * s]; o2[3 * k + 1] = a2[3 * s + 1]; o2[3 * k + 2] = a2[3 * s + 2]; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]); }
    if (lane_id() == 0)
        for (int a = 0; a < 3; ++a) { sh[a][wave_id()] = mn[a]; sh[3 + a][wave_id()] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sh[threadIdx.x][0];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, sh[threadIdx.x][w]) : fmaxf(v, sh[threadIdx.x][w]);
        part[(int64_t)blockIdx.x * 6 + threadIdx.x] = (double)v;
    }
}
__global__ __launch_bounds__(256) void mark_kernel(const int32_t *__restrict__ idx, int64_t n_idx, int64_t n, uint8_t *__restrict__ flag)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_idx; k += (int64_t)gridDim.x * blockDim.x) {
        int64_t s = idx[k];
        if (s >= 0 && s < n) flag[s] = 1;
    }
}
struct UnmarkedPred {
    const uint8_t *flag;
    __device__ bool operator()(int64_t i, int) const { return flag[i] == 0; }
};
struct MarkedPred {
    const uint8_t *flag;
    __device__ bool operator()(int64_t i, int) const { return flag[i] != 0; }
};
struct Gather3Emit {
    const float *a0, *a1, *a2; float *o0, *o1, *o2;
    __device__ void operator()(int64_t i, int, int32_t dst) const
    {
        int64_t d = dst;
        if (a0) { o0[3 * d] = a0[3 * i]; o0[3 * d + 1] = a0[3 * i + 1]; o0[3 * d + 2] = a0[3 * i + 2]; }
        if (a1) { o1[3 * d] = a1[3 * i]; o1[3 * d + 1] = a1[3 * i + 1]; o1[3 * d + 2] = a1[3 * i + 2]; }
        if (a2) { o2[3 * d] = a2[3 * i]; o2[3 * d + 1] = a2[3 * i + 1]; o2[3 * d + 2] = a2[3 * i + 2]; }
    }
};

// ---- half-space / slab ------------------------------------------------------------------------------
struct HalfspacePred {
    const float *pts; double a, b, c, d;
    __device__ bool operator()(int64_t i, int) const
    {
        double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        double v = ((a * x + b * y) + c * z) + d;     // floor_removal.py:43, left to right
        return !(v >= 0.0);
    }
};
struct HalfspaceXYZ {
    double a, b, c, d;
    __device__ unsigned operator()(float fx, float fy, float fz) const
    {
        const double x = fx, y = fy, z = fz;
        const double v = ((a * x + b * y) + c * z) + d;     // floor_removal.py:43, left to right
        return !(v >= 0.0) ? 1u : 0u;
    }
};
struct SlabXYZ {                                             // bit 0: lower list (y >= cut), bit 1: upper list (y < cut)
    const double *bbox; double slab;
    __device__ unsigned operator()(float, float fy, float) const
    {
        const double cut = bbox[4] - slab;                  // y.max() - 200 (floor_removal.py:65-66)
        const double y = fy;
        return (y >= cut ? 1u : 0u) | (y < cut ? 2u : 0u);
    }
};
struct IndexEmit {
    int32_t *idx;
    __device__ void operator()(int64_t i, int, int32_t dst) const { idx[dst] = (int32_t)i; }
};
static Affine affine_from(const double *T)
{
    Affine A;
    for (int k = 0; k < 3; ++k) for (int c = 0; c < 4; ++c) A.m[4 * k + c] = T[4 * k + c];
    return A;
}
static int grid_for(int64_t work, int per_block, int cap = 4096)
{
    int64_t b = cdiv(work > 0 ? work : 1, per_block);
    return (int)(b > cap ? cap : b);
}

}  // namespace kpx

using namespace kpx;

KPX_EXPORT int kpx_transform(const float *pts, int64_t n, const double *h_T, float *out, void *stream)
{
    KPX_REQUIRE(n >= 0, "kpx_transform: negative size");
    if (n == 0) return KPX_OK;
    KPX_REQUIRE(pts && h_T && out, "kpx_transform: null pointer");
    int vec = (((uintptr_t)pts | (uintptr_t)out) % 16 == 0) ? 1 : 0;
    const int64_t bulk = vec ? (n & ~(int64_t)255) : 0;
    if (bulk >= 65536) {                                              // large aligned clouds: contiguous-KiB accesses
        hipLaunchKernelGGL(transform_lds_kernel<false>, dim3(grid_for(bulk / 1024, 1, 4096)), dim3(256), 0, (hipStream_t)stream, pts, bulk,
                           affine_from(h_T), out);
        if (n > bulk)
            hipLaunchKernelGGL(transform_kernel<false>, dim3(1), dim3(256), 0, (hipStream_t)stream, pts + 3 * bulk, n - bulk, affine_from(h_T),
                               out + 3 * bulk, vec);
        KPX_LAUNCH_CHECK();
        return KPX_OK;
    }
    hipLaunchKernelGGL(transform_kernel<false>, dim3(grid_for(n / 4 + 1, 256, 2048)), dim3(256), 0, (hipStream_t)stream, pts, n,
                       affine_from(h_T), out, vec);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}
KPX_EXPORT int kpx_rotate(const float *nrm, int64_t n, const double *h_T, float *out, void *stream)
{
    KPX_REQUIRE(n >= 0, "kpx_rotate: negative size");
    if (n == 0) return KPX_OK;
    KPX_REQUIRE(nrm && h_T && out, "kpx_rotate: null pointer");
    int vec = (((uintptr_t)nrm | (uintptr_t)out) % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL(transform_kernel<true>, dim3(grid_for(n / 4 + 1, 256, 2048)), dim3(256), 0, (hipStream_t)stream, nrm, n,
                       affine_from(h_T), out, vec);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}
KPX_EXPORT int kpx_joints_affine_f64(const double *x, int64_t rows, const double *h_A, const double *h_t, double *out,
                                     void *stream)
{
    KPX_REQUIRE(rows >= 0, "kpx_joints_affine_f64: negative size");
    if (rows == 0) return KPX_OK;
    KPX_REQUIRE(x && h_A && h_t && out, "kpx_joints_affine_f64: null pointer");
    Affine A;
    for (int k = 0; k < 9; ++k) A.m[k] = h_A[k];
    for (int k = 0; k < 3; ++k) A.m[9 + k] = h_t[k];
    hipLaunchKernelGGL(joints_affine_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, (hipStream_t)stream, x, rows, A, out);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

KPX_EXPORT int kpx_fuse_skeletons(const double *skeletons, int32_t cams, int64_t frames, int32_t joints, double alpha, double beta,
                                  int32_t initial_frame, double *out, void *stream)
{
    KPX_REQUIRE(cams >= 3 && frames >= 0 && joints >= 0 && initial_frame >= 1, "kpx_fuse_skeletons: needs >= 3 cameras, initial_frame >= 1");
    if (frames == 0 || joints == 0) return KPX_OK;
    KPX_REQUIRE(skeletons && out, "kpx_fuse_skeletons: null pointer");
    hipLaunchKernelGGL(fuse_skeletons_kernel, dim3((unsigned)cdiv(joints, 64)), dim3(64), 0, (hipStream_t)stream, skeletons, cams, frames, joints,
                       alpha, beta, initial_frame, out);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

// ---- stable sort of (uint32 key, int32 value) pairs on the low end_bit key bits: the library's own radix sort up to
// kRadixMaxPairs pairs, rocPRIM above (exported for tests and for callers that order their own index lists)
static void sort_u32_carve(Arena &a, int64_t n, RadixScratch *rx, char **tmp, size_t *tmp_bytes)
{
    if (n <= kRadixMaxPairs) { radix_carve(a, n, rx); return; }
    *tmp_bytes = memo_bytes(1, n, [&] { size_t b = 0; (void)sort_pairs<uint32_t>(nullptr, b, nullptr, nullptr, nullptr, nullptr, n, 32, (hipStream_t) nullptr); return b; });
    *tmp = a.get<char>(*tmp_bytes);
}
KPX_EXPORT size_t kpx_sort_pairs_u32_workspace_bytes(int64_t n)
{
    Arena a(nullptr, 0);
    RadixScratch rx;
    char *tmp = nullptr;
    size_t tb = 0;
    sort_u32_carve(a, n, &rx, &tmp, &tb);
    return a.off;
}
KPX_EXPORT int kpx_sort_pairs_u32(const uint32_t *keys_in, const int32_t *vals_in, int64_t n, int32_t end_bit, uint32_t *keys_out,
                                  int32_t *vals_out, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && end_bit >= 1 && end_bit <= 32, "kpx_sort_pairs_u32: bad size or bit count");
    if (n == 0) return KPX_OK;
    KPX_REQUIRE(keys_in && vals_in && keys_out && vals_out && ws, "kpx_sort_pairs_u32: null pointer");
    KPX_REQUIRE(keys_in != keys_out && vals_in != vals_out, "kpx_sort_pairs_u32: outputs may not alias the inputs");
    Arena a(ws, ws_bytes);
    RadixScratch rx;
    char *tmp = nullptr;
    size_t tb = 0;
    sort_u32_carve(a, n, &rx, &tmp, &tb);
    KPX_ARENA_CHECK(a);
    if (n <= kRadixMaxPairs) return radix_sort_pairs_u32(rx, keys_in, keys_out, vals_in, vals_out, n, end_bit, (hipStream_t)stream);
    KPX_HIP(sort_pairs<uint32_t>(tmp, tb, keys_in, keys_out, vals_in, vals_out, n, end_bit, (hipStream_t)stream));
    return KPX_OK;
}

KPX_EXPORT size_t kpx_select_workspace_bytes(int64_t n)
{
    Arena a(nullptr, 0);
    a.get<uint8_t>((size_t)(n > 0 ? n : 1));
    a.get<int32_t>((size_t)compact_ws_ints(n));
    a.get<double>((size_t)kBboxBlocks * 6 + 8);
    CompactPtsScratch cs;
    compact_pts_carve(a, n, &cs);
    a.get<double>((size_t)kGatherBoundsBlocks * 6);              // kpx_select_by_index_bounds' partial boxes
    return a.off;
}
KPX_EXPORT size_t kpx_bounds_workspace_bytes(void)
{
    Arena a(nullptr, 0);
    a.get<double>((size_t)kBboxBlocks * 6);
    return a.off;
}
KPX_EXPORT int kpx_bounds(const float *pts, int64_t n, double *d_bbox6, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(n > 0 && pts && d_bbox6 && ws, "kpx_bounds: bad arguments");
    Arena a(ws, ws_bytes);
    double *part = a.get<double>((size_t)kBboxBlocks * 6);
    KPX_ARENA_CHECK(a);
    return bbox_f32(pts, n, d_bbox6, part, (hipStream_t)stream);
}
KPX_EXPORT int kpx_select_by_index_bounds(const float *a0, const float *a1, const float *a2, int64_t n, const int32_t *idx, int64_t n_idx,
                                          float *o0, float *o1, float *o2, double *d_bbox6, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(n >= 0 && n_idx > 0, "kpx_select_by_index_bounds: empty selection (an empty cloud has no bounds)");
    KPX_REQUIRE(idx && a0 && o0 && d_bbox6 && ws, "kpx_select_by_index_bounds: null pointer");
    hipStream_t st = (hipStream_t)stream;
    Arena a(ws, ws_bytes);
    a.get<uint8_t>((size_t)(n > 0 ? n : 1));
    a.get<int32_t>((size_t)compact_ws_ints(n));
    a.get<double>((size_t)kBboxBlocks * 6 + 8);
    CompactPtsScratch cs;
    compact_pts_carve(a, n, &cs);
    double *part = a.get<double>((size_t)kGatherBoundsBlocks * 6);
    KPX_ARENA_CHECK(a);
    const int nb = grid_for(n_idx, 256, kGatherBoundsBlocks);
    hipLaunchKernelGGL(gather3_bounds_kernel, dim3(nb), dim3(256), 0, st, a0, a1, a2, idx, n_idx, o0, o1, o2, part);
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(256), 0, st, part, nb, d_bbox6);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}
KPX_EXPORT int kpx_select_by_index(const float *a0, const float *a1, const float *a2, int64_t n, const int32_t *idx,
                                   int64_t n_idx, int32_t invert, float *o0, float *o1, float *o2, int32_t *d_count,
                                   void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(n >= 0 && n_idx >= 0, "kpx_select_by_index: negative size");
    KPX_REQUIRE(n_idx == 0 || idx, "kpx_select_by_index: null index array");
    hipStream_t st = (hipStream_t)stream;
    if (!invert) {
        if (n_idx)
            hipLaunchKernelGGL(gather3_kernel, dim3(grid_for(n_idx, 256)), dim3(256), 0, st, a0, a1, a2, idx, n_idx, o0, o1, o2);
        KPX_LAUNCH_CHECK();
        return KPX_OK;
    }
    KPX_REQUIRE(invert == KPX_SELECT_INVERT || invert == KPX_SELECT_MASK, "kpx_select_by_index: unknown mode %d", invert);
    KPX_REQUIRE(ws && d_count, "kpx_select_by_index: the mask modes need workspace and d_count");
    Arena a(ws, ws_bytes);
    uint8_t *flag = a.get<uint8_t>((size_t)(n > 0 ? n : 1));
    int32_t *counts = a.get<int32_t>((size_t)compact_ws_ints(n));
    KPX_ARENA_CHECK(a);
    KPX_HIP(hipMemsetAsync(flag, 0, (size_t)(n > 0 ? n : 1), st));
    if (n_idx) hipLaunchKernelGGL(mark_kernel, dim3(grid_for(n_idx, 256)), dim3(256), 0, st, idx, n_idx, n, flag);
    if (invert == KPX_SELECT_MASK) return compact(MarkedPred{ flag }, Gather3Emit{ a0, a1, a2, o0, o1, o2 }, n, 1, counts, d_count, st);
    return compact(UnmarkedPred{ flag }, Gather3Emit{ a0, a1, a2, o0, o1, o2 }, n, 1, counts, d_count, st);
}

KPX_EXPORT int kpx_halfspace_select(const float *pts, int64_t n, const double *h_plane, int32_t *idx, int32_t *d_count,
                                    void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(n >= 0 && h_plane && idx && d_count && ws, "kpx_halfspace_select: bad arguments");
    Arena a(ws, ws_bytes);
    a.get<uint8_t>((size_t)(n > 0 ? n : 1));
    int32_t *counts = a.get<int32_t>((size_t)compact_ws_ints(n));
    a.get<double>((size_t)kBboxBlocks * 6 + 8);
    CompactPtsScratch cs;
    compact_pts_carve(a, n, &cs);
    KPX_ARENA_CHECK(a);
    if (n == 0) return compact(HalfspacePred{ pts, h_plane[0], h_plane[1], h_plane[2], h_plane[3] }, IndexEmit{ idx }, n, 1, counts, d_count,
                               (hipStream_t)stream);
    return compact_points(pts, n, HalfspaceXYZ{ h_plane[0], h_plane[1], h_plane[2], h_plane[3] }, idx, d_count, nullptr, nullptr, cs,
                          (hipStream_t)stream);
}

// d_bounds != NULL: the cloud's bounds are known (kpx_bounds, kpx_select_by_index_bounds): no pass over the points for max(y)
static int slab_split_impl(const float *pts, int64_t n, double slab, const double *d_bounds, int32_t *lower_idx, int32_t *d_lower,
                           int32_t *upper_idx, int32_t *d_upper, void *ws, size_t ws_bytes, hipStream_t st)
{
    Arena a(ws, ws_bytes);
    a.get<uint8_t>((size_t)n);
    a.get<int32_t>((size_t)compact_ws_ints(n));
    double *part = a.get<double>((size_t)kBboxBlocks * 6 + 8);
    CompactPtsScratch cs;
    compact_pts_carve(a, n, &cs);
    KPX_ARENA_CHECK(a);
    double *bbox = part + (size_t)kBboxBlocks * 6;
    if (!d_bounds) {
        int rc = bbox_f32(pts, n, bbox, part, st);
        if (rc) return rc;
    }
    return compact_points(pts, n, SlabXYZ{ d_bounds ? d_bounds : bbox, slab }, lower_idx, d_lower, upper_idx, d_upper, cs, st);      // one pass feeds both lists
}
KPX_EXPORT int kpx_slab_split(const float *pts, int64_t n, double slab, int32_t *lower_idx, int32_t *d_lower,
                              int32_t *upper_idx, int32_t *d_upper, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(n > 0 && pts && lower_idx && upper_idx && d_lower && d_upper && ws, "kpx_slab_split: bad arguments");
    return slab_split_impl(pts, n, slab, nullptr, lower_idx, d_lower, upper_idx, d_upper, ws, ws_bytes, (hipStream_t)stream);
}
KPX_EXPORT int kpx_slab_split_bounded(const float *pts, int64_t n, double slab, const double *d_bbox6, int32_t *lower_idx, int32_t *d_lower,
                                      int32_t *upper_idx, int32_t *d_upper, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(n > 0 && pts && d_bbox6 && lower_idx && upper_idx && d_lower && d_upper && ws, "kpx_slab_split_bounded: bad arguments");
    return slab_split_impl(pts, n, slab, d_bbox6, lower_idx, d_lower, upper_idx, d_upper, ws, ws_bytes, (hipStream_t)stream);
}
