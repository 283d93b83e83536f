// kpx_extract.hip -- extract stage: depth -> XYZ int16, exact median, mask + gate + compaction.
// Reference: preprocessing/extractor.py:68-80 (external unprojection, `.dat` contract),
// utils/io.py:15-43, preprocessing/data.py:165-178.  All HBM-streaming kernels.
#include <stdarg.h>

#include <mutex>

#include "kpx_internal.h"

namespace kpx {

thread_local char g_err[512] = "";
int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- profiler -------------------------------------------------------------------------------------
static struct {
    bool on = false;
    int stride = 1;                          // time every stride-th launch of a kernel id
    long long total[KPX_PROF_KERNELS] = {};  // launches seen while armed, timed or not
    int cap = 0, used = 0;
    hipEvent_t *e0 = nullptr, *e1 = nullptr;
    int *kid = nullptr;
    double *work = nullptr;
} g_prof;

bool prof_armed() { return g_prof.on; }
static std::mutex g_prof_mutex;         // launches may come from several host threads
ProfScope::ProfScope(int kernel_id, double w, hipStream_t stream) : slot(-1), st(stream)
{
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if ((g_prof.total[kernel_id]++ % g_prof.stride) != 0 || g_prof.used >= g_prof.cap) return;
    slot = g_prof.used++;
    g_prof.kid[slot] = kernel_id;
    g_prof.work[slot] = w;
    (void)hipEventRecord(g_prof.e0[slot], st);
}
ProfScope::~ProfScope()
{
    if (slot >= 0) (void)hipEventRecord(g_prof.e1[slot], st);
}

// ------------------------------------------------------------------------------------------------
// a1 unproject: 8 pixels per thread.  Reads 16 B of depth + 64 B of table, writes 48 B of int16 xyz.
// Arithmetic contract: float32, x = floorf(xt*d + 0.5f) with separate multiply and add.
__device__ __forceinline__ void unproject1(uint16_t d, float xt, float yt, int16_t &x, int16_t &y, int16_t &z)
{
    if (d != 0 && !__builtin_isnan(xt) && !__builtin_isnan(yt)) {
        float fd = (float)d;
        float px = __fmul_rn(xt, fd);
        px = __fadd_rn(px, 0.5f);
        float py = __fmul_rn(yt, fd);
        py = __fadd_rn(py, 0.5f);
        x = (int16_t)(int)floorf(px);
        y = (int16_t)(int)floorf(py);
        z = (int16_t)d;
    } else {
        x = 0; y = 0; z = 0;
    }
}

__global__ __launch_bounds__(256) void unproject_vec8_kernel(const uint16_t *__restrict__ depth,
                                                             const float *__restrict__ xy, int64_t n_px,
                                                             int16_t *__restrict__ xyz)
{
    const int frame = blockIdx.y;
    const int64_t groups = n_px >> 3;
    const uint16_t *dp = depth + (int64_t)frame * n_px;
    int16_t *op = xyz + (int64_t)frame * n_px * 3;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
        uint4 dv = *reinterpret_cast<const uint4 *>(dp + g * 8);
        const float4 *tp = reinterpret_cast<const float4 *>(xy + g * 16);
        float4 t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3];
        uint16_t d[8] = { (uint16_t)(dv.x & 0xffff), (uint16_t)(dv.x >> 16), (uint16_t)(dv.y & 0xffff), (uint16_t)(dv.y >> 16),
                          (uint16_t)(dv.z & 0xffff), (uint16_t)(dv.z >> 16), (uint16_t)(dv.w & 0xffff), (uint16_t)(dv.w >> 16) };
        float xt[8] = { t0.x, t0.z, t1.x, t1.z, t2.x, t2.z, t3.x, t3.z };
        float yt[8] = { t0.y, t0.w, t1.y, t1.w, t2.y, t2.w, t3.y, t3.w };
        union { int16_t s[24]; uint4 v[3]; } o;
#pragma unroll
        for (int k = 0; k < 8; ++k) unproject1(d[k], xt[k], yt[k], o.s[3 * k], o.s[3 * k + 1], o.s[3 * k + 2]);
        uint4 *dst = reinterpret_cast<uint4 *>(op + g * 24);
        dst[0] = o.v[0]; dst[1] = o.v[1]; dst[2] = o.v[2];
    }
}
// Same arithmetic; the 48 bytes a lane produces go through LDS so that each store instruction of a wave writes one
// contiguous kilobyte (lane l of round r writes bytes (64 r + l) 16 .. of the wave's 3 KiB) instead of 16-byte pieces
// 48 bytes apart.  Requires whole waves of groups: the host uses it when n_px is a multiple of 512.
// blockIdx.y owns `fpb` consecutive frames: the table entries of a lane's 8 pixels are loaded once and reused for all
// of them (the table is per camera, every frame of a batch uses the same one).
__global__ __launch_bounds__(256) void unproject_vec8_lds_kernel(const uint16_t *__restrict__ depth, const float *__restrict__ xy,
                                                                 int64_t n_px, int32_t frames, int32_t fpb, int16_t *__restrict__ xyz)
{
    __shared__ uint4 stage[4][192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f0 = blockIdx.y * fpb, f1 = min(frames, f0 + fpb);
    const int64_t groups = n_px >> 3;
    for (int64_t g0 = ((int64_t)blockIdx.x * 4 + wave) * 64; g0 < groups; g0 += (int64_t)gridDim.x * 256) {
        const int64_t g = g0 + lane;
        const float4 *tp = reinterpret_cast<const float4 *>(xy + g * 16);
        const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3];
        const float xt[8] = { t0.x, t0.z, t1.x, t1.z, t2.x, t2.z, t3.x, t3.z };
        const float yt[8] = { t0.y, t0.w, t1.y, t1.w, t2.y, t2.w, t3.y, t3.w };
        for (int f = f0; f < f1; ++f) {
            const uint4 dv = KPX_STREAM_LOAD(reinterpret_cast<const uint4 *>(depth + (int64_t)f * n_px + g * 8));
            const uint16_t d[8] = { (uint16_t)(dv.x & 0xffff), (uint16_t)(dv.x >> 16), (uint16_t)(dv.y & 0xffff), (uint16_t)(dv.y >> 16),
                                    (uint16_t)(dv.z & 0xffff), (uint16_t)(dv.z >> 16), (uint16_t)(dv.w & 0xffff), (uint16_t)(dv.w >> 16) };
            union { int16_t s[24]; uint4 v[3]; } o;
#pragma unroll
            for (int k = 0; k < 8; ++k) unproject1(d[k], xt[k], yt[k], o.s[3 * k], o.s[3 * k + 1], o.s[3 * k + 2]);
            stage[wave][3 * lane] = o.v[0]; stage[wave][3 * lane + 1] = o.v[1]; stage[wave][3 * lane + 2] = o.v[2];
            wave_lds_fence();
            uint4 *dst = reinterpret_cast<uint4 *>(xyz + ((int64_t)f * n_px) * 3 + g0 * 24);
#pragma unroll
            for (int r = 0; r < 3; ++r) KPX_STREAM_STORE(stage[wave][64 * r + lane], dst + 64 * r + lane);
            wave_lds_fence();
        }
    }
}

__global__ __launch_bounds__(256) void unproject_scalar_kernel(const uint16_t *__restrict__ depth,
                                                               const float *__restrict__ xy, int64_t n_px,
                                                               int64_t first, int16_t *__restrict__ xyz)
{
    const int frame = blockIdx.y;
    for (int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_px; i += (int64_t)gridDim.x * blockDim.x) {
        int16_t x, y, z;
        unproject1(depth[(int64_t)frame * n_px + i], xy[2 * i], xy[2 * i + 1], x, y, z);
        int16_t *o = xyz + ((int64_t)frame * n_px + i) * 3;
        o[0] = x; o[1] = y; o[2] = z;
    }
}

// ------------------------------------------------------------------------------------------------
// a4 exact median by two 8-bit radix-histogram passes (order key = v ^ 0x8000).
struct MedianSel {
    int32_t bin_a, bin_b;       // top-byte bins of the two middle ranks
    int64_t rank_a, rank_b;     // ranks inside those bins
};

// Depth values crowd into a handful of bins and neighbouring pixels share them: 64 lanes adding to one LDS word
// serialise.  Sixteen copies of every bin (lane & 15), laid out copy-minor so that equal bins of different copies fall
// into different banks, cut the serialisation 16-fold; the copies are folded before the global add.
constexpr int kHistCopies = 16;
// xy != NULL (fused depth path, stride 1): the value of pixel i is the z the unprojection would store, i.e. 0 where a
// table entry is NaN (unproject1), the raw depth otherwise.
__global__ __launch_bounds__(256) void median_hist_kernel(const int16_t *__restrict__ v, int64_t n, int64_t stride,
                                                          int64_t frame_stride, const MedianSel *__restrict__ sel,
                                                          uint32_t *__restrict__ hist /* [frames][2][256] */, int pass,
                                                          const float *__restrict__ xy, const uint8_t *__restrict__ nanmask)
{
    __shared__ uint32_t h[2][256][kHistCopies];
    const int frame = blockIdx.y;
    const int cp = threadIdx.x & (kHistCopies - 1);
    const int16_t *p = v + (int64_t)frame * frame_stride;
    int ba = 0, bb = 0;
    if (pass == 1) { ba = sel[frame].bin_a; bb = sel[frame].bin_b; }
    const int used = (pass == 1 && ba != bb) ? 2 : 1;                 // histograms in use (block-uniform)
    for (int e = threadIdx.x; e < used * 256 * kHistCopies; e += 256) (&h[0][0][0])[e] = 0;
    __syncthreads();
    // Neighbouring pixels mostly fall into the same bin: a thread adds a run of equal bins with ONE atomic.  Pass 1 feeds
    // the second histogram only when the two middle ranks lie in different top bins (median_select reads the first one
    // for both otherwise).
    const bool two = pass == 1 && ba != bb;
    int run_bin = -1;                     // pass 0: top byte; pass 1: low byte + 256 * (which histogram)
    uint32_t run_cnt = 0;
    auto flush = [&]() {
        if (run_cnt) atomicAdd(&h[run_bin >> 8][run_bin & 255][cp], run_cnt);
    };
    auto put = [&](int bin) {
        if (bin == run_bin) { ++run_cnt; return; }
        flush();
        run_bin = bin; run_cnt = 1;
    };
    auto take = [&](uint32_t raw) {
        const uint32_t key = (raw & 0xFFFFu) ^ 0x8000u;
        if (pass == 0) {
            put((int)(key >> 8));
        } else {
            const int top = (int)(key >> 8);
            if (top == ba) put((int)(key & 255u));
            else if (two && top == bb) put(256 + (int)(key & 255u));
        }
    };
    const bool vec = stride == 1 && (n % 8 == 0) && (frame_stride % 8 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)xy % 16 == 0);
    if (vec) {
        const int64_t groups = n >> 3;
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
            union { uint4 q; uint16_t s[8]; } d;
            d.q = reinterpret_cast<const uint4 *>(p)[g];
            if (xy) {                                       // one byte per 8 pixels instead of their 64 bytes of table
                const unsigned m = nanmask[g];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (m & (1u << k)) d.s[k] = 0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) take(d.s[k]);
        }
    } else if (stride == 3 && (n % 8 == 0) && (frame_stride % 8 == 0)) {
        // int16 XYZ image, z channel: 8 pixels = 48 bytes = three 16-byte loads, z at elements 3k (v points at z of pixel 0)
        const int64_t groups = n >> 3;
        const int16_t *img = p - 2;                                     // frame base (x of pixel 0)
        const bool ok = ((uintptr_t)img % 16) == 0;
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
            if (ok) {
                union { uint4 q[3]; uint16_t s[24]; } d;
                const uint4 *ip = reinterpret_cast<const uint4 *>(img + g * 24);
                d.q[0] = ip[0]; d.q[1] = ip[1]; d.q[2] = ip[2];
#pragma unroll
                for (int k = 0; k < 8; ++k) take(d.s[3 * k + 2]);
            } else {
                for (int k = 0; k < 8; ++k) take((uint32_t)(uint16_t)p[(g * 8 + k) * 3]);
            }
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            uint32_t raw = (uint32_t)(uint16_t)p[i * stride];
            if (xy && (__builtin_isnan(xy[2 * i]) || __builtin_isnan(xy[2 * i + 1]))) raw = 0;
            take(raw);
        }
    }
    flush();
    __syncthreads();
    uint32_t *g = hist + (int64_t)frame * 512;
    uint32_t s0 = 0, s1 = 0;
#pragma unroll
    for (int c = 0; c < kHistCopies; ++c) s0 += h[0][threadIdx.x][c];
    if (s0) atomicAdd(&g[threadIdx.x], s0);
    if (used == 2) {
#pragma unroll
        for (int c = 0; c < kHistCopies; ++c) s1 += h[1][threadIdx.x][c];
        if (s1) atomicAdd(&g[256 + threadIdx.x], s1);
    }
}

// bit k of nanmask[g]: a table entry of pixel 8 g + k is NaN (the unprojection stores z = 0 there)
__global__ __launch_bounds__(256) void xy_nanmask_kernel(const float *__restrict__ xy, int64_t groups, uint8_t *__restrict__ nanmask)
{
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
        union { float4 v4[4]; float f[16]; } t;
        const float4 *tp = reinterpret_cast<const float4 *>(xy + 16 * g);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) t.v4[q4] = tp[q4];
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (__builtin_isnan(t.f[2 * k]) || __builtin_isnan(t.f[2 * k + 1])) m |= 1u << k;
        nanmask[g] = (uint8_t)m;
    }
}

// one wave per frame: lane l owns bins 4 l .. 4 l + 3; a wave scan of the lane sums locates the lane, which then walks
// its four bins (a serial walk of the 256 bins by one thread took 23 us of dependent loads per pass)
__device__ __forceinline__ void rank_to_bin(const uint32_t c[4], int64_t rank, int &bin, int64_t &within, int &owner)
{
    const int s = (int)(c[0] + c[1] + c[2] + c[3]);
    const int64_t incl = wave_incl_scan(s), excl = incl - s;
    const bool mine = rank >= excl && rank < incl;
    const unsigned long long m = __ballot(mine);
    owner = m ? __builtin_ctzll(m) : -1;
    bin = 255; within = 0;
    if (mine) {
        int64_t cum = excl;
        for (int q = 0; q < 4; ++q) {
            if (rank < cum + c[q]) { bin = 4 * lane_id() + q; within = rank - cum; break; }
            cum += c[q];
        }
    }
}
__global__ __launch_bounds__(64) void median_select_kernel(uint32_t *__restrict__ hist, int64_t n, MedianSel *sel,
                                                           double *d_median, int pass)
{
    const int frame = blockIdx.x, lane = lane_id();
    uint32_t *g = hist + (int64_t)frame * 512;
    const uint4 v0 = reinterpret_cast<const uint4 *>(g)[lane];
    const uint32_t c0[4] = { v0.x, v0.y, v0.z, v0.w };
    if (pass == 0) {
        int bin_a, bin_b, own_a, own_b;
        int64_t ra, rb;
        rank_to_bin(c0, (n - 1) / 2, bin_a, ra, own_a);
        rank_to_bin(c0, n / 2, bin_b, rb, own_b);
        reinterpret_cast<uint4 *>(g)[lane] = make_uint4(0, 0, 0, 0);          // reused by pass 1
        if (own_a < 0 && lane == 0) { sel[frame].bin_a = 255; sel[frame].rank_a = 0; }
        if (own_b < 0 && lane == 0) { sel[frame].bin_b = 255; sel[frame].rank_b = 0; }
        if (lane == own_a) { sel[frame].bin_a = bin_a; sel[frame].rank_a = ra; }
        if (lane == own_b) { sel[frame].bin_b = bin_b; sel[frame].rank_b = rb; }
    } else {
        const MedianSel s = sel[frame];
        uint32_t c1[4] = { c0[0], c0[1], c0[2], c0[3] };
        if (s.bin_a != s.bin_b) {
            const uint4 v1 = reinterpret_cast<const uint4 *>(g + 256)[lane];
            c1[0] = v1.x; c1[1] = v1.y; c1[2] = v1.z; c1[3] = v1.w;
        }
        int va, vb, own_a, own_b;
        int64_t wa, wb;
        rank_to_bin(c0, s.rank_a, va, wa, own_a);
        rank_to_bin(c1, s.rank_b, vb, wb, own_b);
        va = own_a >= 0 ? __shfl(va, own_a, 64) : 0;
        vb = own_b >= 0 ? __shfl(vb, own_b, 64) : 0;
        if (lane == 0) {
            const int ka = ((s.bin_a << 8) | va) ^ 0x8000, kb = ((s.bin_b << 8) | vb) ^ 0x8000;
            const double a = (double)(int16_t)(uint16_t)ka, b = (double)(int16_t)(uint16_t)kb;
            d_median[frame] = 0.5 * (a + b);
        }
    }
}

// ---- a4 exact median of a BATCH of frames: one block per frame, the frame's histogram in LDS (round 5) ---------------------------------
// The two radix passes above read every value twice and are four launches; for a batch (one block per CU has work from 64 frames on) a
// block can hold a frame's fine histogram instead, count the frame in ONE read and pick both middle ranks itself.  A depth frame is a
// few surfaces -- in the bench's frame ten values hold two thirds of the pixels -- so a wave's LDS atomics hit few addresses and
// serialise; the histogram therefore covers the Kinect's own range, 0 .. 8191 mm, in FOUR copies (lane & 3; 4 x 32 KB), and everything
// else -- negative values, depths beyond 8 m: not a Kinect range, but the contract is np.median of any int16 -- is counted by its top
// byte only (256 coarse bins).  A middle rank that falls outside the window is found by a second read of the frame with the fine
// histogram over the 256 values of its coarse bin.  Same access forms as median_hist_kernel (8 values per 16-byte load, the z channel of
// an int16 XYZ image, the fused depth path's NaN mask), four loads of a thread in flight, runs of equal neighbours added with one atomic.
// (Tried: ONE copy of 32768 bins -- 135 us for 256 frames where the loads alone take ~40, against ~100 with the four copies; four (value,
// count) register pairs per thread in front of it -- the divergent compare chains cost more than the conflicts they avoid; the zeros,
// a fifth of a frame, counted in a register -- no change.)
constexpr int kMedWin = 8192, kMedCopies = 4, kMedFrameThreads = 1024;
template <class Take>
__device__ __forceinline__ void median_frame_scan(const int16_t *__restrict__ p, int64_t n, int64_t stride, int64_t frame_stride, bool vec, const float *__restrict__ xy,
                                                  const uint8_t *__restrict__ nanmask, Take take)
{
    const int tid = threadIdx.x;
    if (vec) {
        const int64_t groups = n >> 3;
        for (int64_t g0 = tid; g0 < groups; g0 += 4 * kMedFrameThreads) {
            union { uint4 q; uint16_t s[8]; } d[4];
            unsigned m[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t g = g0 + (int64_t)u * kMedFrameThreads;
                const bool on = g < groups;
                d[u].q = reinterpret_cast<const uint4 *>(p)[on ? g : g0];
                m[u] = (xy && on) ? nanmask[g] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (g0 + (int64_t)u * kMedFrameThreads >= groups) break;
#pragma unroll
                for (int k = 0; k < 8; ++k) take((m[u] & (1u << k)) ? 0u : (uint32_t)d[u].s[k]);
            }
        }
    } else if (stride == 3 && (n % 8 == 0) && (frame_stride % 8 == 0)) {
        const int64_t groups = n >> 3;
        const int16_t *img = p - 2;                                     // frame base (x of pixel 0)
        const bool ok = ((uintptr_t)img % 16) == 0;
        for (int64_t g0 = tid; g0 < groups; g0 += 2 * kMedFrameThreads) {
            if (ok) {                                                   // two groups = six 16-byte loads in flight
                union { uint4 q[3]; uint16_t s[24]; } d[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int64_t g = g0 + (int64_t)u * kMedFrameThreads;
                    const uint4 *ip = reinterpret_cast<const uint4 *>(img + (g < groups ? g : g0) * 24);
                    d[u].q[0] = ip[0]; d[u].q[1] = ip[1]; d[u].q[2] = ip[2];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (g0 + (int64_t)u * kMedFrameThreads >= groups) break;
#pragma unroll
                    for (int k = 0; k < 8; ++k) take(d[u].s[3 * k + 2]);
                }
            } else {
                for (int u = 0; u < 2; ++u) {
                    const int64_t g = g0 + (int64_t)u * kMedFrameThreads;
                    if (g >= groups) break;
                    for (int k = 0; k < 8; ++k) take((uint32_t)(uint16_t)p[(g * 8 + k) * 3]);
                }
            }
        }
    } else {
        for (int64_t i = tid; i < n; i += kMedFrameThreads) {
            uint32_t raw = (uint32_t)(uint16_t)p[i * stride];
            if (xy && (__builtin_isnan(xy[2 * i]) || __builtin_isnan(xy[2 * i + 1]))) raw = 0;
            take(raw);
        }
    }
}
__global__ __launch_bounds__(kMedFrameThreads) void median_frame_kernel(const int16_t *__restrict__ v, int64_t n, int64_t stride, int64_t frame_stride,
                                                                        double *__restrict__ d_median, const float *__restrict__ xy,
                                                                        const uint8_t *__restrict__ nanmask)
{
    extern __shared__ uint32_t fh[];                                // kMedCopies x kMedWin counts (copy-major)
    __shared__ uint32_t coarse[256];                                // out-of-window values by the top byte of their order key (v ^ 0x8000)
    __shared__ int64_t wsum[kMedFrameThreads / 64 + 1];
    __shared__ int s_val[2], s_bin[2];
    __shared__ int64_t s_rank[2];
    const int frame = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int16_t *p = v + (int64_t)frame * frame_stride;
    const bool vec = stride == 1 && (n % 8 == 0) && (frame_stride % 8 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)xy % 16 == 0);
    const int64_t rank[2] = { (n - 1) / 2, n / 2 };
    for (int e = tid; e < kMedCopies * kMedWin / 4; e += kMedFrameThreads) reinterpret_cast<uint4 *>(fh)[e] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 256) coarse[tid] = 0u;
    if (tid < 2) s_bin[tid] = -1;
    __syncthreads();
    {
        uint32_t *mine = fh + (size_t)(lane & (kMedCopies - 1)) * kMedWin;
        int run_bin = -1;
        uint32_t run_cnt = 0;
        median_frame_scan(p, n, stride, frame_stride, vec, xy, nanmask, [&](uint32_t raw) {
            const unsigned val = raw & 0xFFFFu;
            if (val < (unsigned)kMedWin) {
                if ((int)val == run_bin) { ++run_cnt; return; }
                if (run_cnt) atomicAdd(&mine[run_bin], run_cnt);
                run_bin = (int)val; run_cnt = 1;
            } else {
                atomicAdd(&coarse[(val ^ 0x8000u) >> 8], 1u);
            }
        });
        if (run_cnt) atomicAdd(&mine[run_bin], run_cnt);
    }
    __syncthreads();
    // order: coarse bins 0 .. 127 (negative values), the window (keys 0x8000 .. 0x9FFF = coarse bins 128 .. 159, counted fine), coarse bins
    // 160 .. 255.  Thread t owns the window's values 8 t .. 8 t + 7 (the four copies folded).
    uint32_t c[8];
    int64_t mine_sum = 0;
    {
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = 0u;
#pragma unroll
        for (int cp = 0; cp < kMedCopies; ++cp) {
            const uint4 a4 = reinterpret_cast<const uint4 *>(fh + (size_t)cp * kMedWin)[2 * tid], b4 = reinterpret_cast<const uint4 *>(fh + (size_t)cp * kMedWin)[2 * tid + 1];
            c[0] += a4.x; c[1] += a4.y; c[2] += a4.z; c[3] += a4.w; c[4] += b4.x; c[5] += b4.y; c[6] += b4.z; c[7] += b4.w;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) mine_sum += (int64_t)c[q];
    }
    int64_t incl = mine_sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int64_t t2 = __shfl_up(incl, o, 64); if (lane >= o) incl += t2; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int64_t runs = 0;
        for (int w = 0; w < kMedFrameThreads / 64; ++w) { const int64_t t2 = wsum[w]; wsum[w] = runs; runs += t2; }
        wsum[kMedFrameThreads / 64] = runs;                                 // values inside the window
        // the ranks that fall outside the window: which coarse bin, which rank inside it
        int64_t below = 0;
        for (int b = 0; b < 128; ++b) below += (int64_t)coarse[b];
        const int64_t inwin = runs;
        for (int w = 0; w < 2; ++w) {
            const int64_t r = rank[w];
            if (r < below) {
                int64_t cum = 0;
                for (int b = 0; b < 128; ++b) { if (r < cum + (int64_t)coarse[b]) { s_bin[w] = b; s_rank[w] = r - cum; break; } cum += (int64_t)coarse[b]; }
            } else if (r >= below + inwin) {
                int64_t cum = below + inwin;
                for (int b = 160; b < 256; ++b) { if (r < cum + (int64_t)coarse[b]) { s_bin[w] = b; s_rank[w] = r - cum; break; } cum += (int64_t)coarse[b]; }
            } else {
                s_rank[w] = r - below;                                      // rank inside the window
            }
        }
    }
    __syncthreads();
    const int64_t excl = wsum[wave] + incl - mine_sum;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const int64_t r = s_rank[w];
        if (s_bin[w] < 0 && r >= excl && r < excl + mine_sum) {
            int64_t cum = excl;
            int found = -1;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (found < 0 && r < cum + (int64_t)c[q]) found = 8 * tid + q;
                cum += (int64_t)c[q];
            }
            s_val[w] = found;
        }
    }
    __syncthreads();
    // the rare second read: a middle rank among the values the window does not hold -- the 256 values of its coarse bin, counted fine
    for (int w = 0; w < 2; ++w) {
        const int b = s_bin[w];                                             // block-uniform
        if (b < 0) continue;
        if (w == 1 && s_bin[0] == b && s_rank[0] == s_rank[1]) { if (tid == 0) s_val[1] = s_val[0]; __syncthreads(); continue; }
        if (tid < 256) fh[tid] = 0u;
        __syncthreads();
        median_frame_scan(p, n, stride, frame_stride, vec, xy, nanmask, [&](uint32_t raw) {
            const unsigned key = (raw & 0xFFFFu) ^ 0x8000u;
            if ((int)(key >> 8) == b) atomicAdd(&fh[key & 255u], 1u);
        });
        __syncthreads();
        if (tid == 0) {
            int64_t cum = 0;
            const int64_t r = s_rank[w];
            for (int q = 0; q < 256; ++q) {
                if (r < cum + (int64_t)fh[q]) { s_val[w] = (int)(int16_t)(uint16_t)((((unsigned)b << 8) | (unsigned)q) ^ 0x8000u); break; }
                cum += (int64_t)fh[q];
            }
        }
        __syncthreads();
    }
    if (tid == 0) d_median[frame] = 0.5 * ((double)s_val[0] + (double)s_val[1]);
}

// clear_from: start of a region in FRONT of the histogram (carved just before this call) that the one memset of the histogram
// shall cover too -- the fused depth path's tile states
static int median_impl(const int16_t *v, int64_t n, int64_t stride, int64_t frame_stride, int32_t frames,
                       double *d_median, Arena &a, hipStream_t st, const float *xy = nullptr, void *clear_from = nullptr)
{
    uint32_t *hist = a.get<uint32_t>((size_t)frames * 512);
    MedianSel *sel = a.get<MedianSel>((size_t)frames);
    // carved for the fused depth path only (its size query runs dry with the frame size; kpx_median_i16 has no table)
    uint8_t *nanmask = (xy || (a.dry && n > 0)) ? a.get<uint8_t>((size_t)(n / 8 + 1)) : nullptr;
    if (a.dry) return KPX_OK;
    KPX_ARENA_CHECK(a);
    {
        char *c0 = clear_from ? reinterpret_cast<char *>(clear_from) : reinterpret_cast<char *>(hist);
        KPX_HIP(hipMemsetAsync(c0, 0, (size_t)(reinterpret_cast<char *>(hist + (size_t)frames * 512) - c0), st));
    }
    const bool vec = stride == 1 && (n % 8 == 0) && (frame_stride % 8 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)xy % 16 == 0);
    if (xy && vec)
        hipLaunchKernelGGL(xy_nanmask_kernel, dim3((unsigned)(cdiv(n / 8, 256) > 1024 ? 1024 : cdiv(n / 8, 256))), dim3(256), 0, st, xy, n / 8, nanmask);
    // a batch: one block per frame with the frame's whole histogram in LDS -- one read of the values, one launch (KPX_MEDIAN_FRAME=0 / 1
    // forces either form; read at every call so that a test can cover both in one process)
    {
        const char *e = getenv("KPX_MEDIAN_FRAME");
        const bool by_frame = e ? e[0] == '1' : (frames >= 64 && n >= 32768);
        if (by_frame) {
            static bool attr_set = false;
            if (!attr_set) {
                KPX_HIP(hipFuncSetAttribute((const void *)median_frame_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMedCopies * kMedWin * 4));
                attr_set = true;
            }
            hipLaunchKernelGGL(median_frame_kernel, dim3(frames), dim3(kMedFrameThreads), (size_t)kMedCopies * kMedWin * 4, st, v, n, stride, frame_stride, d_median,
                               xy, (const uint8_t *)(vec ? nanmask : nullptr));
            KPX_LAUNCH_CHECK();
            return KPX_OK;
        }
    }
    // values per thread: a block pays a fixed price for zeroing and folding its LDS histogram, so batches use fewer,
    // longer blocks; a handful of frames keeps many short ones to fill the chip
    const int per_thread = frames >= 64 ? 64 : (frames >= 16 ? 32 : 16);
    int bx = (int)(cdiv(n, 256 * per_thread) < 1 ? 1 : (cdiv(n, 256 * per_thread) > 512 ? 512 : cdiv(n, 256 * per_thread)));
    dim3 grid(bx, frames);
    hipLaunchKernelGGL(median_hist_kernel, grid, dim3(256), 0, st, v, n, stride, frame_stride, sel, hist, 0, xy, nanmask);
    hipLaunchKernelGGL(median_select_kernel, dim3(frames), dim3(64), 0, st, hist, n, sel, d_median, 0);
    hipLaunchKernelGGL(median_hist_kernel, grid, dim3(256), 0, st, v, n, stride, frame_stride, sel, hist, 1, xy, nanmask);
    hipLaunchKernelGGL(median_select_kernel, dim3(frames), dim3(64), 0, st, hist, n, sel, d_median, 1);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

// ------------------------------------------------------------------------------------------------
// a3 + a4 compaction functors
struct XyzPred {
    const int16_t *xyz; const uint8_t *rgb; const double *median; int64_t n; int flags; double gate;
    __device__ bool operator()(int64_t i, int f) const
    {
        const int16_t *p = xyz + ((int64_t)f * n + i) * 3;
        int16_t x = p[0], y = p[1], z = p[2];
        bool keep = (x != 0) & (y != 0) & (z != 0);
        if ((flags & KPX_COMPACT_COLOR_MASK) && rgb) {
            const uint8_t *c = rgb + ((int64_t)f * n + i) * 3;
            keep = keep & (c[0] != 0) & (c[1] != 0) & (c[2] != 0);
        }
        if (flags & KPX_COMPACT_DEPTH_GATE) keep = keep & ((double)z <= median[f] + gate);
        return keep;
    }
};
struct XyzEmit {
    const int16_t *xyz; const uint8_t *rgb; int64_t n; float *pts; float *col; int32_t *idx;
    __device__ void operator()(int64_t i, int f, int32_t dst) const
    {
        const int16_t *p = xyz + ((int64_t)f * n + i) * 3;
        int64_t o = ((int64_t)f * n + dst) * 3;
        pts[o] = (float)p[0]; pts[o + 1] = (float)p[1]; pts[o + 2] = (float)p[2];
        if (col && rgb) {
            const uint8_t *c = rgb + ((int64_t)f * n + i) * 3;
            col[o] = (float)((double)c[0] / 255.0); col[o + 1] = (float)((double)c[1] / 255.0); col[o + 2] = (float)((double)c[2] / 255.0);
        }
        if (idx) idx[(int64_t)f * n + dst] = (int32_t)i;
    }
};

// fused depth -> cloud functors (never materialise the int16 image)
struct DepthPred {
    const uint16_t *depth; const float *xy; const uint8_t *rgb; const double *median; int64_t n; int flags; double gate;
    __device__ bool operator()(int64_t i, int f) const
    {
        int16_t x, y, z;
        unproject1(depth[(int64_t)f * n + i], xy[2 * i], xy[2 * i + 1], x, y, z);
        bool keep = (x != 0) & (y != 0) & (z != 0);
        if ((flags & KPX_COMPACT_COLOR_MASK) && rgb) {
            const uint8_t *c = rgb + ((int64_t)f * n + i) * 3;
            keep = keep & (c[0] != 0) & (c[1] != 0) & (c[2] != 0);
        }
        if (flags & KPX_COMPACT_DEPTH_GATE) keep = keep & ((double)z <= median[f] + gate);
        return keep;
    }
};
struct DepthEmit {
    const uint16_t *depth; const float *xy; const uint8_t *rgb; int64_t n; float *pts; float *col; int32_t *idx;
    __device__ void operator()(int64_t i, int f, int32_t dst) const
    {
        int16_t x, y, z;
        unproject1(depth[(int64_t)f * n + i], xy[2 * i], xy[2 * i + 1], x, y, z);
        int64_t o = ((int64_t)f * n + dst) * 3;
        pts[o] = (float)x; pts[o + 1] = (float)y; pts[o + 2] = (float)z;
        if (col && rgb) {
            const uint8_t *c = rgb + ((int64_t)f * n + i) * 3;
            col[o] = (float)((double)c[0] / 255.0); col[o + 1] = (float)((double)c[1] / 255.0); col[o + 2] = (float)((double)c[2] / 255.0);
        }
        if (idx) idx[(int64_t)f * n + dst] = (int32_t)i;
    }
};

}  // namespace kpx

using namespace kpx;

KPX_EXPORT const char *kpx_last_error(void) { return g_err; }

KPX_EXPORT int kpx_prof_begin(int32_t capacity)
{
    KPX_REQUIRE(capacity > 0 && capacity <= (1 << 20), "kpx_prof_begin: bad capacity");
    if (g_prof.cap < capacity) {
        for (int i = 0; i < g_prof.cap; ++i) { (void)hipEventDestroy(g_prof.e0[i]); (void)hipEventDestroy(g_prof.e1[i]); }
        delete[] g_prof.e0; delete[] g_prof.e1; delete[] g_prof.kid; delete[] g_prof.work;
        g_prof.e0 = new hipEvent_t[capacity]; g_prof.e1 = new hipEvent_t[capacity];
        g_prof.kid = new int[capacity]; g_prof.work = new double[capacity];
        for (int i = 0; i < capacity; ++i) { KPX_HIP(hipEventCreate(&g_prof.e0[i])); KPX_HIP(hipEventCreate(&g_prof.e1[i])); }
        g_prof.cap = capacity;
    }
    g_prof.used = 0;
    for (int k = 0; k < KPX_PROF_KERNELS; ++k) g_prof.total[k] = 0;
    g_prof.on = true;
    (void)nn_local_take_visits();
    return KPX_OK;
}
KPX_EXPORT int kpx_prof_stride(int32_t stride)
{
    KPX_REQUIRE(stride >= 1, "kpx_prof_stride: stride must be >= 1");
    g_prof.stride = stride;
    return KPX_OK;
}
KPX_EXPORT int kpx_prof_end(double *h_ms, int64_t *h_launches, double *h_work)
{
    KPX_REQUIRE(h_ms && h_launches && h_work, "kpx_prof_end: null pointer");
    g_prof.on = false;
    for (int k = 0; k < KPX_PROF_KERNELS; ++k) { h_ms[k] = 0.0; h_launches[k] = 0; h_work[k] = 0.0; }
    for (int i = 0; i < g_prof.used; ++i) {
        KPX_HIP(hipEventSynchronize(g_prof.e1[i]));
        float ms = 0.f;
        KPX_HIP(hipEventElapsedTime(&ms, g_prof.e0[i], g_prof.e1[i]));
        int k = g_prof.kid[i];
        h_ms[k] += ms; h_launches[k] += 1; h_work[k] += g_prof.work[i];
    }
    g_prof.used = 0;
    // flops actually issued (16 x 16 x 4 MAC per tile), scaled to the timed share of the launches
    h_work[KPX_PROF_NN_LOCAL] = 2048.0 * nn_local_take_visits() * (g_prof.total[KPX_PROF_NN_LOCAL] > 0
                                    ? (double)h_launches[KPX_PROF_NN_LOCAL] / (double)g_prof.total[KPX_PROF_NN_LOCAL] : 0.0);
    return KPX_OK;
}
KPX_EXPORT int kpx_version(void) { return KPX_VERSION; }

KPX_EXPORT int kpx_unproject_u16(const uint16_t *depth, const float *xy, int64_t n_px, int32_t frames, int16_t *xyz,
                                 void *stream)
{
    KPX_REQUIRE(n_px >= 0 && frames >= 0, "kpx_unproject_u16: negative size");
    if (n_px == 0 || frames == 0) return KPX_OK;
    KPX_REQUIRE(depth && xy && xyz, "kpx_unproject_u16: null pointer");
    hipStream_t st = (hipStream_t)stream;
    bool vec = (n_px % 8 == 0) && (((uintptr_t)depth | (uintptr_t)xy | (uintptr_t)xyz) % 16 == 0);
    if (vec && n_px % 512 == 0) {
        int64_t groups = n_px / 8;
        int bx = (int)(cdiv(groups, 256) > 2048 ? 2048 : cdiv(groups, 256));
        const int fpb = frames >= 64 ? 8 : (frames >= 16 ? 4 : 1);          // keep >= 8 blocks per CU in flight for small batches
        hipLaunchKernelGGL(unproject_vec8_lds_kernel, dim3(bx, (unsigned)cdiv(frames, fpb)), dim3(256), 0, st, depth, xy, n_px, frames, fpb, xyz);
    } else if (vec) {
        int64_t groups = n_px / 8;
        int bx = (int)(cdiv(groups, 256) > 2048 ? 2048 : cdiv(groups, 256));
        hipLaunchKernelGGL(unproject_vec8_kernel, dim3(bx, frames), dim3(256), 0, st, depth, xy, n_px, xyz);
    } else {
        int bx = (int)(cdiv(n_px, 256) > 4096 ? 4096 : cdiv(n_px, 256));
        hipLaunchKernelGGL(unproject_scalar_kernel, dim3(bx, frames), dim3(256), 0, st, depth, xy, n_px, (int64_t)0, xyz);
    }
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

KPX_EXPORT size_t kpx_median_workspace_bytes(int32_t frames)
{
    Arena a(nullptr, 0);
    median_impl(nullptr, 0, 1, 0, frames < 1 ? 1 : frames, nullptr, a, nullptr);
    return a.off;
}
KPX_EXPORT int kpx_median_i16(const int16_t *v, int64_t n, int64_t stride, int32_t frames, double *d_median, void *ws,
                              size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(n > 0 && stride > 0 && frames > 0, "kpx_median_i16: empty input");
    KPX_REQUIRE(v && d_median && ws, "kpx_median_i16: null pointer");
    Arena a(ws, ws_bytes);
    return median_impl(v, n, stride, n * stride, frames, d_median, a, (hipStream_t)stream);
}

KPX_EXPORT size_t kpx_compact_workspace_bytes(int64_t n, int32_t frames)
{
    Arena a(nullptr, 0);
    a.get<int32_t>((size_t)(frames < 1 ? 1 : frames) * compact_ws_ints(n));
    return a.off;
}
// ---- vectorised fused depth -> cloud (n_px % 8 == 0, 16-byte aligned frames) ---------------------------------------
// Same predicate and arithmetic as DepthPred / DepthEmit.  A thread owns 8 consecutive pixels: one 16-byte load of
// depth, four of the xy table, three 8-byte loads of rgb.  The scatter pass stages the block's kept points in LDS and
// writes them with consecutive lanes on consecutive dwords (the block's output range is contiguous).
struct Px8 {
    int16_t x[8], y[8], z[8];
    unsigned keep;                      // bit k: pixel k survives
    uint8_t c[24];
};
// xy == NULL: `depth` is an int16 XYZ image (3 values per pixel, the .dat contract) instead of u16 depth + table
__device__ __forceinline__ void load_px8(const uint16_t *__restrict__ depth, const float *__restrict__ xy, const uint8_t *__restrict__ rgb,
                                         const double *__restrict__ median, int64_t n, int f, int64_t base, int flags, double gate, Px8 &p)
{
    union { uint4 v; uint16_t s[8]; } d;
    union { float4 v[4]; float s[16]; } t;
    union { uint4 v[3]; int16_t s[24]; } im;
    if (xy) {
        d.v = *reinterpret_cast<const uint4 *>(depth + (int64_t)f * n + base);
        const float4 *tp = reinterpret_cast<const float4 *>(xy + 2 * base);
#pragma unroll
        for (int q = 0; q < 4; ++q) t.v[q] = tp[q];
    } else {
        const uint4 *ip = reinterpret_cast<const uint4 *>(reinterpret_cast<const int16_t *>(depth) + ((int64_t)f * n + base) * 3);
#pragma unroll
        for (int q = 0; q < 3; ++q) im.v[q] = ip[q];
    }
    const bool mask = (flags & KPX_COMPACT_COLOR_MASK) && rgb;
    if (rgb) {
        union { uint2 v[3]; uint8_t b[24]; } c;
        const uint2 *cp = reinterpret_cast<const uint2 *>(rgb + ((int64_t)f * n + base) * 3);
#pragma unroll
        for (int q = 0; q < 3; ++q) c.v[q] = cp[q];
#pragma unroll
        for (int q = 0; q < 24; ++q) p.c[q] = c.b[q];
    }
    const double lim = (flags & KPX_COMPACT_DEPTH_GATE) ? median[f] + gate : 0.0;
    p.keep = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (xy) unproject1(d.s[k], t.s[2 * k], t.s[2 * k + 1], p.x[k], p.y[k], p.z[k]);
        else { p.x[k] = im.s[3 * k]; p.y[k] = im.s[3 * k + 1]; p.z[k] = im.s[3 * k + 2]; }
        bool keep = (p.x[k] != 0) & (p.y[k] != 0) & (p.z[k] != 0);
        if (mask) keep = keep & (p.c[3 * k] != 0) & (p.c[3 * k + 1] != 0) & (p.c[3 * k + 2] != 0);
        if (flags & KPX_COMPACT_DEPTH_GATE) keep = keep & ((double)p.z[k] <= lim);
        p.keep |= keep ? 1u << k : 0u;
    }
}
__global__ __launch_bounds__(kCompactThreads) void depth_count_vec_kernel(const uint16_t *__restrict__ depth, const float *__restrict__ xy,
                                                                          const uint8_t *__restrict__ rgb, const double *__restrict__ median,
                                                                          int64_t n, int flags, double gate, int32_t *__restrict__ block_counts)
{
    __shared__ int sh[kCompactThreads / 64];
    const int f = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * kCompactTile + (int64_t)threadIdx.x * kCompactItems;
    int c = 0;
    if (base < n) {
        Px8 p;
        load_px8(depth, xy, rgb, median, n, f, base, flags, gate, p);
        c = __builtin_popcount(p.keep);
    }
    c = block_sum(c, sh);
    if (threadIdx.x == 0) block_counts[(int64_t)f * gridDim.x + blockIdx.x] = c;
}
template <bool COL, bool IDX>
__global__ __launch_bounds__(kCompactThreads) void depth_scatter_vec_kernel(const uint16_t *__restrict__ depth, const float *__restrict__ xy,
                                                                            const uint8_t *__restrict__ rgb, const double *__restrict__ median,
                                                                            int64_t n, int flags, double gate, const int32_t *__restrict__ block_offsets,
                                                                            float *__restrict__ pts, float *__restrict__ col, int32_t *__restrict__ idx)
{
    __shared__ int sh[kCompactThreads / 64 + 1];
    // staged as the integers they are (converted when written): 18 KB instead of 48 KB per block, 8 blocks per CU instead of 3
    __shared__ int16_t sp[kCompactTile * 3];
    __shared__ uint8_t sc[COL ? kCompactTile * 3 : 1];
    __shared__ int32_t si[IDX ? kCompactTile : 1];
    const int f = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * kCompactTile + (int64_t)threadIdx.x * kCompactItems;
    Px8 p;
    p.keep = 0;
    if (base < n) load_px8(depth, xy, rgb, median, n, f, base, flags, gate, p);
    int tot;
    int pos = block_excl_scan(__builtin_popcount(p.keep), sh, &tot);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (p.keep & (1u << k)) {
            sp[3 * pos] = p.x[k]; sp[3 * pos + 1] = p.y[k]; sp[3 * pos + 2] = p.z[k];
            if (COL) { sc[3 * pos] = p.c[3 * k]; sc[3 * pos + 1] = p.c[3 * k + 1]; sc[3 * pos + 2] = p.c[3 * k + 2]; }
            if (IDX) si[pos] = (int32_t)(base + k);
            ++pos;
        }
    __syncthreads();
    const int64_t o = (int64_t)f * n + block_offsets[(int64_t)f * gridDim.x + blockIdx.x];
    for (int e = threadIdx.x; e < tot * 3; e += kCompactThreads) {
        pts[o * 3 + e] = (float)sp[e];
        if (COL) col[o * 3 + e] = (float)((double)sc[e] / 255.0);
    }
    if (IDX)
        for (int e = threadIdx.x; e < tot; e += kCompactThreads) idx[o + e] = si[e];
}

// ONE pass over the frame: the tile's pixels are unprojected and tested once, its offset inside the frame's cloud comes from the
// decoupled look-back (kpx_common.h), the kept points leave through LDS as consecutive dwords.  Against count -> scan ->
// scatter: depth / table / colour are read once instead of twice and three launches become one (+ the clear of the tile words).
template <bool COL, bool IDX>
__global__ __launch_bounds__(kCompactThreads) void depth_onepass_vec_kernel(const uint16_t *__restrict__ depth, const float *__restrict__ xy,
                                                                            const uint8_t *__restrict__ rgb, const double *__restrict__ median,
                                                                            int64_t n, int flags, double gate, unsigned long long *__restrict__ tile_state,
                                                                            float *__restrict__ pts, float *__restrict__ col, int32_t *__restrict__ idx,
                                                                            int32_t *__restrict__ d_count, int frame_major)
{
    __shared__ int sh[kCompactThreads / 64 + 2];
    __shared__ int16_t sp[kCompactTile * 3];
    __shared__ uint8_t sc[COL ? kCompactTile * 3 : 1];
    __shared__ int32_t si[IDX ? kCompactTile : 1];
    // frame_major (batches): blockIdx.x is the FRAME, so that a frame's consecutive tiles are dispatched a whole row of frames apart and
    // a tile's predecessor has usually published its count by the time it is asked for (tile-major, every block waits for the block
    // dispatched just before it: measured 2.2 TB/s against 3.0 for count -> scan -> scatter on 256 frames)
    const int f = frame_major ? blockIdx.x : blockIdx.y;
    const unsigned tile = frame_major ? blockIdx.y : blockIdx.x, tiles = frame_major ? gridDim.y : gridDim.x;
    const int64_t base = (int64_t)tile * kCompactTile + (int64_t)threadIdx.x * kCompactItems;
    Px8 p;
    p.keep = 0;
    if (base < n) load_px8(depth, xy, rgb, median, n, f, base, flags, gate, p);
    int tot;
    int pos = block_excl_scan(__builtin_popcount(p.keep), sh, &tot);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (p.keep & (1u << k)) {
            sp[3 * pos] = p.x[k]; sp[3 * pos + 1] = p.y[k]; sp[3 * pos + 2] = p.z[k];
            if (COL) { sc[3 * pos] = p.c[3 * k]; sc[3 * pos + 1] = p.c[3 * k + 1]; sc[3 * pos + 2] = p.c[3 * k + 2]; }
            if (IDX) si[pos] = (int32_t)(base + k);
            ++pos;
        }
    __syncthreads();
    const int before = lookback_exclusive(tile_state + (int64_t)f * tiles, tile, tot, sh + kCompactThreads / 64 + 1);
    const int64_t o = (int64_t)f * n + before;
    for (int e = threadIdx.x; e < tot * 3; e += kCompactThreads) {
        pts[o * 3 + e] = (float)sp[e];
        if (COL) col[o * 3 + e] = (float)((double)sc[e] / 255.0);
    }
    if (IDX)
        for (int e = threadIdx.x; e < tot; e += kCompactThreads) idx[o + e] = si[e];
    if (tile == tiles - 1 && threadIdx.x == 0) d_count[f] = before + tot;
}

// the 8-pixel kernels (xy == NULL: int16 XYZ image input).  counts: frames * compact_ws_ints(n) ints (the 64-bit tile words)
static int px8_compact(const uint16_t *depth, const float *xy, const uint8_t *rgb, const double *med, int64_t n, int32_t frames, int32_t flags,
                       double gate, int32_t *counts, float *pts, float *col, int32_t *idx, int32_t *d_count, hipStream_t st, bool state_is_clear = false)
{
    const int32_t tiles = (int32_t)compact_tiles(n);
    const dim3 grid(tiles, frames), thr(kCompactThreads);
    const bool wc = col && rgb;
    // batches beyond the one-pass limit: the one-pass kernel FRAME-major (KPX_ONEPASS_BATCH=0: count -> scan -> scatter as before)
    static const bool batch_onepass = [] { const char *e = getenv("KPX_ONEPASS_BATCH"); return !(e && e[0] == '0'); }();
    const bool small = use_onepass((int64_t)tiles * frames);
    const bool frame_major = !small && batch_onepass && frames >= 8 && frames <= 65535;
    const bool onepass = small || frame_major;
    const dim3 ogrid = frame_major ? dim3(frames, tiles) : grid;
    unsigned long long *state = reinterpret_cast<unsigned long long *>(counts);
    if (onepass) { if (!state_is_clear) KPX_HIP(hipMemsetAsync(state, 0, (size_t)tiles * frames * sizeof(unsigned long long), st)); }
    else {
        hipLaunchKernelGGL(depth_count_vec_kernel, grid, thr, 0, st, depth, xy, rgb, med, n, flags, gate, counts);
        hipLaunchKernelGGL(compact_scan_kernel, dim3(frames), dim3(compact_scan_threads(tiles)), 0, st, counts, tiles, d_count);
    }
#define KPX_D2C(COL, IDX)                                                                                                       \
    do {                                                                                                                        \
        if (onepass)                                                                                                            \
            hipLaunchKernelGGL((depth_onepass_vec_kernel<COL, IDX>), ogrid, thr, 0, st, depth, xy, rgb, med, n, flags, gate, state, pts, col, idx, d_count, \
                               frame_major ? 1 : 0);                                                                           \
        else                                                                                                                    \
            hipLaunchKernelGGL((depth_scatter_vec_kernel<COL, IDX>), grid, thr, 0, st, depth, xy, rgb, med, n, flags, gate, counts, pts, col, idx);        \
    } while (0)
    if (wc && idx) KPX_D2C(true, true);
    else if (wc) KPX_D2C(true, false);
    else if (idx) KPX_D2C(false, true);
    else KPX_D2C(false, false);
#undef KPX_D2C
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

static int depth_to_cloud_impl(const uint16_t *depth, const float *xy, const uint8_t *rgb, int64_t n, int32_t frames,
                               int32_t flags, double gate, float *pts, float *col, int32_t *idx, int32_t *d_count,
                               Arena &a, hipStream_t st)
{
    int32_t *counts = a.get<int32_t>((size_t)frames * compact_ws_ints(n));
    double *med = a.get<double>((size_t)frames);
    int rc = KPX_OK;
    const bool gated = (flags & KPX_COMPACT_DEPTH_GATE) != 0;
    if (gated || a.dry)          // the median's histogram clear also covers the tile states carved in front of it (counts, med)
        rc = median_impl(reinterpret_cast<const int16_t *>(depth), n, 1, n, frames, med, a, st, xy, counts);
    if (a.dry || rc) return rc;
    KPX_ARENA_CHECK(a);
    DepthPred pred{ depth, xy, rgb, med, n, flags, gate };
    DepthEmit emit{ depth, xy, rgb, n, pts, col, idx };
    // algorithmic input bytes (u16 depth + rgb); outputs depend on the kept count and are added by the caller
    ProfScope prof(KPX_PROF_COMPACT, (double)frames * (double)n * (2.0 + (rgb ? 3.0 : 0.0)), st);
    const bool vec = (n % 8 == 0) && (((uintptr_t)depth | (uintptr_t)xy) % 16 == 0) && ((uintptr_t)rgb % 8 == 0);
    if (!vec) return compact(pred, emit, n, frames, counts, d_count, st, gated);
    return px8_compact(depth, xy, rgb, med, n, frames, flags, gate, counts, pts, col, idx, d_count, st, gated);
}
KPX_EXPORT int kpx_rgbd_compact(const int16_t *xyz, const uint8_t *rgb, int64_t n, int32_t frames, int32_t flags,
                                const double *d_median, double gate, float *pts, float *col, int32_t *idx,
                                int32_t *d_count, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(n >= 0 && frames > 0, "kpx_rgbd_compact: bad size");
    KPX_REQUIRE(xyz && pts && d_count && ws, "kpx_rgbd_compact: null pointer");
    KPX_REQUIRE(!(flags & KPX_COMPACT_DEPTH_GATE) || d_median, "kpx_rgbd_compact: depth gate needs d_median");
    KPX_REQUIRE(n < ((int64_t)1 << 31), "kpx_rgbd_compact: frame too large");
    Arena a(ws, ws_bytes);
    int32_t *counts = a.get<int32_t>((size_t)frames * compact_ws_ints(n));
    KPX_ARENA_CHECK(a);
    XyzPred pred{ xyz, rgb, d_median, n, flags, gate };
    XyzEmit emit{ xyz, rgb, n, pts, col, idx };
    if (n == 0) return compact(pred, emit, n, frames, counts, d_count, (hipStream_t)stream);
    const bool vec = (n % 8 == 0) && ((uintptr_t)xyz % 16 == 0) && ((uintptr_t)rgb % 8 == 0);
    if (!vec) return compact(pred, emit, n, frames, counts, d_count, (hipStream_t)stream);
    return px8_compact(reinterpret_cast<const uint16_t *>(xyz), nullptr, rgb, d_median, n, frames, flags, gate, counts, pts, col, idx, d_count,
                       (hipStream_t)stream);
}

KPX_EXPORT size_t kpx_depth_to_cloud_workspace_bytes(int64_t n_px, int32_t frames)
{
    Arena a(nullptr, 0);
    depth_to_cloud_impl(nullptr, nullptr, nullptr, n_px, frames < 1 ? 1 : frames, KPX_COMPACT_DEPTH_GATE, 0, nullptr,
                        nullptr, nullptr, nullptr, a, nullptr);
    return a.off;
}
KPX_EXPORT int kpx_depth_to_cloud(const uint16_t *depth, const float *xy, const uint8_t *rgb, int64_t n_px,
                                  int32_t frames, int32_t flags, double gate, float *pts, float *col, int32_t *idx,
                                  int32_t *d_count, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(n_px > 0 && frames > 0, "kpx_depth_to_cloud: bad size");
    KPX_REQUIRE(depth && xy && pts && d_count && ws, "kpx_depth_to_cloud: null pointer");
    KPX_REQUIRE(n_px < ((int64_t)1 << 31), "kpx_depth_to_cloud: frame too large");
    Arena a(ws, ws_bytes);
    return depth_to_cloud_impl(depth, xy, rgb, n_px, frames, flags, gate, pts, col, idx, d_count, a, (hipStream_t)stream);
}
