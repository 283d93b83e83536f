// kpx_frame.hip -- one frame set of the reference's frame loop as ONE native call (preprocessing/data.py:35-61 with the
// registration of data.py:127-161 folded in, the unit bench.py measures): host orchestration in C++ over the library's own
// C ABI.  The Python mirror of the same loop (kinectpy_amd/pipeline.py) issues ~12 library calls per frame with interpreter
// work, tensor allocations and the GIL in between; with several frames in flight on host threads that interpreter time is
// serialised.  Here a frame is a single call with the GIL released throughout; everything lives in the caller's workspace.
#include <chrono>
#include <vector>

#include "kpx_internal.h"

namespace kpx {

// measurement hook: KPX_FRAME_EXTRA_DISPATCHES=n queues n empty kernels behind the extraction of every frame (how much of the frame
// rate is the dispatch count itself?)
__global__ void frame_empty_kernel() {}

// A frame's read-backs without the runtime's wait: a one-thread kernel behind everything queued so far stores a sequence number in
// pinned host memory (system scope) and the host thread spins on that word.  hipStreamSynchronize costs 40-80 us per call with four
// frame threads inside the runtime at once (profiles/r05/overlap_timeline_*.txt: ~7 idle gaps of ~80 us per frame and stream, every
// one behind a read-back); the word arrives a few microseconds after the kernel has run.  Stream order makes everything queued before
// the flag kernel (kernels' stores to pinned memory, device-to-host copies) complete, and visible, before its store.  A fault or a
// hang still surfaces: the wait falls back to hipStreamQuery every millisecond and to the runtime's error.  KPX_FRAME_SPIN=0: A/B switch.
__global__ void frame_flag_kernel(unsigned long long *flag, unsigned long long seq)
{
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static int frame_wait(hipStream_t st)
{
    static const bool spin = [] { const char *e = getenv("KPX_FRAME_SPIN"); return !(e && e[0] == '0'); }();
    if (!spin) {
        KPX_HIP(hipStreamSynchronize(st));
        return KPX_OK;
    }
    static thread_local unsigned long long *flag = nullptr;
    static thread_local unsigned long long seq = 0;
    if (!flag) {
        KPX_HIP(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
        *flag = 0ull;
    }
    ++seq;
    hipLaunchKernelGGL(frame_flag_kernel, dim3(1), dim3(1), 0, st, flag, seq);
    KPX_LAUNCH_CHECK();
    auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return KPX_OK;
        __builtin_ia32_pause();
        if ((spins & 4095u) == 4095u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(1)) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) {                          // drained: the word is there (or the launch itself was lost)
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return KPX_OK;
                KPX_HIP(hipStreamSynchronize(st));
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return KPX_OK;
                return fail(KPX_ERR_HIP, "kpx_frame_step: the frame's completion word never arrived");
            }
            if (q != hipErrorNotReady) return fail(KPX_ERR_HIP, "kpx_frame_step: %s while waiting for a read-back", hipGetErrorString(q));
            t0 = std::chrono::steady_clock::now();
        }
    }
}

struct FrameLayout {
    float *full_pts, *mask_pts, *mask_col, *down_pts, *normals, *vox_pts, *vox_col;
    int32_t *vox_cnt, *keep_idx;
    double *icp_res, *sor_stats;
    void *op_ws;
    size_t op_bytes;
};
static size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }
static void frame_carve(Arena &a, int32_t S, int64_t n_px, FrameLayout *L)
{
    const size_t px = (size_t)n_px, all = (size_t)S * px;
    L->full_pts = a.get<float>(all * 3);
    L->mask_pts = a.get<float>(all * 3);
    L->mask_col = a.get<float>(all * 3);
    L->down_pts = a.get<float>(all * 3);
    L->normals = a.get<float>(px * 3);
    L->vox_pts = a.get<float>(all * 3);
    L->vox_col = a.get<float>(all * 3);
    L->keep_idx = a.get<int32_t>(all);
    L->icp_res = a.get<double>((size_t)S * 20 + 1);              // + one slot for the fused cloud's count: both go home in ONE copy
    L->vox_cnt = reinterpret_cast<int32_t *>(L->icp_res + (size_t)S * 20);
    L->sor_stats = a.get<double>(4);
    // one scratch region for whichever operator runs (they run one after the other), sized for the worst case of each
    std::vector<int64_t> worst((size_t)S, n_px);
    size_t w = kpx_depth_to_cloud_workspace_bytes(n_px, S);
    w = max_sz(w, kpx_voxel_batch_workspace_bytes(S, worst.data()));
    w = max_sz(w, kpx_normals_workspace_bytes(n_px, KPX_NORMALS_LDS_NN));
    w = max_sz(w, kpx_icp_batch_workspace_bytes(S > 1 ? S - 1 : 1, worst.data(), n_px));
    w = max_sz(w, kpx_fuse_voxel_workspace_bytes((int64_t)all));
    w = max_sz(w, kpx_sor_workspace_bytes((int64_t)all, KPX_SOR_LDS_K));
    w = max_sz(w, kpx_select_workspace_bytes((int64_t)all));
    L->op_bytes = w;
    L->op_ws = a.get<char>(w);
}

}  // namespace kpx

using namespace kpx;

KPX_EXPORT size_t kpx_frame_step_workspace_bytes(int32_t sensors, int64_t n_px)
{
    if (sensors < 1 || n_px < 1) return 0;
    Arena a(nullptr, 0);
    FrameLayout L;
    frame_carve(a, sensors, n_px, &L);
    return a.off;
}

#define KPX_SUB(call)                    \
    do {                                 \
        const int rc__ = (call);         \
        if (rc__) return rc__;           \
    } while (0)

KPX_EXPORT int kpx_frame_step(const uint16_t *depth, const uint8_t *rgb, const float *xy_table, int64_t n_px, int32_t sensors,
                              const double *h_init, const kpx_frame_params *prm, float *out_pts, float *out_col, int32_t *h_count,
                              double *h_T, int32_t *h_info, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(sensors >= 1 && sensors <= 16 && n_px > 0 && n_px < ((int64_t)1 << 31) / 16, "kpx_frame_step: 1 .. 16 sensors of at most 2^27 pixels");
    KPX_REQUIRE(depth && rgb && xy_table && prm && out_pts && out_col && h_count && h_T && ws, "kpx_frame_step: null pointer");
    KPX_REQUIRE(sensors == 1 || h_init, "kpx_frame_step: initial transforms missing");
    KPX_REQUIRE(prm->icp_mode == KPX_ICP_POINT_TO_POINT || prm->icp_mode == KPX_ICP_POINT_TO_PLANE || prm->icp_mode == KPX_ICP_FIXED, "kpx_frame_step: bad icp_mode");
    // KPX_ICP_FIXED: the reference's loop after its first frame (preprocessing/data.py:35-41 registers `if i == 0`, every later frame reuses
    // registration_transformations): no registration in the frame, h_init ARE the transforms
    const bool fixed = prm->icp_mode == KPX_ICP_FIXED;
    KPX_REQUIRE(prm->filt_k <= KPX_SOR_LDS_K && prm->normals_nn <= KPX_NORMALS_LDS_NN, "kpx_frame_step: filt_k <= %d and normals_nn <= %d (the frame's scratch is "
                "sized for the LDS forms; kpx_sor / kpx_estimate_normals take larger values)", KPX_SOR_LDS_K, KPX_NORMALS_LDS_NN);
    const int S = sensors;
    hipStream_t st = (hipStream_t)stream;
    BusyScope busy;                                        // (a frame in flight: see kpx_internal.h)
    // (the layout is a function of (workspace, sensors, pixels): carved once per thread and slot -- the carve walks every operator's
    // workspace query, ~100 us of host time in front of a frame's first kernel)
    static thread_local struct { void *ws; size_t bytes; int S; int64_t n_px; FrameLayout L; size_t off; } cache = { nullptr, 0, 0, 0, {}, 0 };
    FrameLayout L;
    if (cache.ws == ws && cache.bytes == ws_bytes && cache.S == S && cache.n_px == n_px) {
        L = cache.L;
    } else {
        Arena a(ws, ws_bytes);
        frame_carve(a, S, n_px, &L);
        KPX_ARENA_CHECK(a);
        cache.ws = ws; cache.bytes = ws_bytes; cache.S = S; cache.n_px = n_px; cache.L = L; cache.off = a.off;
    }
    // pinned read-back area of the calling thread: counts and ICP results.  ONE allocation (the doubles first), published only when
    // it succeeded; it lives as long as the thread's runtime context (a few KiB per host thread that ever ran a frame).
    static thread_local double *h_d = nullptr;
    static thread_local int32_t *h_i = nullptr;
    if (!h_i) {
        void *blk = nullptr;
        KPX_HIP(hipHostMalloc(&blk, (16 * 20 + 1) * sizeof(double) + 256 * sizeof(int32_t), hipHostMallocDefault));
        h_d = static_cast<double *>(blk);
        h_i = reinterpret_cast<int32_t *>(h_d + (16 * 20 + 1));
    }
    auto negative = [&](const int32_t *c, int n) { for (int i = 0; i < n; ++i) if (c[i] < 0) return c[i]; return 0; };

    // extract: the registration input (every valid pixel) and the person clouds (mask + depth gate + colours); both are queued
    // before the first count is read
    // Counts that only the HOST reads next are written by the kernels straight into the thread's pinned area (device-visible host
    // memory): every D2H copy of a few bytes is a dispatch of its own (~5 us) in front of the read-back it serves.  Counts that
    // later kernels read (the fused cloud's, the registrations' results) stay in device memory and are copied.
    if (!fixed) KPX_SUB(kpx_depth_to_cloud(depth, xy_table, nullptr, n_px, S, 0, prm->gate, L.full_pts, nullptr, nullptr, h_i, L.op_ws, L.op_bytes, st));
    else for (int i = 0; i < S; ++i) h_i[i] = 0;
    // (every operator runs on `st`: stream order alone makes the shared scratch region safe)
    KPX_SUB(kpx_depth_to_cloud(depth, xy_table, rgb, n_px, S, KPX_COMPACT_COLOR_MASK | KPX_COMPACT_DEPTH_GATE, prm->gate, L.mask_pts, L.mask_col, nullptr,
                               h_i + 16, L.op_ws, L.op_bytes, st));
    static const int extra_dispatches = [] { const char *e = getenv("KPX_FRAME_EXTRA_DISPATCHES"); return e ? atoi(e) : 0; }();
    for (int e = 0; e < extra_dispatches; ++e) hipLaunchKernelGGL(frame_empty_kernel, dim3(1), dim3(64), 0, st);
    std::vector<int64_t> fk((size_t)S), mk((size_t)S), dk((size_t)S);
    // registration: voxel_down_sample(reg_voxel) of every sensor's cloud, normals of the master's, point-to-plane ICP of every sub
    std::vector<const float *> p_in((size_t)S), c_in((size_t)S);
    std::vector<float *> p_out((size_t)S);
    for (int i = 0; i < S; ++i) { p_in[(size_t)i] = L.full_pts + (size_t)i * n_px * 3; p_out[(size_t)i] = L.down_pts + (size_t)i * n_px * 3; }
    KPX_SUB(frame_wait(st));                     // read-back 1: both extractions' counts
    for (int i = 0; i < S; ++i) { fk[(size_t)i] = h_i[i]; mk[(size_t)i] = h_i[16 + i]; }
    if (negative(h_i, S) || negative(h_i + 16, S)) return fail(KPX_ERR_RANGE, "kpx_frame_step: extraction reported %d", negative(h_i, S) | negative(h_i + 16, S));
    // The sort-key width of the registration voxel grid is speculated from the last frame of this thread (the scene's extent in
    // voxels barely changes from frame to frame): its read-back inside the call is one host round trip less; a frame that needs
    // more bits is detected with the counts and done again the careful way.
    // The registration clouds leave the voxel grid along the Z-curve of their voxel indices (one point per voxel: the order the
    // culled search would otherwise establish with a Morton sort of its own -- 12 dispatches per frame); nothing downstream depends
    // on their order: normals and nearest neighbours are per point, the update sums are exact.  KPX_FRAME_ZORDER=0: A/B switch.
    static const bool zorder_on = [] { const char *e = getenv("KPX_FRAME_ZORDER"); return !(e && e[0] == '0'); }();
    const bool zorder = zorder_on && S <= 8;         // the voxel batch's one-pass form (the only one with the Z-curve order) takes 8 clouds
    static thread_local int spec_bits = 0;
    static const bool speculate = [] { const char *e = getenv("KPX_FRAME_SPECULATE"); return !(e && e[0] == '0'); }();      // A/B switch
    if (!speculate) spec_bits = 0;
    for (int i = 0; i < S; ++i) h_i[32 + i] = 0;
    for (int attempt = 0; attempt < 2 && !fixed; ++attempt) {
        KPX_SUB(voxel_downsample_batch_spec(S, p_in.data(), nullptr, fk.data(), prm->reg_voxel, p_out.data(), nullptr, h_i + 32, L.op_ws, L.op_bytes, st,
                                            attempt == 0 ? spec_bits : 0, h_i + 50, zorder));
        KPX_SUB(frame_wait(st));
        const int need = h_i[50];
        const bool narrow = attempt == 0 && spec_bits > 0 && need > spec_bits;
        spec_bits = need > 0 && need <= 32 ? (need + 7) / 8 * 8 : 0;      // whole 8-bit passes; wide keys are not speculated
        if (!narrow) break;
    }
    if (negative(h_i + 32, S)) return fail(KPX_ERR_RANGE, "voxel_size is too small");
    for (int i = 0; i < S; ++i) dk[(size_t)i] = h_i[32 + i];
    for (int q = 0; q < 16; ++q) h_T[q] = (q % 5 == 0) ? 1.0 : 0.0;
    if (h_info) for (int i = 0; i < S; ++i) { h_info[i] = (int32_t)dk[(size_t)i]; h_info[16 + i] = (int32_t)mk[(size_t)i]; h_info[32 + i] = 0; }
    if (fixed)
        for (int i = 1; i < S; ++i)
            for (int q = 0; q < 16; ++q) h_T[16 * i + q] = h_init[16 * (i - 1) + q];
    if (S > 1 && !fixed) {
        const bool plane = prm->icp_mode == KPX_ICP_POINT_TO_PLANE;
        if (plane) KPX_SUB(kpx_estimate_normals(L.down_pts, dk[0], 2.0 * prm->reg_voxel, prm->normals_nn, L.normals, L.op_ws, L.op_bytes, st));
        std::vector<const float *> subs((size_t)S - 1);
        for (int i = 1; i < S; ++i) {
            KPX_REQUIRE(dk[(size_t)i] >= 1 && dk[0] >= 1, "kpx_frame_step: sensor %d has no valid pixel", dk[0] >= 1 ? i : 0);
            subs[(size_t)i - 1] = p_out[(size_t)i];
        }
        KPX_SUB(icp_batch_ordered(S - 1, subs.data(), dk.data() + 1, L.down_pts, plane ? L.normals : nullptr, dk[0], prm->icp_max_dist, h_init, prm->icp_mode,
                                  prm->icp_max_iteration, 1e-6, 1e-6, L.icp_res, L.op_ws, L.op_bytes, st, zorder));
        // the results reach the host with the NEXT read-back: the fuse below takes the transforms from device memory, in stream order
    }
    // fuse: pcd.transform(T_i) + np.vstack + voxel_down_sample in one fp64 pass, remove_statistical_outlier, selection
    std::vector<const double *> dT((size_t)S, nullptr);
    for (int i = 0; i < S; ++i) {
        p_in[(size_t)i] = L.mask_pts + (size_t)i * n_px * 3; c_in[(size_t)i] = L.mask_col + (size_t)i * n_px * 3;
        if (i > 0 && !fixed) dT[(size_t)i] = L.icp_res + 20 * (size_t)(i - 1);
    }
    // The fused cloud's sort-key width is speculated from this thread's previous frame, like the registration grids' above: <= 32 bits,
    // the library's own radix sort; a frame that needs more is seen at the read-back below and fused again the careful way.
    static thread_local int fuse_spec = 0;
    if (!speculate) fuse_spec = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        h_i[51] = 0;
        KPX_SUB(fuse_voxel_downsample_dev(S, p_in.data(), c_in.data(), mk.data(), h_T, dT.data(), prm->filt_voxel, L.vox_pts, L.vox_col, L.vox_cnt, L.op_ws,
                                          L.op_bytes, st, attempt == 0 ? fuse_spec : 0, h_i + 51));
        KPX_HIP(hipMemcpyAsync(h_d, L.icp_res, ((size_t)S * 20 + 1) * sizeof(double), hipMemcpyDeviceToHost, st));
        KPX_SUB(frame_wait(st));
        const int need = h_i[51];
        const bool narrow = attempt == 0 && fuse_spec > 0 && need > fuse_spec;
        fuse_spec = (speculate && need > 0 && need <= 32) ? (need + 7) / 8 * 8 : 0;      // whole 8-bit passes; wide keys are not speculated
        if (!narrow) break;
    }
    h_i[48] = *reinterpret_cast<const int32_t *>(h_d + (size_t)S * 20);
    for (int i = 1; i < S && !fixed; ++i)                  // (a one-launch ICP chain that lost its race for residency: its results are NaN)
        if (h_d[20 * (i - 1) + 16] != h_d[20 * (i - 1) + 16]) {
            (void)icp_chain_abort_take();
            return fail(KPX_ERR_HIP, "kpx_frame_step: the one-launch ICP chain of sensor %d gave up waiting for its blocks to become resident; KPX_ICP_CHAIN=0 "
                                     "selects the launch-per-iteration form", i);
        }
    for (int i = 1; i < S && !fixed; ++i) {
        for (int q = 0; q < 16; ++q) h_T[16 * i + q] = h_d[20 * (i - 1) + q];
        if (h_info) h_info[32 + i] = (int32_t)h_d[20 * (i - 1) + 18];
    }
    if (h_i[48] < 0) return fail(KPX_ERR_RANGE, "voxel_size is too small");
    const int64_t M = h_i[48];
    if (h_info) h_info[48] = (int32_t)M;
    *h_count = 0;
    if (M == 0) return KPX_OK;
    // filter + selection in one pass (no count read-back between them)
    KPX_SUB(kpx_sor_select(L.vox_pts, L.vox_col, M, prm->filt_k, prm->filt_ratio, out_pts, out_col, L.keep_idx, h_i + 49, L.sor_stats, L.op_ws, L.op_bytes, st));
    KPX_SUB(frame_wait(st));
    *h_count = h_i[49];
    return KPX_OK;
}

// The same frame handed over in HOST memory ("depth frame resident in host pinned memory -> fused registered cloud
// resident on the GPU"; the reference's loop starts from files, data.py:87-124).  The two images are copied into staging buffers
// at the head of the caller's workspace on the frame's own stream -- no allocation, no extra synchronisation; with several
// frames in flight (one stream each) the copy of frame k+1 runs under the kernels of frame k.
KPX_EXPORT size_t kpx_frame_step_host_workspace_bytes(int32_t sensors, int64_t n_px)
{
    if (sensors < 1 || n_px < 1) return 0;
    Arena a(nullptr, 0);
    a.get<uint16_t>((size_t)sensors * (size_t)n_px);
    a.get<uint8_t>((size_t)sensors * (size_t)n_px * 3);
    return a.off + kpx_frame_step_workspace_bytes(sensors, n_px);
}

KPX_EXPORT int kpx_frame_step_host(const uint16_t *h_depth, const uint8_t *h_rgb, const float *xy_table, int64_t n_px, int32_t sensors,
                                   const double *h_init, const kpx_frame_params *prm, float *out_pts, float *out_col, int32_t *h_count,
                                   double *h_T, int32_t *h_info, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(sensors >= 1 && sensors <= 16 && n_px > 0 && n_px < ((int64_t)1 << 31) / 16, "kpx_frame_step_host: 1 .. 16 sensors of at most 2^27 pixels");
    KPX_REQUIRE(h_depth && h_rgb && ws, "kpx_frame_step_host: null pointer");
    Arena a(ws, ws_bytes);
    const size_t all = (size_t)sensors * (size_t)n_px;
    uint16_t *d_depth = a.get<uint16_t>(all);
    uint8_t *d_rgb = a.get<uint8_t>(all * 3);
    KPX_ARENA_CHECK(a);
    hipStream_t st = (hipStream_t)stream;
    KPX_HIP(hipMemcpyAsync(d_depth, h_depth, all * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    KPX_HIP(hipMemcpyAsync(d_rgb, h_rgb, all * 3, hipMemcpyHostToDevice, st));
    return kpx_frame_step(d_depth, d_rgb, xy_table, n_px, sensors, h_init, prm, out_pts, out_col, h_count, h_T, h_info, (char *)ws + a.off,
                          ws_bytes - a.off, stream);
}

// ---- the same frame over several GPUs: the north-star partition, host side in C++ ---------------------------------------------
// Sensor g lives on GPU g (fewer GPUs than sensors: contiguous blocks, `shard_first`); per frame three collectives on the frame's
// stream (kpx_comm: RCCL from C++, or the caller's callbacks), issued in the slot-independent order of kpx_order:
//   0  broadcast of the master's down-sampled cloud + normals from rank 0     (data.py:140-157: every sub registers onto the master)
//   1  all-gather of the masked clouds, UNMOVED, with their counts and registration results in header rows   (data.py:44-58)
//   2  all-gather of the slabs' mean neighbour distances (sharded filter_outliers on the fused cloud, data.py:61)
// Message capacities follow the slot's previous frame (+25 %, the same on every rank because every rank reads the same headers);
// a frame that outgrows one is NOT patched up with an extra collective (its issue order would depend on timing): every rank sees
// the overflow in the same header, enlarges the capacity, passes the frame's remaining stages and returns KPX_RETRY -- the caller
// runs the frame again.  The first frame of a slot uses the worst-case capacities.
namespace kpx {

int64_t &comm_cap_master(kpx_comm *c);
int64_t &comm_cap_clouds(kpx_comm *c);
int &comm_spec_bits(kpx_comm *c);
int &comm_fuse_bits(kpx_comm *c);

constexpr int kHdrDoubles = 24;        // per sensor: [0] masked points, [1] down-sampled points, [2..21] registration (T, fitness, rmse, iterations, pairs)

static int shard_first(int n_sensors, int rank, int world)
{
    const int base = n_sensors / world, rem = n_sensors % world;
    return rank * base + (rank < rem ? rank : rem);
}
static int shard_count(int n_sensors, int rank, int world) { return n_sensors / world + (rank < n_sensors % world ? 1 : 0); }
static int64_t round_cap(int64_t need, int64_t worst)
{
    int64_t c = (need + need / 4 + 4095) / 4096 * 4096;
    if (c < 4096) c = 4096;
    return c < worst ? c : worst;
}

struct ShardHeaderArgs {
    int32_t k_max, local, first_is_master;
    int32_t masked[16], down[16];
};
// header rows of this rank's exchange message: counts by value, registration results from device memory (stream order)
__global__ void shard_header_kernel(double *hdr, ShardHeaderArgs a, const double *icp_res)
{
    const int j = blockIdx.x, t = threadIdx.x;
    if (t >= kHdrDoubles) return;
    double v = 0.0;
    if (j < a.local) {
        const int sub = j - (a.first_is_master ? 1 : 0);        // index into icp_res; -1 = the master itself
        if (t == 0) v = (double)a.masked[j];
        else if (t == 1) v = (double)a.down[j];
        else if (t < 22) v = sub >= 0 ? icp_res[20 * sub + (t - 2)] : ((t - 2) < 16 && (t - 2) % 5 == 0 ? 1.0 : 0.0);
    } else if (t >= 2 && t < 18 && (t - 2) % 5 == 0) v = 1.0;
    hdr[j * kHdrDoubles + t] = v;
}

struct ShardLayout {
    uint16_t *stage_depth;
    uint8_t *stage_rgb;
    float *full_pts, *mask_pts, *mask_col, *down_pts, *normals, *vox_pts, *vox_col;
    char *msg_master, *xchg_send, *xchg_recv;
    int32_t *keep_idx, *order_idx, *vox_cnt, *keep_cnt;
    double *avg_send, *avg_all, *icp_res, *sor_stats;
    void *op_ws;
    size_t op_bytes, hdr_bytes;
    int k_max;
};
static void shard_carve(Arena &a, int S, int S_l, int world, int64_t n_px, bool host_input, ShardLayout *L)
{
    const size_t px = (size_t)n_px, loc = (size_t)S_l * px, all = (size_t)S * px;
    L->k_max = (S + world - 1) / world;
    L->hdr_bytes = ((size_t)L->k_max * kHdrDoubles * sizeof(double) + 255) & ~(size_t)255;
    L->stage_depth = host_input ? a.get<uint16_t>(loc) : nullptr;
    L->stage_rgb = host_input ? a.get<uint8_t>(loc * 3) : nullptr;
    L->full_pts = a.get<float>(loc * 3);
    L->mask_pts = a.get<float>(loc * 3);
    L->mask_col = a.get<float>(loc * 3);
    L->down_pts = a.get<float>(loc * 3);
    L->normals = a.get<float>(px * 3);
    L->msg_master = a.get<char>(px * 24 + 256);
    const size_t xmsg = (size_t)L->k_max * px * 24 + L->hdr_bytes;
    L->xchg_send = a.get<char>(xmsg);
    L->xchg_recv = a.get<char>(xmsg * (size_t)world);
    L->vox_pts = a.get<float>(all * 3);
    L->vox_col = a.get<float>(all * 3);
    L->keep_idx = a.get<int32_t>(all);
    L->order_idx = a.get<int32_t>(all);
    const size_t rows = (all + (size_t)world - 1) / (size_t)world;
    L->avg_send = a.get<double>(rows);
    L->avg_all = a.get<double>(rows * (size_t)world);
    L->icp_res = a.get<double>((size_t)(S_l > 0 ? S_l : 1) * 20);
    L->vox_cnt = a.get<int32_t>(64);
    L->keep_cnt = L->vox_cnt + 1;
    L->sor_stats = a.get<double>(4);
    std::vector<int64_t> worst((size_t)(S_l > 0 ? S_l : 1), n_px);
    size_t w = kpx_depth_to_cloud_workspace_bytes(n_px, S_l);
    w = max_sz(w, kpx_voxel_batch_workspace_bytes(S_l, worst.data()));
    w = max_sz(w, kpx_normals_workspace_bytes(n_px, KPX_NORMALS_LDS_NN));
    w = max_sz(w, kpx_icp_batch_workspace_bytes(S_l > 0 ? S_l : 1, worst.data(), n_px));
    w = max_sz(w, kpx_fuse_voxel_workspace_bytes((int64_t)all));
    w = max_sz(w, kpx_sor_workspace_bytes((int64_t)all, KPX_SOR_LDS_K));
    w = max_sz(w, kpx_sor_finish_workspace_bytes((int64_t)all));
    w = max_sz(w, kpx_select_workspace_bytes((int64_t)all));
    L->op_bytes = w;
    L->op_ws = a.get<char>(w);
}

// pinned read-back block of the calling thread (32 KiB): [0, 8K) doubles, [8K, 12K) ints, [12K, 32K) gathered headers
static int pinned_block(char **out)
{
    static thread_local char *blk = nullptr;
    if (!blk) {
        void *p = nullptr;
        KPX_HIP(hipHostMalloc(&p, 32768, hipHostMallocDefault));
        blk = static_cast<char *>(p);
    }
    *out = blk;
    return KPX_OK;
}

}  // namespace kpx

KPX_EXPORT size_t kpx_frame_step_sharded_workspace_bytes(int32_t sensors, int32_t rank, int32_t world, int64_t n_px, int32_t host_input)
{
    if (sensors < 1 || world < 1 || rank < 0 || rank >= world || world > sensors || n_px < 1) return 0;
    Arena a(nullptr, 0);
    ShardLayout L;
    shard_carve(a, sensors, shard_count(sensors, rank, world), world, n_px, host_input != 0, &L);
    return a.off;
}

KPX_EXPORT int kpx_frame_step_sharded(kpx_comm *comm, kpx_order *order, int64_t frame, const uint16_t *depth, const uint8_t *rgb, int32_t host_input,
                                      const float *xy_table, int64_t n_px, int32_t sensors, const double *h_init, const kpx_frame_params *prm,
                                      int32_t fused_filter, float *out_pts, float *out_col, int32_t *h_count, double *h_T, int32_t *h_info, void *ws,
                                      size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(comm, "kpx_frame_step_sharded: communicator missing (kpx_comm_create_*)");
    const int rank = kpx_comm_rank(comm), world = kpx_comm_world(comm), S = sensors;
    KPX_REQUIRE(S >= 1 && S <= 16 && world <= S && n_px > 0 && n_px < ((int64_t)1 << 31) / 16, "kpx_frame_step_sharded: 1 .. 16 sensors on at most as many ranks");
    KPX_REQUIRE(depth && rgb && xy_table && prm && out_pts && out_col && h_count && h_T && ws, "kpx_frame_step_sharded: null pointer");
    KPX_REQUIRE(S == 1 || h_init, "kpx_frame_step_sharded: initial transforms missing");
    KPX_REQUIRE(prm->icp_mode == KPX_ICP_POINT_TO_POINT || prm->icp_mode == KPX_ICP_POINT_TO_PLANE, "kpx_frame_step_sharded: bad icp_mode");
    KPX_REQUIRE(fused_filter >= 0 && fused_filter <= 2, "kpx_frame_step_sharded: fused_filter is 0 (sharded), 1 (rank 0) or 2 (the frames' fuse + filter dealt round robin)");
    // fused_filter 2 (round 5): frame f's fused transform + voxel + filter run on ONE rank, f mod world, the others are done behind the
    // exchange -- the sharded form repeats the fused pass and the filter's grid on every rank (~170 us of a rank's ~790 us of kernel
    // time per frame, bench.py --emulate-world 8); dealt round robin every rank does them for one frame in `world`.  The frame number is
    // the kpx_order's (the same on every rank; without an order: rank 0 owns every frame).  The owner returns the frame; the others 0 rows.
    const int owner = fused_filter == 2 ? (int)((frame < 0 ? 0 : frame) % world) : 0;
    KPX_REQUIRE(prm->filt_k <= KPX_SOR_LDS_K && prm->normals_nn <= KPX_NORMALS_LDS_NN, "kpx_frame_step_sharded: filt_k <= %d and normals_nn <= %d", KPX_SOR_LDS_K,
                KPX_NORMALS_LDS_NN);
    hipStream_t st = (hipStream_t)stream;
    BusyScope busy;
    const int g0 = shard_first(S, rank, world), S_l = shard_count(S, rank, world);
    const bool owns_master = g0 == 0, plane = prm->icp_mode == KPX_ICP_POINT_TO_PLANE;
    Arena a(ws, ws_bytes);
    ShardLayout L;
    shard_carve(a, S, S_l, world, n_px, host_input != 0, &L);
    KPX_ARENA_CHECK(a);
    const int K = L.k_max;
    char *pin = nullptr;
    KPX_SUB(pinned_block(&pin));
    double *h_d = reinterpret_cast<double *>(pin);
    int32_t *h_i = reinterpret_cast<int32_t *>(pin + 8192);
    double *h_hdr = reinterpret_cast<double *>(pin + 12288);                     // world x K x kHdrDoubles doubles <= 16 x 24 x 8 B
    auto negative = [&](const int32_t *c, int n) { for (int i = 0; i < n; ++i) if (c[i] < 0) return c[i]; return 0; };
    *h_count = 0;
    if (h_info) memset(h_info, 0, 64 * sizeof(int32_t));

    if (host_input) {
        KPX_HIP(hipMemcpyAsync(L.stage_depth, depth, (size_t)S_l * n_px * sizeof(uint16_t), hipMemcpyHostToDevice, st));
        KPX_HIP(hipMemcpyAsync(L.stage_rgb, rgb, (size_t)S_l * n_px * 3, hipMemcpyHostToDevice, st));
        depth = L.stage_depth;
        rgb = L.stage_rgb;
    }
    // -- this rank's sensors: registration input (every valid pixel) and person clouds (mask + gate + colours)
    KPX_SUB(kpx_depth_to_cloud(depth, xy_table, nullptr, n_px, S_l, 0, prm->gate, L.full_pts, nullptr, nullptr, h_i, L.op_ws, L.op_bytes, st));
    KPX_SUB(kpx_depth_to_cloud(depth, xy_table, rgb, n_px, S_l, KPX_COMPACT_COLOR_MASK | KPX_COMPACT_DEPTH_GATE, prm->gate, L.mask_pts, L.mask_col, nullptr,
                               h_i + 16, L.op_ws, L.op_bytes, st));
    std::vector<int64_t> fk((size_t)S_l), mk((size_t)S_l), dk((size_t)S_l);
    std::vector<const float *> p_in((size_t)S_l);
    std::vector<float *> p_out((size_t)S_l);
    for (int i = 0; i < S_l; ++i) { p_in[(size_t)i] = L.full_pts + (size_t)i * n_px * 3; p_out[(size_t)i] = L.down_pts + (size_t)i * n_px * 3; }
    KPX_SUB(frame_wait(st));
    for (int i = 0; i < S_l; ++i) { fk[(size_t)i] = h_i[i]; mk[(size_t)i] = h_i[16 + i]; }
    // A data-dependent failure on ONE rank (an occluded camera, a voxel size too small for its cloud) must not leave its peers
    // spinning inside a collective it never enters: the rank keeps its place in the frame's collectives with an empty payload and a
    // NEGATIVE count in the header it contributes (master header: rank 0; exchange header rows: every rank), every rank reads the
    // same headers and all of them return an error behind the same collective -- as the capacity overflow does with KPX_RETRY.
    int lerr = KPX_OK;
    if (negative(h_i, S_l) || negative(h_i + 16, S_l)) lerr = fail(KPX_ERR_RANGE, "kpx_frame_step_sharded: extraction reported %d", negative(h_i, S_l) | negative(h_i + 16, S_l));
    static const bool zorder_on = [] { const char *e = getenv("KPX_FRAME_ZORDER"); return !(e && e[0] == '0'); }();
    const bool zorder = zorder_on && K <= 8;                   // the same decision on every rank: the master arrives in the order rank 0 gave it
    int &spec_bits = comm_spec_bits(comm);
    for (int attempt = 0; attempt < 2 && !lerr; ++attempt) {
        KPX_SUB(voxel_downsample_batch_spec(S_l, p_in.data(), nullptr, fk.data(), prm->reg_voxel, p_out.data(), nullptr, h_i + 32, L.op_ws, L.op_bytes, st,
                                            attempt == 0 ? spec_bits : 0, h_i + 50, zorder));
        KPX_SUB(frame_wait(st));
        const int need = h_i[50];
        const bool narrow = attempt == 0 && spec_bits > 0 && need > spec_bits;
        spec_bits = need > 0 && need <= 32 ? (need + 7) / 8 * 8 : 0;
        if (!narrow) break;
    }
    if (!lerr && negative(h_i + 32, S_l)) lerr = fail(KPX_ERR_RANGE, "voxel_size is too small");
    for (int i = 0; i < S_l; ++i) dk[(size_t)i] = lerr ? 0 : h_i[32 + i];
    for (int i = 0; i < S_l && !lerr; ++i)
        if (dk[(size_t)i] < 1) lerr = fail(KPX_ERR_INVALID, "kpx_frame_step_sharded: sensor %d has no valid pixel", g0 + i);

    // -- collective 0: the master's down-sampled cloud to every rank.  Message: cap rows xyz | header.  The NORMALS do not travel
    // (round 4): rank 0 used to estimate them in front of the broadcast -- ~0.1 ms during which every other rank waited -- and ship
    // them as the message's second half.  Every rank now estimates them itself from the broadcast points, beside the others: the same
    // kernels on the same array give the same normals bit for bit, the message is half as long, and nothing serial is left between
    // rank 0's voxel grid and everyone's registrations (registration.py:9-13 is a function of the master cloud alone).
    // KPX_SHARD_FIXED_CAP=1: the worst-case capacities in every frame (no adaptation, no retry) -- message sizes then do not depend on
    // the slot's history, which the replay transport of bench.py --emulate-world needs (kpx_comm_create_replay)
    static const bool fixed_cap = [] { const char *e = getenv("KPX_SHARD_FIXED_CAP"); return e && e[0] == '1'; }();
    int64_t &cap_m = comm_cap_master(comm);
    if (cap_m <= 0 || fixed_cap) cap_m = n_px;
    const int64_t capm = cap_m;
    // Round 5: the normals travel again (KPX_SHARD_NORMALS=1: round 4's form) -- rank 0 estimates them once and ships them as the message's second
    // half (cap rows xyz | cap rows normal | header).  Round 4 took them out because rank 0's ~0.1 ms in front of the broadcast was
    // serial for every rank; with the collective order's own lookahead (kpx_stream) the broadcast of a frame is two frames ahead of the
    // exchanges that wait, and what counts is the kernel time a rank spends per frame: the estimate (grid build + search, ~150 us) on
    // ONE rank -- the one without a registration of its own -- instead of on all of them.  Same kernels on the same array: the same
    // normals bit for bit either way.  Measured with bench.py --emulate-world 8.
    static const bool normals_travel = [] { const char *e = getenv("KPX_SHARD_NORMALS"); return !(e && e[0] == '1'); }();      // 1: every rank estimates them (round 4)
    const bool ship_n = normals_travel && plane && world > 1;
    const size_t row_b = ship_n ? 24 : 12;
    char *msg = L.msg_master;
    float *m_xyz = reinterpret_cast<float *>(msg), *m_nrm = reinterpret_cast<float *>(msg + (size_t)capm * 12);
    double *m_hdr = reinterpret_cast<double *>(msg + (size_t)capm * row_b);
    if (owns_master) {
        if (!lerr && ship_n) lerr = kpx_estimate_normals(L.down_pts, dk[0], 2.0 * prm->reg_voxel, prm->normals_nn, L.normals, L.op_ws, L.op_bytes, st);
        if (!lerr) {
            const size_t rows = (size_t)(dk[0] < capm ? dk[0] : capm);
            KPX_HIP(hipMemcpyAsync(m_xyz, L.down_pts, rows * 12, hipMemcpyDeviceToDevice, st));
            if (ship_n) KPX_HIP(hipMemcpyAsync(m_nrm, L.normals, rows * 12, hipMemcpyDeviceToDevice, st));
        }
        h_d[0] = lerr ? -1.0 : (double)dk[0];                   // a negative count: rank 0 cannot provide the master (every rank returns)
        KPX_HIP(hipMemcpyAsync(m_hdr, h_d, sizeof(double), hipMemcpyHostToDevice, st));
    }
    kpx_order_turn_begin(order, frame, 0);
    int rc = kpx_comm_broadcast(comm, msg, (size_t)capm * row_b + 256, 0, st);
    kpx_order_turn_end(order, frame, 0);
    if (rc) return rc;
    if (owns_master && lerr) {                                 // (the message of `lerr` is the thread's last error)
        kpx_order_finish(order, frame);
        return lerr;
    }
    int64_t m = owns_master ? dk[0] : 0;
    if (!owns_master) {
        KPX_HIP(hipMemcpyAsync(h_d, m_hdr, sizeof(double), hipMemcpyDeviceToHost, st));
        KPX_SUB(frame_wait(st));
        m = (int64_t)h_d[0];
        if (!(m >= 1 && m <= n_px)) {
            kpx_order_finish(order, frame);
            return m < 0 ? fail(KPX_ERR_RANGE, "kpx_frame_step_sharded: rank 0 could not provide the master cloud (see its error)")
                         : fail(KPX_ERR_RANGE, "kpx_frame_step_sharded: bad master header (%lld points)", (long long)m);
        }
    }
    cap_m = fixed_cap ? n_px : round_cap(m, n_px);
    if (m > capm) {                                            // every rank reads the same m: all retry, none goes on
        kpx_order_finish(order, frame);
        return KPX_RETRY;
    }
    // -- registration of this rank's sub sensors onto the master
    const int n_sub = S_l - (owns_master ? 1 : 0);
    if (n_sub > 0 && !lerr) {
        std::vector<const float *> subs((size_t)n_sub);
        std::vector<int64_t> ns((size_t)n_sub);
        for (int j = 0; j < n_sub; ++j) {
            const int i = j + (owns_master ? 1 : 0);
            subs[(size_t)j] = p_out[(size_t)i];
            ns[(size_t)j] = dk[(size_t)i];
        }
        const float *tgt = owns_master ? L.down_pts : m_xyz, *tn = plane ? ((ship_n && !owns_master) ? m_nrm : L.normals) : nullptr;
        // (a failure here is this rank's alone: it keeps its place in collective 1 with a negative count -- `lerr` -- and every rank
        // returns together; returning from here would leave the peers spinning in the all-gather)
        if (plane && !ship_n) lerr = kpx_estimate_normals(tgt, m, 2.0 * prm->reg_voxel, prm->normals_nn, L.normals, L.op_ws, L.op_bytes, st);
        const int first_sub = g0 + (owns_master ? 1 : 0);      // global sensor number of subs[0]; h_init[g - 1] belongs to sensor g
        if (!lerr)
            lerr = icp_batch_ordered(n_sub, subs.data(), ns.data(), tgt, tn, m, prm->icp_max_dist, h_init + 16 * (size_t)(first_sub - 1), prm->icp_mode,
                                     prm->icp_max_iteration, 1e-6, 1e-6, L.icp_res, L.op_ws, L.op_bytes, st, zorder);
    }
    // -- collective 1: the masked clouds, unmoved, planar (xyz rows, then rgb rows), + header rows
    int64_t &cap_c = comm_cap_clouds(comm);
    const int64_t worst_c = (int64_t)K * n_px;
    if (cap_c <= 0 || fixed_cap) cap_c = worst_c;
    const int64_t capc = cap_c;
    const size_t xbytes = (size_t)capc * 24 + L.hdr_bytes;
    {
        ShardHeaderArgs ha;
        memset(&ha, 0, sizeof(ha));
        ha.k_max = K; ha.local = S_l; ha.first_is_master = owns_master ? 1 : 0;
        int64_t off = 0;
        for (int i = 0; i < S_l; ++i) {
            ha.masked[i] = lerr ? -1 : (int32_t)mk[(size_t)i]; ha.down[i] = (int32_t)dk[(size_t)i];   // -1: this rank failed, see above
            int64_t k = capc - off < mk[(size_t)i] ? capc - off : mk[(size_t)i];
            if (k < 0 || lerr) k = 0;
            if (k > 0) {
                KPX_HIP(hipMemcpyAsync(L.xchg_send + (size_t)off * 12, L.mask_pts + (size_t)i * n_px * 3, (size_t)k * 12, hipMemcpyDeviceToDevice, st));
                KPX_HIP(hipMemcpyAsync(L.xchg_send + (size_t)(capc + off) * 12, L.mask_col + (size_t)i * n_px * 3, (size_t)k * 12, hipMemcpyDeviceToDevice, st));
            }
            off += k;
        }
        hipLaunchKernelGGL(shard_header_kernel, dim3((unsigned)K), dim3(32), 0, st, reinterpret_cast<double *>(L.xchg_send + (size_t)capc * 24), ha, L.icp_res);
        KPX_LAUNCH_CHECK();
    }
    kpx_order_turn_begin(order, frame, 1);
    rc = kpx_comm_allgather(comm, L.xchg_send, L.xchg_recv, xbytes, st);
    kpx_order_turn_end(order, frame, 1);
    if (rc) return rc;
    const size_t hrow = (size_t)K * kHdrDoubles * sizeof(double);
    KPX_HIP(hipMemcpy2DAsync(h_hdr, hrow, L.xchg_recv + (size_t)capc * 24, xbytes, hrow, (size_t)world, hipMemcpyDeviceToHost, st));
    KPX_SUB(frame_wait(st));
    std::vector<const float *> f_p((size_t)S), f_c((size_t)S);
    std::vector<int64_t> f_n((size_t)S);
    int64_t need = 0;
    for (int r = 0, g = 0; r < world; ++r) {
        const int own = shard_count(S, r, world);
        int64_t off = 0;
        for (int j = 0; j < own; ++j, ++g) {
            const double *row = h_hdr + ((size_t)r * K + j) * kHdrDoubles;
            const int64_t n = (int64_t)row[0];
            if (!(n >= 0 && n <= n_px)) {                          // the same rows on every rank: everyone returns here
                kpx_order_finish(order, frame);
                if (lerr) return lerr;
                return n < 0 ? fail(KPX_ERR_RANGE, "kpx_frame_step_sharded: rank %d failed on its own sensors (see its error)", r)
                             : fail(KPX_ERR_RANGE, "kpx_frame_step_sharded: bad exchange header (rank %d)", r);
            }
            if (row[2 + 16] != row[2 + 16]) {                      // NaN fitness: that rank's one-launch ICP chain gave up its residency wait (icp_chain_kernel);
                kpx_order_finish(order, frame);                    // the same rows on every rank: everyone returns here, behind the same collective
                (void)icp_chain_abort_take();
                return fail(KPX_ERR_HIP, "kpx_frame_step_sharded: the one-launch ICP chain of sensor %d (rank %d) gave up waiting for its blocks to become "
                                         "resident; KPX_ICP_CHAIN=0 selects the launch-per-iteration form", g, r);
            }
            f_p[(size_t)g] = reinterpret_cast<const float *>(L.xchg_recv + (size_t)r * xbytes + (size_t)off * 12);
            f_c[(size_t)g] = reinterpret_cast<const float *>(L.xchg_recv + (size_t)r * xbytes + (size_t)(capc + off) * 12);
            f_n[(size_t)g] = n;
            for (int q = 0; q < 16; ++q) h_T[16 * g + q] = row[2 + q];
            if (h_info) { h_info[g] = (int32_t)row[1]; h_info[16 + g] = (int32_t)n; h_info[32 + g] = g == 0 ? 0 : (int32_t)row[2 + 18]; }
            off += n;
        }
        need = off > need ? off : need;
    }
    cap_c = fixed_cap ? worst_c : round_cap(need, worst_c);
    if (need > capc) {
        kpx_order_finish(order, frame);
        return KPX_RETRY;
    }
    if (fused_filter >= 1 && rank != owner) {                  // the owner filters alone: the others are done with this frame
        kpx_order_skip(order, frame, 2);
        return KPX_OK;
    }
    // -- fuse: pcd.transform(T_i) + np.vstack + voxel_down_sample in one fp64 pass (every rank that filters: identical everywhere)
    // (the fused cloud's key width speculated from the slot's previous frame, as in kpx_frame_step: the library's own radix sort instead
    // of the vendor's merge sort; every rank fuses the same cloud, so every rank speculates and -- rarely -- repeats alike)
    int &fuse_spec = comm_fuse_bits(comm);
    static const bool speculate_fuse = [] { const char *e = getenv("KPX_FRAME_SPECULATE"); return !(e && e[0] == '0'); }();
    if (!speculate_fuse) fuse_spec = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        h_i[51] = 0;
        KPX_SUB(fuse_voxel_downsample_dev(S, f_p.data(), f_c.data(), f_n.data(), h_T, nullptr, prm->filt_voxel, L.vox_pts, L.vox_col, L.vox_cnt, L.op_ws,
                                          L.op_bytes, st, attempt == 0 ? fuse_spec : 0, h_i + 51));
        KPX_HIP(hipMemcpyAsync(h_i + 48, L.vox_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        KPX_SUB(frame_wait(st));
        const int need_bits = h_i[51];
        const bool narrow = attempt == 0 && fuse_spec > 0 && need_bits > fuse_spec;
        fuse_spec = (speculate_fuse && need_bits > 0 && need_bits <= 32) ? (need_bits + 7) / 8 * 8 : 0;
        if (!narrow) break;
    }
    if (h_i[48] < 0) return fail(KPX_ERR_RANGE, "voxel_size is too small");
    const int64_t M = h_i[48];
    if (h_info) h_info[48] = (int32_t)M;
    if (M == 0) {
        kpx_order_skip(order, frame, 2);
        return KPX_OK;
    }
    if (fused_filter >= 1) {                                   // the owner, alone: the one-GPU filter + selection
        kpx_order_skip(order, frame, 2);
        KPX_SUB(kpx_sor_select(L.vox_pts, L.vox_col, M, prm->filt_k, prm->filt_ratio, out_pts, out_col, L.keep_idx, h_i + 49, L.sor_stats, L.op_ws, L.op_bytes, st));
        KPX_SUB(frame_wait(st));
        *h_count = h_i[49];
        return KPX_OK;
    }
    // -- sharded filter: this rank searches the neighbours of slab `rank` of the grid order; collective 2 = the slabs' mean distances
    const int64_t rows = (M + world - 1) / world;
    const int64_t q0 = rank * rows < M ? rank * rows : M, q1 = (rank + 1) * rows < M ? (rank + 1) * rows : M;
    KPX_SUB(kpx_sor_partial(L.vox_pts, M, prm->filt_k, q0, q1, L.avg_send, L.order_idx, L.op_ws, L.op_bytes, st));
    kpx_order_turn_begin(order, frame, 2);
    rc = kpx_comm_allgather(comm, L.avg_send, L.avg_all, (size_t)rows * sizeof(double), st);
    kpx_order_turn_end(order, frame, 2);
    if (rc) return rc;
    KPX_SUB(kpx_sor_finish(L.avg_all, L.order_idx, M, prm->filt_ratio, L.keep_idx, L.keep_cnt, L.sor_stats, nullptr, L.op_ws, L.op_bytes, st));
    KPX_HIP(hipMemcpyAsync(h_i + 49, L.keep_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    KPX_SUB(frame_wait(st));
    const int64_t kept = h_i[49];
    if (kept > 0) {
        KPX_SUB(kpx_select_by_index(L.vox_pts, L.vox_col, nullptr, M, L.keep_idx, kept, KPX_SELECT_GATHER, out_pts, out_col, nullptr, nullptr, L.op_ws,
                                    L.op_bytes, st));
        // the rows are complete when the call returns, as kpx_frame_step's are (a kpx_stream hands them to the caller without another
        // synchronisation; until round 5 pipeline.FrameStream synchronised the slot's stream behind every step)
        KPX_SUB(frame_wait(st));
    }
    *h_count = (int32_t)kept;
    return KPX_OK;
}
