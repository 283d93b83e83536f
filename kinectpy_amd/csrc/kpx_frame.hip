// kpx_frame.hip -- one frame set of the reference's frame loop as ONE native call (preprocessing/data.py:35-61 with the
// registration of data.py:127-161 folded in, the unit bench.py measures): host orchestration in C++ over the library's own
// C ABI.  The Python mirror of the same loop (kinectpy_amd/pipeline.py) issues ~12 library calls per frame with interpreter
// work, tensor allocations and the GIL in between; with several frames in flight on host threads that interpreter time is
// serialised.  Here a frame is a single call with the GIL released throughout; everything lives in the caller's workspace.
#include <vector>

#include "kpx_internal.h"

namespace kpx {

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
    w = max_sz(w, kpx_normals_workspace_bytes(n_px, KPX_NORMALS_MAX_NN));
    w = max_sz(w, kpx_icp_batch_workspace_bytes(S > 1 ? S - 1 : 1, worst.data(), n_px));
    w = max_sz(w, kpx_fuse_voxel_workspace_bytes((int64_t)all));
    w = max_sz(w, kpx_sor_workspace_bytes((int64_t)all, KPX_SOR_MAX_K));
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
    KPX_REQUIRE(prm->icp_mode == KPX_ICP_POINT_TO_POINT || prm->icp_mode == KPX_ICP_POINT_TO_PLANE, "kpx_frame_step: bad icp_mode");
    const int S = sensors;
    hipStream_t st = (hipStream_t)stream;
    Arena a(ws, ws_bytes);
    FrameLayout L;
    frame_carve(a, S, n_px, &L);
    KPX_ARENA_CHECK(a);
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
    KPX_SUB(kpx_depth_to_cloud(depth, xy_table, nullptr, n_px, S, 0, prm->gate, L.full_pts, nullptr, nullptr, h_i, L.op_ws, L.op_bytes, st));
    // (every operator runs on `st`: stream order alone makes the shared scratch region safe)
    KPX_SUB(kpx_depth_to_cloud(depth, xy_table, rgb, n_px, S, KPX_COMPACT_COLOR_MASK | KPX_COMPACT_DEPTH_GATE, prm->gate, L.mask_pts, L.mask_col, nullptr,
                               h_i + 16, L.op_ws, L.op_bytes, st));
    std::vector<int64_t> fk((size_t)S), mk((size_t)S), dk((size_t)S);
    // registration: voxel_down_sample(reg_voxel) of every sensor's cloud, normals of the master's, point-to-plane ICP of every sub
    std::vector<const float *> p_in((size_t)S), c_in((size_t)S);
    std::vector<float *> p_out((size_t)S);
    for (int i = 0; i < S; ++i) { p_in[(size_t)i] = L.full_pts + (size_t)i * n_px * 3; p_out[(size_t)i] = L.down_pts + (size_t)i * n_px * 3; }
    KPX_HIP(hipStreamSynchronize(st));                     // read-back 1: both extractions' counts
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
    for (int attempt = 0; attempt < 2; ++attempt) {
        KPX_SUB(voxel_downsample_batch_spec(S, p_in.data(), nullptr, fk.data(), prm->reg_voxel, p_out.data(), nullptr, h_i + 32, L.op_ws, L.op_bytes, st,
                                            attempt == 0 ? spec_bits : 0, h_i + 50, zorder));
        KPX_HIP(hipStreamSynchronize(st));
        const int need = h_i[50];
        const bool narrow = attempt == 0 && spec_bits > 0 && need > spec_bits;
        spec_bits = need > 0 && need <= 32 ? (need + 7) / 8 * 8 : 0;      // whole 8-bit passes; wide keys are not speculated
        if (!narrow) break;
    }
    if (negative(h_i + 32, S)) return fail(KPX_ERR_RANGE, "voxel_size is too small");
    for (int i = 0; i < S; ++i) dk[(size_t)i] = h_i[32 + i];
    for (int q = 0; q < 16; ++q) h_T[q] = (q % 5 == 0) ? 1.0 : 0.0;
    if (h_info) for (int i = 0; i < S; ++i) { h_info[i] = (int32_t)dk[(size_t)i]; h_info[16 + i] = (int32_t)mk[(size_t)i]; h_info[32 + i] = 0; }
    if (S > 1) {
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
        if (i > 0) dT[(size_t)i] = L.icp_res + 20 * (size_t)(i - 1);
    }
    KPX_SUB(fuse_voxel_downsample_dev(S, p_in.data(), c_in.data(), mk.data(), h_T, dT.data(), prm->filt_voxel, L.vox_pts, L.vox_col, L.vox_cnt, L.op_ws,
                                      L.op_bytes, st));
    KPX_HIP(hipMemcpyAsync(h_d, L.icp_res, ((size_t)S * 20 + 1) * sizeof(double), hipMemcpyDeviceToHost, st));
    KPX_HIP(hipStreamSynchronize(st));
    h_i[48] = *reinterpret_cast<const int32_t *>(h_d + (size_t)S * 20);
    for (int i = 1; i < S; ++i) {
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
    KPX_HIP(hipStreamSynchronize(st));
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
