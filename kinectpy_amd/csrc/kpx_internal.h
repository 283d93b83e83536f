// kpx_internal.h -- functions shared between the translation units of libkinectpx.so.
#pragma once
#include "kpx_common.h"

namespace kpx {

// Per-axis min / max of a float32 (n,3) cloud as doubles: d_bbox6 = (minx,miny,minz,maxx,maxy,maxz).
// Exact (min/max need no rounding).  Scratch: bbox_partials_count(n) * 6 doubles.
constexpr int kBboxBlocks = 1024;
int bbox_f32(const float *pts, int64_t n, double *d_bbox6, double *ws_partials, hipStream_t st);

// ---- uniform grid over a cloud (exact neighbour searches: SOR, normals) --------------------------
struct GridParams {
    double org[3];
    double h;
    int32_t dim[3];
    int32_t ncell;
};
constexpr int32_t kGridMaxCells = 1 << 22;

struct Grid {
    GridParams *params;      // device
    uint32_t *cell_start;    // device [kGridMaxCells + 1]
    float *sorted_pts;       // device [n*3] points in cell order
    int32_t *sorted_idx;     // device [n] original index of each sorted point
    uint32_t *sorted_keys;   // device [n] cell id of each sorted point
    int32_t *spare;          // device [16] words cleared by the build, for the caller's counters
    int32_t *extra;          // device [kGridExtraWords] more cleared words (the tile states of a compaction that follows the search)
};
constexpr int kGridExtraWords = 4104;
// Carves a Grid out of the arena (dry arenas only count bytes) and, when real, builds it on `st`.
// target_per_cell: desired mean occupancy of non-empty cells.
int grid_build(const float *pts, int64_t n, double target_per_cell, Arena &a, Grid *g, hipStream_t st);

// ---- library-internal lanes ------------------------------------------------------------------------
// Batched entry points run their independent problems (chains of short, latency-bound kernels) side by side on
// kLaneCount internal streams: lanes_fork makes the lanes wait for everything queued on the caller's stream,
// lanes_join makes the caller's stream wait for everything queued on the lanes.  One set per host thread and device,
// created at first use.
constexpr int kLaneCount = 4;
struct LaneSet {
    hipStream_t s[kLaneCount];
    hipEvent_t fork, join[kLaneCount];
};
int lanes_get(LaneSet **out);
int lanes_fork(LaneSet *l, hipStream_t caller, int used);
int lanes_join(LaneSet *l, hipStream_t caller, int used);

// kpx_voxel_downsample_batch with a speculated sort-key width (kpx_frame_step): spec_bits > 0 skips the width's read-back (one host
// round trip per call); *d_bits receives the width the batch needs (0 = the call did not speculate) and the caller repeats the call
// with spec_bits = 0 when that is larger than spec_bits.  spec_bits = 0, d_bits = NULL: the exported behaviour.
// morton: the voxels of every cloud leave in Morton (Z-curve) order of their grid indices instead of ascending (ix, iy, iz) -- same
// voxels, same means; for consumers that only need SOME spatially coherent order (the registration inside kpx_frame_step).
int voxel_downsample_batch_spec(int32_t count, const float *const *h_pts, const float *const *h_col, const int64_t *h_n, double voxel,
                                float *const *h_opts, float *const *h_ocol, int32_t *d_counts, void *ws, size_t ws_bytes, void *stream,
                                int spec_bits, int32_t *d_bits, bool morton = false);

// kpx_fuse_voxel_downsample whose transforms may live in device memory: h_dT[i] non-null = cloud i's row-major 4x4 is read there
// (h_T[16 i ..] is then ignored); h_dT == NULL: the exported behaviour.
int fuse_voxel_downsample_dev(int32_t count, const float *const *h_pts, const float *const *h_col, const int64_t *h_n, const double *h_T,
                              const double *const *h_dT, double voxel, float *opts, float *ocol, int32_t *d_count, void *ws, size_t ws_bytes,
                              void *stream, int spec_bits = 0, int32_t *d_bits = nullptr);
// (spec_bits > 0: the keys are taken to fit that many (<= 32) bits and sorted by the library's own radix sort; *d_bits -- pinned host
// memory -- receives the width the cloud really needs: larger than spec_bits = the outputs are garbage, call again with spec_bits = 0)

// kpx_icp_batch for clouds that already lie along a space-filling curve (presorted = true: no Morton sort inside; the culled
// search's tiles are then 16 consecutive points of the caller's order).  presorted = false: the exported behaviour.
// Host threads of this process that are inside a frame step or an ICP batch right now (each thread counted once).  The one-launch ICP
// chain is a LATENCY form: its resident blocks hold wave slots for the whole chain, most of the time waiting; with other frames in
// flight those slots are worth more to the other frames' kernels (two-sensor rig, four frames in flight: 1320 vs 1570-1620 Mpoints/s
// with the chain, profiles/r04/exp_icp_chain_two_sensors.txt), so a chain is admitted only while the calling thread is alone.
struct BusyScope {
    BusyScope();
    ~BusyScope();
    BusyScope(const BusyScope &) = delete;
    BusyScope &operator=(const BusyScope &) = delete;
    bool counted;
};
int busy_threads();
int icp_chain_abort_take();   // 1 if a one-launch ICP chain gave up waiting since the last call (kpx_icp.hip, icp_chain_kernel)
int icp_batch_ordered(int32_t count, const float *const *h_src, const int64_t *h_n_src, const float *tgt, const float *tgt_normals, int64_t n_tgt,
                      double max_dist, const double *h_init, int32_t mode, int32_t max_iteration, double relative_fitness, double relative_rmse,
                      double *d_results, void *ws, size_t ws_bytes, void *stream, bool presorted);

// The device's ICP engine (kpx_icp.hip, IcpEngine): one host thread + one stream that carry the registrations of every frame in flight
// in one launch per tick.  acquire / release: reference-counted per device (kpx_stream_create / destroy); attach: the CALLING thread's
// icp_batch_ordered hands its groups to the engine from now on (nullptr: drives its own chain of launches).
struct IcpEngine;
IcpEngine *icp_engine_acquire();
void icp_engine_release(IcpEngine *e);
void icp_engine_attach(IcpEngine *e);
void icp_engine_counters(IcpEngine *e, unsigned long long *launches, unsigned long long *ticks);

// Column tiles (16 rows x 16 columns, 2048 flops each) the culled nearest-neighbour sweep has multiplied since the
// last call; resets the device counter (kpx_icp.hip).
double nn_local_take_visits();

}  // namespace kpx
