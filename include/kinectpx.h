/*
 * kinectpx.h -- C ABI of libkinectpx.so: the MI355X (gfx950) hot path behind KinectPy's
 * preprocessing.{extractor,filtering,registration} and floor_removal module APIs.
 *
 * The reference (tiborcamargo/KinectPy) is pure Python: it has no FFI layer; its "operators" are
 * module functions that call Open3D.  Each entry point below names the reference interface it
 * replaces (file:line under the reference root) -- the ctypes stubs a maintainer would add are in
 * INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name starts with `h_`; the caller owns all memory
 *    (torch-ROCm tensors in the Python host); the library never allocates user-visible memory;
 *  - scratch comes from a caller-provided workspace `ws` of at least `kpx_*_workspace_bytes()` bytes,
 *    256-byte aligned;
 *  - clouds are float32 (N,3) row-major ("xyz xyz ..."), colours/normals likewise; index arrays are
 *    int32; rigid transforms are float64 row-major 4x4 (the reference's .npy interchange format,
 *    preprocessing/data.py:158-160);
 *  - outputs of data-dependent length are written into worst-case-sized buffers and their length
 *    into a device int32 (`d_count`);
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*, NULL = default stream)
 *    and returns 0 or a negative kpx_status; `kpx_last_error()` (thread-local) has the message;
 *  - no C++ exception crosses this boundary.
 */
#ifndef KINECTPX_H
#define KINECTPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KPX_VERSION 100

typedef enum {
    KPX_OK = 0,
    KPX_ERR_INVALID = -1,     /* invalid argument (Open3D raises RuntimeError / ValueError here) */
    KPX_ERR_WORKSPACE = -2,   /* workspace too small */
    KPX_ERR_RANGE = -3,       /* value out of the supported range (e.g. voxel index overflow) */
    KPX_ERR_HIP = -4,         /* a HIP runtime call failed */
    KPX_RETRY = 1             /* kpx_frame_step_sharded: a message outgrew its capacity on every rank alike; run the frame again */
} kpx_status;

const char *kpx_last_error(void);
int kpx_version(void);

/* ---- extract stage -------------------------------------------------------------------------- */

/* a1: depth -> XYZ int16 millimetres, the `<ts>_depth.dat` layout that utils/io.py:15-20
 * (load_depth) reads and that preprocessing/extractor.py:68-80 obtains from an external binary.
 * xyz[i] = (floorf(xt*d+0.5f), floorf(yt*d+0.5f), d), (0,0,0) when d==0 or the table entry is NaN.
 * depth: u16 [frames*n_px]; xy_table: f32 [n_px*2] shared by all frames; xyz: i16 [frames*n_px*3]. */
int kpx_unproject_u16(const uint16_t *depth, const float *xy_table, int64_t n_px, int32_t frames,
                      int16_t *xyz, void *stream);

/* a4: np.median of the z column (preprocessing/data.py:170-171), exact, per frame.
 * v: i16 values with `stride` elements between consecutive samples (3 for the z column of an
 * (N,3) .dat array, 1 for a raw depth frame); d_median: f64 [frames]. */
size_t kpx_median_workspace_bytes(int32_t frames);
int kpx_median_i16(const int16_t *v, int64_t n, int64_t stride, int32_t frames, double *d_median,
                   void *ws, size_t ws_bytes, void *stream);

/* a3 + a4: utils/io.py:23-43 rgbd_to_pointcloud (keep = x!=0 & y!=0 & z!=0, colours/255) composed
 * with preprocessing/data.py:165-178 (_transform_filtered_image_to_pointcloud: all colour
 * channels != 0, z <= median + gate).  Order-preserving compaction, per frame.
 * rgb may be NULL (no colours, no colour mask).  flags: bit0 = colour mask, bit1 = depth gate
 * (needs d_median).  Outputs per frame f at offset f*n: pts f32 [n*3], col f32 [n*3] (optional),
 * idx i32 [n] (optional, source pixel index), d_count i32 [frames]. */
#define KPX_COMPACT_COLOR_MASK 1
#define KPX_COMPACT_DEPTH_GATE 2
size_t kpx_compact_workspace_bytes(int64_t n, int32_t frames);
int kpx_rgbd_compact(const int16_t *xyz, const uint8_t *rgb, int64_t n, int32_t frames, int32_t flags,
                     const double *d_median, double gate, float *pts, float *col, int32_t *idx,
                     int32_t *d_count, void *ws, size_t ws_bytes, void *stream);

/* Fused a1+a3+a4 for the streaming pipeline: u16 depth (+ xy table, + optional rgb) straight to the
 * compacted float32 cloud, never materialising the int16 XYZ image.  Same outputs as
 * kpx_rgbd_compact; the median is taken over the raw depth (== the z column). */
size_t kpx_depth_to_cloud_workspace_bytes(int64_t n_px, int32_t frames);
int kpx_depth_to_cloud(const uint16_t *depth, const float *xy_table, const uint8_t *rgb, int64_t n_px,
                       int32_t frames, int32_t flags, double gate, float *pts, float *col, int32_t *idx,
                       int32_t *d_count, void *ws, size_t ws_bytes, void *stream);

/* ---- container operations (Open3D surface used by the path) ------------------ */

/* a17: pcd.transform(T) (preprocessing/data.py:48) -- out = R p + t (fp64 fma chain, stored f32).
 * normals (optional, in place rotation only).  in == out allowed. */
int kpx_transform(const float *pts, int64_t n, const double *h_T, float *out, void *stream);
int kpx_rotate(const float *nrm, int64_t n, const double *h_T, float *out, void *stream);
/* a5: align_skeletons / transform_joints (preprocessing/extractor.py:113-116,
 * utils/processing.py:357-383): out = x @ inv(R) + t on f64 (rows,3) joints; h_A = inv(R) row-major 3x3. */
int kpx_joints_affine_f64(const double *x, int64_t rows, const double *h_A, const double *h_t, double *out,
                          void *stream);

/* select_by_index (floor_removal.py:50,69,71,72): selection of up to three (n,3) f32 attributes.
 * [O3D] SelectByIndex marks the listed points in a mask and emits the marked (or, inverted, the unmarked) points in
 * ascending original order: duplicates collapse, the order of the list does not matter.
 * KPX_SELECT_GATHER (0): out[k] = in[idx[k]] for k < n_idx -- equal to Open3D for ascending duplicate-free lists (the
 *                        keep lists / inlier lists this library produces, np.argwhere results), one gather, no count;
 * KPX_SELECT_INVERT (1): ascending complement of idx; d_count receives its length;
 * KPX_SELECT_MASK   (2): the mask semantics for arbitrary lists (unsorted, repeated); d_count receives the length. */
#define KPX_SELECT_GATHER 0
#define KPX_SELECT_INVERT 1
#define KPX_SELECT_MASK 2
size_t kpx_select_workspace_bytes(int64_t n);
int kpx_select_by_index(const float *a0, const float *a1, const float *a2, int64_t n, const int32_t *idx,
                        int64_t n_idx, int32_t invert, float *o0, float *o1, float *o2, int32_t *d_count,
                        void *ws, size_t ws_bytes, void *stream);

/* The same gather (KPX_SELECT_GATHER) that also leaves the bounds of the points it wrote -- min x, y, z, max x, y, z as six doubles at
 * d_bbox6 -- so that the selected cloud's next consumer (kpx_slab_split_bounded: floor_removal.py:64-66 takes max(y) of the cloud it just
 * filtered) needs no pass of its own for them.  n_idx > 0; workspace: kpx_select_workspace_bytes(n). */
int kpx_select_by_index_bounds(const float *a0, const float *a1, const float *a2, int64_t n, const int32_t *idx, int64_t n_idx,
                               float *o0, float *o1, float *o2, double *d_bbox6, void *ws, size_t ws_bytes, void *stream);
/* Open3D get_min_bound / get_max_bound, numpy's pcd_points[:, 1].max() (floor_removal.py:65-66): min x, y, z, max x, y, z of an
 * (n,3) f32 cloud as six doubles in device memory.  n > 0. */
size_t kpx_bounds_workspace_bytes(void);
int kpx_bounds(const float *pts, int64_t n, double *d_bbox6, void *ws, size_t ws_bytes, void *stream);

/* Stable sort of (key, value) pairs by the low end_bit (1..32) bits of the key -- the ordering step behind a7 / a8 / a11 (voxel keys,
 * grid cells, Morton codes; the reference gets it from Open3D's hash maps and np.argsort).  Inputs are not modified and may not
 * alias the outputs. */
size_t kpx_sort_pairs_u32_workspace_bytes(int64_t n);
int kpx_sort_pairs_u32(const uint32_t *keys_in, const int32_t *vals_in, int64_t n, int32_t end_bit, uint32_t *keys_out,
                       int32_t *vals_out, void *ws, size_t ws_bytes, void *stream);

/* a19: pcd_above_plane (floor_removal.py:39-51): indices of points with a x + b y + c z + d < 0
 * (evaluated left to right in fp64, as the reference's Python loop). */
int kpx_halfspace_select(const float *pts, int64_t n, const double *h_plane, int32_t *idx, int32_t *d_count,
                         void *ws, size_t ws_bytes, void *stream);
/* floor_removal.py:64-66: y >= max(y) - slab  -> lower indices, y < max(y) - slab -> upper indices. */
int kpx_slab_split(const float *pts, int64_t n, double slab, int32_t *lower_idx, int32_t *d_lower,
                   int32_t *upper_idx, int32_t *d_upper, void *ws, size_t ws_bytes, void *stream);
/* The same split for a cloud whose bounds are already in device memory (d_bbox6 as written by kpx_bounds or
 * kpx_select_by_index_bounds; only max y, d_bbox6[4], is read): one pass over the points instead of two. */
int kpx_slab_split_bounded(const float *pts, int64_t n, double slab, const double *d_bbox6, int32_t *lower_idx, int32_t *d_lower,
                           int32_t *upper_idx, int32_t *d_upper, void *ws, size_t ws_bytes, void *stream);

/* ---- filter stage ------------------------------------------------------------------------------ */

/* a7: PointCloud.voxel_down_sample(v) (preprocessing/filtering.py:23, registration.py:8,100,101).
 * origin = min_bound - v/2, index = floor((p-origin)/v), per-voxel mean of points / colours /
 * normals (normals renormalised).  Output order: ascending (ix,iy,iz) (documented deviation from
 * Open3D's hash order).  col/nrm and their outputs may be NULL. */
size_t kpx_voxel_workspace_bytes(int64_t n);
int kpx_voxel_downsample(const float *pts, const float *col, const float *nrm, int64_t n, double voxel,
                         float *opts, float *ocol, float *onrm, int32_t *d_count, void *ws, size_t ws_bytes,
                         void *stream);

/* The same for `count` independent clouds (the per-device clouds of one frame set, preprocessing/data.py:129-143 /
 * registration.py:24-29 down-sample every device's cloud): the clouds are processed side by side on internal
 * lanes; `stream` continues after all of them.  h_*: host arrays of device pointers (h_col / h_ocol may be NULL),
 * d_counts: i32 [count] on the device. */
size_t kpx_voxel_batch_workspace_bytes(int32_t count, const int64_t *h_n);
int kpx_voxel_downsample_batch(int32_t count, const float *const *h_pts, const float *const *h_col, const int64_t *h_n,
                               double voxel, float *const *h_opts, float *const *h_ocol, int32_t *d_counts, void *ws,
                               size_t ws_bytes, void *stream);

/* a17 + a7 in one pass -- the fuse of the frame loop, preprocessing/data.py:44-61: cloud c is moved by its registration
 * h_T[c] (pcd.transform(T), :48; 4x4 row-major f64 on the host, identity for the master), the clouds are stacked in the given
 * order (np.vstack, :55-58) and the stack is voxel_down_sample'd (filter_outliers, filtering.py:23).  The stacked cloud is
 * never materialised: the reference holds it in float64, so the minimum bound, the voxel index and the per-voxel sums use the
 * fp64 value of the moved point recomputed from the float32 sensor point -- voxel membership is the float64 path's, which a
 * float32 copy of the moved points does not guarantee.  Up to 16 clouds; h_col (and ocol) may be NULL; colours must be on all
 * non-empty clouds or on none.  opts / ocol: f32 [sum n][3] worst case; d_count = voxels.  Output order as kpx_voxel_downsample. */
size_t kpx_fuse_voxel_workspace_bytes(int64_t total_points);
int kpx_fuse_voxel_downsample(int32_t count, const float *const *h_pts, const float *const *h_col, const int64_t *h_n,
                              const double *h_T, double voxel, float *opts, float *ocol, int32_t *d_count, void *ws,
                              size_t ws_bytes, void *stream);

/* a8: PointCloud.remove_statistical_outlier(nb_neighbors, std_ratio) (filtering.py:24,
 * floor_removal.py:73, utils/processing.py:309).  keep_idx ascending, d_count = kept,
 * d_stats f64 [3] = (mean, std, threshold), d_avg f64 [n] (optional) = per-point mean kNN distance.
 * nb_neighbors <= KPX_SOR_MAX_K.  Up to KPX_SOR_LDS_K the neighbour heaps of the queries the wave passes leave over live in LDS;
 * beyond, in the workspace (global memory: slower, the same result) -- the reference takes any value (preprocessing/filtering.py:12-17). */
#define KPX_SOR_MAX_K 4096
#define KPX_SOR_LDS_K 288
size_t kpx_sor_workspace_bytes(int64_t n, int32_t nb_neighbors);
int kpx_sor(const float *pts, int64_t n, int32_t nb_neighbors, double std_ratio, int32_t *keep_idx,
            int32_t *d_count, double *d_stats, double *d_avg, void *ws, size_t ws_bytes, void *stream);

/* `cl, ind = pcd.remove_statistical_outlier(nb_neighbors, std_ratio)` (preprocessing/filtering.py:33, data.py:61) with BOTH results:
 * the keep list as kpx_sor and, in the same pass, the kept rows of pts (and of one more (n,3) attribute array, e.g. colours; NULL for
 * none) -- kpx_sor + kpx_select_by_index without the count read-back in between.  Workspace: kpx_sor_workspace_bytes. */
int kpx_sor_select(const float *pts, const float *attr, int64_t n, int32_t nb_neighbors, double std_ratio, float *out_pts,
                   float *out_attr, int32_t *keep_idx, int32_t *d_count, double *d_stats, void *ws, size_t ws_bytes, void *stream);

/* The same filter sharded over GPUs (the fused-cloud filter_outliers of preprocessing/data.py:61 when every GPU holds the
 * fused cloud): the grid order of the cloud is the same on every GPU, each searches only the queries at
 * cell-sorted positions [q_begin, q_end) -- a spatial slab -- and writes their mean distances, in that order, to
 * d_avg_sorted f64 [q_end - q_begin]; d_order i32 [n] (optional) receives the cell-sorted position -> point index map.
 * After the slabs have been exchanged (one all-gather), kpx_sor_finish(avg of all n positions, d_order) computes mean / std /
 * threshold and the ascending keep list with the same kernels and the same reduction order as kpx_sor: the result is
 * bit-identical to the one-GPU call.  Workspace of kpx_sor_partial: kpx_sor_workspace_bytes(n, nb_neighbors). */
int kpx_sor_partial(const float *pts, int64_t n, int32_t nb_neighbors, int64_t q_begin, int64_t q_end, double *d_avg_sorted,
                    int32_t *d_order, void *ws, size_t ws_bytes, void *stream);
size_t kpx_sor_finish_workspace_bytes(int64_t n);
int kpx_sor_finish(const double *d_avg_sorted, const int32_t *d_order, int64_t n, double std_ratio, int32_t *keep_idx,
                   int32_t *d_count, double *d_stats, double *d_avg, void *ws, size_t ws_bytes, void *stream);

/* estimate_normals(KDTreeSearchParamHybrid(radius, max_nn)) (preprocessing/registration.py:9-13):
 * neighbours = up to max_nn nearest with d2 < radius^2; < 3 neighbours -> (0,0,1); else the
 * eigenvector of the smallest eigenvalue of the neighbourhood covariance.  max_nn <= KPX_NORMALS_MAX_NN (beyond KPX_NORMALS_LDS_NN the
 * fall-back heaps live in the workspace instead of LDS; preprocessing/registration.py:7-21 takes any value). */
#define KPX_NORMALS_MAX_NN 4096
#define KPX_NORMALS_LDS_NN 128
size_t kpx_normals_workspace_bytes(int64_t n, int32_t max_nn);
int kpx_estimate_normals(const float *pts, int64_t n, double radius, int32_t max_nn, float *normals,
                         void *ws, size_t ws_bytes, void *stream);

/* a21: PointCloud.segment_plane(distance_threshold, ransac_n, num_iterations) (floor_removal.py:70).
 * Seeded (Philox4x32-10) so results are reproducible; the reference's is unseeded.
 * d_plane f64 [4], inlier_idx ascending, d_count = #inliers. */
size_t kpx_segment_plane_workspace_bytes(int64_t n, int32_t ransac_n, int32_t num_iterations);
int kpx_segment_plane(const float *pts, int64_t n, double distance_threshold, int32_t ransac_n,
                      int32_t num_iterations, double probability, uint64_t seed, double *d_plane,
                      int32_t *inlier_idx, int32_t *d_count, void *ws, size_t ws_bytes, void *stream);

/* ---- register stage ---------------------------------------------------------------------------- */

/* One correspondence search of registration_icp (manual_pointcloud_registration.py:96-98,
 * preprocessing/registration.py:78-84): for every source point transformed by d_T (device f64 [16]),
 * the nearest target point (exact; ties to the lowest target index) -- fp64 MFMA distance tiles in the K=4
 * augmented form with a fused running argmin, multiplied only where the tile's bounding box can hold a nearer
 * point (Morton-sorted operands).  idx i32 [n], d2 f64 [n] (direct squared distance of the chosen pair). */
size_t kpx_nn_workspace_bytes(int64_t n_src, int64_t n_tgt);
int kpx_nn_search(const float *src, int64_t n_src, const float *tgt, int64_t n_tgt, const double *d_T,
                  int32_t *idx, double *d2, void *ws, size_t ws_bytes, void *stream);

/* TransformationEstimationPointToPoint().compute_transformation(src, tgt, corr)
 * (manual_pointcloud_registration.py:90-91): Umeyama/Kabsch without scale on explicit pairs.
 * corr i32 [n_corr*2] = (source index, target index); d_T f64 [16]. */
size_t kpx_kabsch_workspace_bytes(int64_t n_corr);
int kpx_kabsch(const float *src, const float *tgt, const int32_t *corr, int64_t n_corr, double *d_T, void *ws,
               size_t ws_bytes, void *stream);

/* registration_icp(source, target, max_dist, init, estimation, criteria) -- the whole loop runs on
 * the device without host synchronisation.  mode 0 = point-to-point (Kabsch, a16), 1 = point-to-plane
 * (needs tgt_normals, a14).  h_init: f64 [16] host.  d_result f64 [20]: T (16), fitness, inlier_rmse,
 * iterations done, correspondence count.  idx/d2 (optional) receive the last correspondence set: the
 * nearest target of every source point, or idx = -1 / d2 = +inf where no target lies within max_dist
 * (such points are not correspondences).
 * poll_interval == 0: every iteration is enqueued up front and later launches return at once after
 * convergence (fully asynchronous).  poll_interval = p > 0: the host reads the device `done` flag every
 * p iterations (one 4-byte copy + stream sync) and stops enqueuing. */
#define KPX_ICP_POINT_TO_POINT 0
#define KPX_ICP_POINT_TO_PLANE 1
/* kpx_frame_params.icp_mode only (round 5): no registration inside the frame -- the reference registers on its first frame only
 * (preprocessing/data.py:35-41: `if i == 0: ...register...`, every later frame reuses registration_transformations): h_init ARE the
 * transforms, the frame is extract (mask + gate + colour) -> transform + vstack + voxel -> remove_statistical_outlier.  Not taken by
 * kpx_frame_step_sharded. */
#define KPX_ICP_FIXED 2
size_t kpx_icp_workspace_bytes(int64_t n_src, int64_t n_tgt);
int kpx_icp(const float *src, int64_t n_src, const float *tgt, const float *tgt_normals, int64_t n_tgt,
            double max_dist, const double *h_init, int32_t mode, int32_t max_iteration, double relative_fitness,
            double relative_rmse, int32_t poll_interval, double *d_result, int32_t *idx, double *d2, void *ws,
            size_t ws_bytes, void *stream);

/* ---- global registration (rows a11-a13) ------------------------------------------------- */

/* compute_fpfh_feature(pcd, KDTreeSearchParamHybrid(radius, max_nn)) (preprocessing/registration.py:15-20).
 * fpfh: f64 [n][33] (Open3D's Feature.data is the transpose, (33, n)).  Needs normals.  max_nn <= KPX_NORMALS_MAX_NN. */
size_t kpx_fpfh_workspace_bytes(int64_t n, int32_t max_nn);
int kpx_fpfh(const float *pts, const float *normals, int64_t n, double radius, int32_t max_nn, double *fpfh, void *ws,
             size_t ws_bytes, void *stream);

/* 1-nearest neighbour of every row of fa among the rows of fb in the 33-D feature space (the matching stage of
 * registration_ransac_based_on_feature_matching, registration.py:50-57); ties to the lowest index. */
size_t kpx_feature_nn_workspace_bytes(int64_t na, int64_t nb);
int kpx_feature_nn(const double *fa, int64_t na, const double *fb, int64_t nb, int32_t *idx, void *ws, size_t ws_bytes,
                   void *stream);

/* RegistrationRANSACBasedOnCorrespondence: ransac_n = 3 correspondences per hypothesis (Philox-seeded, with
 * replacement), Umeyama without scale, CorrespondenceCheckerBasedOnEdgeLength(edge_similarity) and
 * ...BasedOnDistance(max_dist), validation by the full nearest-neighbour correspondence count within max_dist,
 * RANSACConvergenceCriteria(max_iteration, confidence).  Synchronous (the exit test needs each batch's result).
 * corres: i32 [n_corres][2] (source, target) on the device.
 * h_result (host) f64 [20]: T (16) | fitness | inlier_rmse | iterations run | validations. */
size_t kpx_ransac_workspace_bytes(int64_t n_src, int64_t n_tgt);
int kpx_ransac_corres(const float *src, int64_t n_src, const float *tgt, int64_t n_tgt, const int32_t *corres, int64_t n_corres,
                      double max_dist, int32_t ransac_n, double edge_similarity, int32_t max_iteration, double confidence,
                      uint64_t seed, double *h_result, void *ws, size_t ws_bytes, void *stream);

/* Several registrations onto ONE shared target (preprocessing/data.py:144-161 registers every sub device onto
 * the master cloud).  The target operand is prepared once; the problems' iterations are queued round-robin on
 * `stream` and each problem's convergence flag is polled through a side stream that waits only for that
 * problem, so the host round trip is hidden behind the other problems' sweeps (kernels never overlap).
 * h_src / h_n_src: host arrays of `count` entries; h_init: count x 16; d_results: count x 20 (as kpx_icp). */
size_t kpx_icp_batch_workspace_bytes(int32_t count, const int64_t *h_n_src, int64_t n_tgt);
int kpx_icp_batch(int32_t count, const float *const *h_src, const int64_t *h_n_src, const float *tgt,
                  const float *tgt_normals, int64_t n_tgt, double max_dist, const double *h_init, int32_t mode,
                  int32_t max_iteration, double relative_fitness, double relative_rmse, double *d_results, void *ws,
                  size_t ws_bytes, void *stream);

/* The correspondence searches above have two interchangeable implementations with identical results: the culled
 * sweep (default) and the all-pairs sweeps it replaced (fp64 MFMA + float32 screening), kept as an independent
 * cross-check and for A/B measurements.  kpx_nn_engine(e) selects one for the following calls and returns the
 * engine that was active (e < 0: query only).  Initial value: environment KPX_NN_ENGINE=dense|culled. */
#define KPX_NN_ENGINE_CULLED 0
#define KPX_NN_ENGINE_DENSE 1
#define KPX_NN_ENGINE_DENSE_FP64 2 /* the all-pairs engine with every search on the fp64 MFMA sweep (no float32 screening) */
int kpx_nn_engine(int32_t engine);

/* ---- the sampler / normaliser after the path ------------------------------------ */
/* select_points_randomly (utils/processing.py:259-275): number_of_points of the cloud without replacement.  The
 * reference draws np.random.choice from NumPy's unseeded global generator; here point i gets the 64-bit key
 * Philox4x32-10(ctr = (i_lo, i_hi, 'SAMP', 0), key = seed) words (1:0) and the sample is the k points with the
 * smallest (key, i), in that order -- a uniformly random ordered k-subset.  k > n is an error (NumPy raises
 * ValueError).  out_pts (k,3) f32 and/or out_idx (k) i32 on the device; either may be NULL. */
size_t kpx_sample_workspace_bytes(int64_t n);
int kpx_sample_points(const float *pts, int64_t n, int64_t k, uint64_t seed, float *out_pts, int32_t *out_idx, void *ws,
                      size_t ws_bytes, void *stream);

/* PointCloud.get_oriented_bounding_box() for `count` clouds of n points each, stored back to back (the batch arrays of
 * utils/normalization.py:38-42, 74-77, 105-106; one cloud: utils/processing.py:341-344).  Open3D's CreateFromPoints:
 * convex hull, PCA of the hull vertices (eigenvectors by descending eigenvalue, third = first x second; here each of
 * the first two has its largest component positive), box of R^T (v - mean).  pts: f32 (pts_f64 = 0) or f64 (1)
 * [count][n][3] on the device.  d_obb f64 [count][16]: R row-major (9) | centre (3) | extent (3) | number of hull
 * vertices, or -1 (fewer than 3 distinct points / all on one line), -2 (hull did not close), -3 (flat hull: Qhull
 * raises for these).  d_is_vertex: optional u8 [count][n] hull-vertex flags.  Asynchronous. */
size_t kpx_obb_workspace_bytes(int32_t count, int64_t n);
int kpx_obb_batch(const void *pts, int32_t pts_f64, int32_t count, int64_t n, double *d_obb, uint8_t *d_is_vertex, void *ws,
                  size_t ws_bytes, void *stream);

/* The normalisations, applied to `count` groups of `rows` f64 triples (points or joints) with the boxes of kpx_obb_batch:
 *   KPX_NORM_OBB            (x @ M + centre) / max(extent), M = get_rotation_matrix_from_yxz([0, pi, 0])
 *                           (obb_normalization_batch, utils/normalization.py:16-64, as written there)
 *   KPX_NORM_OBB_ROT_TRANS  (x - centre) @ R @ M, M = Rz(90 deg)   (obb_rotation_translation_batch, :67-97)
 *   KPX_NORM_TRANSLATE      x - centre                              (translation_normalization_batch, :100-126)
 *   KPX_NORM_OBB_ROT        (x - centre) @ R                        (obb_normalization, utils/processing.py:329-354)
 * h_M: host 3x3 row-major constant matrix of the mode (NULL where unused). */
#define KPX_NORM_OBB 0
#define KPX_NORM_OBB_ROT_TRANS 1
#define KPX_NORM_TRANSLATE 2
#define KPX_NORM_OBB_ROT 3
int kpx_normalize_batch(const double *x, int32_t count, int64_t rows, const double *d_obb, int32_t mode, const double *h_M,
                        double *out, void *stream);

/* a15: coloured ICP (execute_colored_ICP_registration, preprocessing/registration.py:89-114), SURVEY 8f rank 4.
 * kpx_color_gradient = [O3D] InitializePointCloudForColoredICP: per target point the least-squares gradient of the
 * intensity (r + g + b) / 3 in its tangent plane over the hybrid neighbourhood (radius, max_nn; Open3D passes
 * 2 x max_correspondence_distance and 30); grad f64 [n][3] on the device.
 * kpx_colored_icp = [O3D] registration_colored_icp: the registration_icp loop (correspondences, fitness, inlier rmse,
 * convergence as kpx_icp) with TransformationEstimationForColoredICP(lambda_geometric, Open3D default 0.968) as the
 * update: per pair a geometric row sqrt(lambda) ((s - t).n) and a photometric row sqrt(1 - lambda) (I_s - I_t - g.(s' - t)).
 * colours f32 [n][3]; d_result as kpx_icp. */
size_t kpx_color_gradient_workspace_bytes(int64_t n, int32_t max_nn);
int kpx_color_gradient(const float *pts, const float *normals, const float *colors, int64_t n, double radius, int32_t max_nn,
                       double *grad, void *ws, size_t ws_bytes, void *stream);
size_t kpx_colored_icp_workspace_bytes(int64_t n_src, int64_t n_tgt);
int kpx_colored_icp(const float *src, const float *src_colors, int64_t n_src, const float *tgt, const float *tgt_colors,
                    const float *tgt_normals, const double *tgt_gradient, int64_t n_tgt, double max_dist, const double *h_init,
                    double lambda_geometric, int32_t max_iteration, double relative_fitness, double relative_rmse,
                    int32_t poll_interval, double *d_result, void *ws, size_t ws_bytes, void *stream);

/* fuse_skeletons_gradient (utils/skeleton_fusion.py:21-74), SURVEY 8f rank 4: gradient- and centroid-weighted average of the
 * joints seen by three cameras.  skeletons f64 [cams][frames][joints][3] on the device; the first initial_frame (reference: 20)
 * frames are the mean over ALL cameras, later frames weight the FIRST THREE cameras (as the reference does) with
 * w_c = 1 / (|p_c - fused[f-1]|^alpha |p_c - centroid|^beta).  out f64 [frames][joints][3]. */
int kpx_fuse_skeletons(const double *skeletons, int32_t cams, int64_t frames, int32_t joints, double alpha, double beta,
                       int32_t initial_frame, double *out, void *stream);

/* ---- the frame loop as one native call ------------------------------------------------------------------------------- */
/* One synchronised frame set of `sensors` devices held by this GPU: the body of DataProcessor's frame loop
 * (preprocessing/data.py:35-61) with the registration of data.py:127-161 on the same depth images folded in --
 *   depth -> full cloud -> voxel_down_sample(reg_voxel) per sensor; normals of the master's (sensor 0);
 *   execute_point_to_plane_registration of every sub sensor onto the master (h_init: the sensors - 1 initial 4x4, host);
 *   depth + person mask (rgb with the background zeroed, data.py:169) + depth gate -> the person clouds;
 *   pcd.transform(T_i) + np.vstack + voxel_down_sample(filt_voxel) in one fp64 pass; remove_statistical_outlier(filt_k, filt_ratio).
 * Host orchestration in C++ over the entry points above (the Python mirror of this loop is kinectpy_amd/pipeline.py): one
 * call per frame, synchronous (the data-dependent counts are read back on `stream` inside); when it returns the selection
 * kernel that writes out_pts / out_col is queued on `stream`.  depth u16 [sensors][n_px], rgb u8 [sensors][n_px][3], xy_table
 * f32 [n_px][2] on the device; out_pts / out_col f32 [sensors * n_px][3] worst case; h_count: points written; h_T f64
 * [sensors][16] (identity for the master); h_info (optional, 64 ints): [0..15] down-sampled points per sensor, [16..31] masked
 * points per sensor, [32..47] ICP iterations per sensor, [48] voxels of the fused cloud. */
typedef struct {
    double reg_voxel;          /* 35   preprocessing/registration.py:35,69 */
    double icp_max_dist;       /* 100  registration.py:75 */
    double filt_voxel;         /* filter_outliers voxel_size (filtering.py:16) */
    double filt_ratio;         /* filter_outliers std_ratio */
    double gate;               /* 750  data.py:170-171 */
    int32_t normals_nn;        /* 40   registration.py:24 */
    int32_t icp_mode;          /* KPX_ICP_POINT_TO_PLANE (registration.py:83) */
    int32_t icp_max_iteration; /* 30   Open3D's default criteria */
    int32_t filt_k;            /* filter_outliers nb_neighbors */
} kpx_frame_params;
size_t kpx_frame_step_workspace_bytes(int32_t sensors, int64_t n_px);
int kpx_frame_step(const uint16_t *depth, const uint8_t *rgb, const float *xy_table, int64_t n_px, int32_t sensors,
                   const double *h_init, const kpx_frame_params *params, float *out_pts, float *out_col, int32_t *h_count,
                   double *h_T, int32_t *h_info, void *ws, size_t ws_bytes, void *stream);
/* The same frame handed over in HOST memory (pinned for an asynchronous copy) -- SURVEY 8(d)'s interval "depth frame resident in
 * host pinned memory -> fused registered cloud on the GPU"; the reference's loop starts from the files it has just read
 * (preprocessing/data.py:87-124).  h_depth u16 [sensors][n_px], h_rgb u8 [sensors][n_px][3] on the host; both are copied into
 * staging buffers at the head of `ws` on `stream` (no allocation, no extra synchronisation), then kpx_frame_step runs. */
size_t kpx_frame_step_host_workspace_bytes(int32_t sensors, int64_t n_px);
int kpx_frame_step_host(const uint16_t *h_depth, const uint8_t *h_rgb, const float *xy_table, int64_t n_px, int32_t sensors,
                        const double *h_init, const kpx_frame_params *params, float *out_pts, float *out_col, int32_t *h_count,
                        double *h_T, int32_t *h_info, void *ws, size_t ws_bytes, void *stream);

/* ---- the frame loop over several GPUs (preprocessing/data.py:96-122 loops the devices independently, :44-61 fuses and
 * filters, :127-161 registers every sub device onto the master) -------------------------------------------------------------------
 * One PROCESS per GPU, sensor g on GPU g (fewer GPUs than sensors: contiguous blocks in rank order).  The host side of a frame runs
 * in C++; its three collectives -- master cloud broadcast, all-gather of the masked clouds + registration results, all-gather of
 * the fused filter's slab distances -- are issued from here on the frame's stream.
 *
 * kpx_comm: one communicator over all ranks (one per frame slot when several frames are in flight).  Transports:
 *   RCCL        kpx_rccl_load(path of librccl.so, NULL = the loader's search path) once per process; rank 0 draws a 128-byte id
 *               (kpx_rccl_unique_id) and hands it to the other ranks by any means (e.g. torch.distributed.broadcast_object_list);
 *               every rank then calls kpx_comm_create_rccl (collective, like ncclCommInitRank);
 *   callbacks   kpx_comm_create_callbacks: the caller moves the bytes (host-staged gloo, in-process ranks): rehearsals without RCCL.
 * The callbacks get device pointers and the frame's stream; they return 0 on success. */
typedef struct kpx_comm kpx_comm;
typedef int (*kpx_bcast_fn)(void *user, void *d_buf, size_t bytes, int32_t root, void *stream);
typedef int (*kpx_allgather_fn)(void *user, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream);
int kpx_rccl_load(const char *path);
int kpx_rccl_unique_id(void *id128);
int kpx_comm_create_rccl(const void *id128, int32_t rank, int32_t world, kpx_comm **out);
int kpx_comm_create_callbacks(int32_t rank, int32_t world, kpx_bcast_fn bcast, kpx_allgather_fn allgather, void *user, kpx_comm **out);
/* Replay transport (measurement aid of bench.py --emulate-world): ONE rank of a `world`-rank job on one GPU, the peers' messages taken
 * from recordings of a real world-rank run.  d_payloads / bytes: [frames][3][world] device pointers / sizes of what rank r sent in
 * collective c (0 master broadcast -- root's entry only --, 1 cloud exchange, 2 slab all-gather) of recorded frame f.  Call n of the
 * communicator is collective n % 3 of frame (first_frame + stride * (n / 3)) % frames.  Message sizes must be the recording's (both
 * runs with KPX_SHARD_FIXED_CAP=1).  Replaces nothing of the reference: it prices a rank's share of preprocessing/data.py:35-61. */
int kpx_comm_create_replay(int32_t rank, int32_t world, int32_t frames, int32_t first_frame, int32_t stride, int32_t per_frame,
                           const void *const *d_payloads, const size_t *bytes, kpx_comm **out);      /* per_frame: 3, or 2 for frames without collective 2 */
int kpx_comm_destroy(kpx_comm *comm);
int kpx_comm_rank(const kpx_comm *comm);
int kpx_comm_world(const kpx_comm *comm);
/* in-place broadcast of device memory from `root` / all-gather into d_recv [world][bytes_per_rank], on `stream` */
int kpx_comm_broadcast(kpx_comm *comm, void *d_buf, size_t bytes, int32_t root, void *stream);
int kpx_comm_allgather(kpx_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream);
/* utility for callback transports: copy between any two of host / device memory on `stream`; wait != 0 also waits for the stream */
int kpx_copy_bytes(void *dst, const void *src, size_t bytes, void *stream, int32_t wait);

/* kpx_order: ONE issue order for the collectives of the frames in flight on a rank.  Frames in flight run on host threads; a
 * collective kernel spins on the device until its peers arrive, so every rank must enqueue the collectives of its frames in the
 * same order whatever the timing of its threads: stage s (0 broadcast, 1 exchange, 2 slab all-gather) of frame f has the key 3 f
 * (s = 0) or 3 (f + depth - 1) + s, and a collective is issued only when every smaller key of the submitted frames is done.  The
 * main thread calls submit (a frame enters; -> its number), block(frame) while it waits for that frame (block(-1) afterwards) and
 * finish(frame) when the frame is over; kpx_frame_step_sharded takes its turns itself (turn_begin / turn_end / skip, exported for
 * tests).  A NULL order means one frame at a time. */
typedef struct kpx_order kpx_order;
int kpx_order_create(int32_t depth, kpx_order **out);
int kpx_order_destroy(kpx_order *order);
int kpx_order_submit(kpx_order *order, int64_t *frame);
int kpx_order_block(kpx_order *order, int64_t frame);
int kpx_order_turn_begin(kpx_order *order, int64_t frame, int32_t stage);
int kpx_order_turn_end(kpx_order *order, int64_t frame, int32_t stage);
int kpx_order_skip(kpx_order *order, int64_t frame, int32_t stage);
int kpx_order_finish(kpx_order *order, int64_t frame);
int kpx_order_log(kpx_order *order, int64_t *out, int64_t cap, int64_t *count);

/* kpx_frame_step for the rank's share of the rig.  depth / rgb: the images of THIS rank's sensors (sensor order), on the device or
 * (host_input != 0) in host memory; h_init: the sensors - 1 initial transforms of ALL sub sensors; fused_filter 0 = the filter on
 * the fused cloud is sharded over the ranks by slabs of its grid order (every rank ends with the filtered frame, bit-identical to
 * the one-GPU filter), 1 = rank 0 filters alone (the others return *h_count = 0), 2 = frame f's fused pass and filter run on rank
 * f mod world alone (f: the kpx_order's frame number; the owner returns the frame, the others 0 rows).  h_T f64 [sensors][16] and h_info describe the
 * WHOLE rig on every rank (they travel in the exchange headers).  Message capacities adapt to the slot's previous frame; when a
 * frame outgrows one, every rank returns KPX_RETRY (1) and the caller runs the frame again (a new frame number under a kpx_order).
 * Workspace: kpx_frame_step_sharded_workspace_bytes(sensors, rank, world, n_px, host_input). */
size_t kpx_frame_step_sharded_workspace_bytes(int32_t sensors, int32_t rank, int32_t world, int64_t n_px, int32_t host_input);
int kpx_frame_step_sharded(kpx_comm *comm, kpx_order *order, int64_t frame, const uint16_t *depth, const uint8_t *rgb, int32_t host_input,
                           const float *xy_table, int64_t n_px, int32_t sensors, const double *h_init, const kpx_frame_params *params,
                           int32_t fused_filter, float *out_pts, float *out_col, int32_t *h_count, double *h_T, int32_t *h_info, void *ws,
                           size_t ws_bytes, void *stream);

/* ---- frames in flight, scheduled natively (round 5) -------------------------------------------------
 * The reference's frame loop is sequential (preprocessing/data.py:35-61); consecutive frames are independent, and one frame is a
 * chain of ~100 short dispatches with four read-backs, so `depth` frames run side by side: kpx_stream owns `depth` worker threads
 * (C++, inside the library: no interpreter between a frame's end and the next one's start), each with its own HIP stream and its
 * own slice of the caller's workspace (kpx_stream_workspace_bytes).  kpx_stream_submit hands a frame over and returns at once
 * (KPX_ERR_INVALID when kpx_stream_capacity frames are queued); kpx_stream_pop blocks until the OLDEST frame in flight is done and returns
 * its status, its point count (out_pts / out_col given at submit: rows [0, *h_count) are valid when pop returns, on any stream),
 * transforms (f64 [sensors][16]) and info words (int32 [64], as kpx_frame_step).  depth / rgb of a frame: device memory, or pinned
 * host memory with host_input != 0 (the copy then runs on the slot's stream, inside the frame); they must stay valid until the
 * frame has been popped.  submit / pop / destroy: one calling thread.
 * comms == NULL: one GPU (kpx_frame_step / kpx_frame_step_host per frame).  comms = `depth` communicators, one per slot: the
 * rank's share of the rig through kpx_frame_step_sharded, the collectives of the frames in flight in ONE issue order on every rank
 * (a kpx_order owned by the stream); a frame that outgrew its messages on every rank alike is run again inside kpx_stream_pop. */
typedef struct kpx_stream kpx_stream;
size_t kpx_stream_workspace_bytes(int32_t sensors, int32_t rank, int32_t world, int64_t n_px, int32_t depth);
int kpx_stream_create(const float *xy_table, int64_t n_px, int32_t sensors, const double *h_init, const kpx_frame_params *params,
                      int32_t depth, kpx_comm *const *comms, int32_t fused_filter, void *ws, size_t ws_bytes, kpx_stream **out);
int kpx_stream_submit(kpx_stream *stream, const void *depth, const void *rgb, int32_t host_input, float *out_pts, float *out_col);
int kpx_stream_pop(kpx_stream *stream, int32_t *h_count, double *h_T, int32_t *h_info);
int kpx_stream_pending(const kpx_stream *stream);
int kpx_stream_capacity(const kpx_stream *stream);    /* frames kpx_stream_submit takes before a pop: 2 x depth on one GPU (one queue, any free worker
                                                         takes the oldest frame), depth with communicators (frame j runs on slot j % depth on every rank) */
int kpx_stream_destroy(kpx_stream *stream);
/* h_out4: [0] frames finished by the stream's workers, [1] 1 = the frames' registrations run through the device's ICP engine (one host
 * thread and one HIP stream carry the point-to-plane / point-to-point iterations of EVERY frame in flight in one launch per tick --
 * preprocessing/registration.py:78-84 as called from data.py:144-161, for all frames at once; KPX_STREAM_ENGINE=1 -- measured slower
 * than a chain of launches per frame, the default: DESIGN.md), [2] iteration launches and [3] ticks of that engine since it started (process-wide per device). */
int kpx_stream_stats(const kpx_stream *stream, uint64_t *h_out4);

/* ---- measurement hooks (bench.py) --------------------------------------------------------------- */
/* HIP-event timing of the hot kernels on the stream they are launched on.  kpx_prof_begin arms it
 * (capacity = max launches recorded); every launch of a tagged kernel is bracketed by an event pair;
 * kpx_prof_end synchronises, sums per kernel id and disarms.  h_ms / h_launches / h_work: arrays of
 * KPX_PROF_KERNELS entries (work = flops for the MFMA kernels -- for NN_LOCAL the flops of the tiles actually
 * multiplied, counted on the device -- and algorithmic bytes for the others). */
#define KPX_PROF_NN_MFMA 0
#define KPX_PROF_SOR_KNN 1
#define KPX_PROF_PLANE_SCORE 2
#define KPX_PROF_COMPACT 3
#define KPX_PROF_NN_SCREEN 4
#define KPX_PROF_NN_LOCAL 5
#define KPX_PROF_KERNELS 6
int kpx_prof_begin(int32_t capacity);
/* Time only every stride-th launch of each kernel id (default 1).  An event pair costs ~2 us of stream time; around
 * every launch of a 16 us kernel that is measurable in the end-to-end number (bench.py uses 8). */
int kpx_prof_stride(int32_t stride);
int kpx_prof_end(double *h_ms, int64_t *h_launches, double *h_work);
/* Device-side phase clock of the ICP iteration kernel (stamps while the profiler is armed), LAST launch: h_out8 = average
 * microseconds a block spends in [0] the update prologue, [1] row preparation, [2] the culled sweep, [3] the pair epilogue,
 * [4] the block sums; [5] blocks; [6] latest - earliest block start (dispatch ramp); [7] earliest start -> latest end;
 * [8..12] the slowest block of each phase; [13] the longest block lifetime; [14] blocks the launch did not sweep because they
 * provably had no partner within reach.  h_out8 holds 16 doubles. */
int kpx_prof_icp_phases(double *h_out8);
/* Per-wave rows of the same launch (4 x uint64 per wave: sweep start, sweep end in 10 ns ticks; counters = tiles multiplied |
 * tile-box fetches << 16 | operand fetches << 32 | groups kept << 48; sampled rows with a partner).  *h_count = the number of
 * waves written (<= cap_waves). */
int kpx_prof_icp_waves(uint64_t *h_out, int64_t cap_waves, int64_t *h_count);
/* The one-launch form of the culled ICP chain (kpx_icp_batch groups whose blocks fit the device; kpx_icp.hip, icp_chain_kernel): on = 1 / 0
   switches it for the calling process, on = -1 only asks; returns the previous setting (on = -2: the number of chains this process has launched).  Default: on unless KPX_ICP_CHAIN=0.  Results do
   not depend on it (tests/test_parity_gpu.py, test_icp_update_placements_and_light_skip_are_bit_identical). */
int kpx_icp_chain(int32_t on);
/* Self-check of the row certificates of the culled ICP sweep (kpx_icp_batch with KPX_ICP_CERT_CHECK=1 in the environment: rows whose
 * partner is certified unchanged -- registration_icp's correspondence step, preprocessing/registration.py:78-84 -- are searched all
 * the same and compared).  h_out8 (8 x uint64, cleared by the call): [0] rows certified, [1] rows searched, [2] certified rows whose
 * search found another partner (must be 0), [3..7] the first such row: iteration, sorted row, kept and found partner, key bits. */
int kpx_prof_icp_cert(uint64_t *h_out8);
/* Clock of the one-launch ICP chain (KPX_ICP_CHAIN_STAMPS=1): 64 iterations x 48 stamps of the 100 MHz wall clock, first registration of
   the last chain launch (slots: kpx_icp.hip, g_chain_stamp); read and reset.  A development aid like the other kpx_prof_* entries. */
int kpx_prof_icp_chain(uint64_t *h_out3072);

#ifdef __cplusplus
}
#endif
#endif /* KINECTPX_H */
