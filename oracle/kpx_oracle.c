/*
 * kpx_oracle.c -- CPU restatement (fp64, deterministic) of the KinectPy hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or executed by the
 * product (kinectpy_amd/, include/, bench.py's timed GPU leg).  Only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may call it, and only as the checker / the reported baseline.
 *
 * PARITY STATUS: **parity unpinned** for everything whose arithmetic lives inside Open3D
 * (voxel_down_sample, remove_statistical_outlier, segment_plane, registration_icp,
 * estimate_normals, compute_fpfh_feature, registration_ransac_based_on_feature_matching,
 * registration_colored_icp, get_oriented_bounding_box -- the last one's integer part, WHICH points
 * are hull vertices, is pinned against Qhull itself through scipy).  Open3D is an un-vendored, un-pinned dependency of the reference
 * (`import open3d as o3d`, /root/reference/preprocessing/registration.py:4, filtering.py:7,
 * floor_removal.py:4; API usage bounds it to >= 0.12), it is not installed here and the
 * reference has no tests or golden vectors.  Those functions restate Open3D's published
 * algorithms ([O3D], recalled) and are anchored on the reference's call sites.  The pure-NumPy
 * reference functions (load_depth, rgbd_to_pointcloud masks, equation_plane, pcd_above_plane,
 * kalman_filter, transform_joints, fuse_skeletons_gradient) ARE pinned by the known answers captured
 * from the reference itself (tests/golden/ref_kat.json, ref_skeleton_fusion.json).
 *
 * Storage contract shared with the GPU product (DESIGN.md "arithmetic contract"): clouds are
 * float32 (N,3) arrays; every decision scalar is computed in fp64 from the promoted values with
 * the operation order written here (explicit fma(), compiled with -ffp-contract=off), results
 * are rounded to float32 only where they are stored as cloud coordinates.
 *
 * Each function cites the reference file:line it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KPO_API __attribute__((visibility("default")))

/* Cloud storage type.  float (default) is the contract shared with the GPU product: clouds, colours and normals are
 * float32 arrays and every decision scalar is fp64.  -DKPO_STORAGE_F64 builds the reference-faithful variant
 * (_build/libkpx_oracle_f64.so): clouds stay float64 between stages, as the reference keeps them
 * (/root/reference/utils/io.py:29-41, preprocessing/data.py:55-56); tests/test_oracle_cpu.py measures what the float32
 * storage changes (coordinates and index decisions).  The depth unprojection is float32 in both (the [K4A] formula). */
#ifdef KPO_STORAGE_F64
typedef double real_t;
#else
typedef float real_t;
#endif
KPO_API int kpo_storage_bytes(void) { return (int)sizeof(real_t); }

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10 counter RNG (Salmon et al. 2011).  Used for every random draw so that the     */
/* CPU and GPU RANSACs consume identical streams (the reference's RANSAC is unseeded:          */
/* floor_removal.py:70).                                                                        */
/* ------------------------------------------------------------------------------------------ */
KPO_API void kpo_philox4x32(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4])
{
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ------------------------------------------------------------------------------------------ */
/* a1: depth -> XYZ int16 (the `.dat` contract consumed by utils/io.py:15-20).                 */
/* The reference has no unprojection arithmetic (extractor.py:68-80 shells out to an external  */
/* binary); this restates the Azure-Kinect-SDK xy-table formula [K4A, recalled]:               */
/*   x = (int16) floorf(xt * (float)d + 0.5f), y likewise, z = d; invalid (d==0 or NaN table)  */
/*   -> (0,0,0).  All arithmetic in float32, no contraction.                                   */
/* ------------------------------------------------------------------------------------------ */
KPO_API void kpo_xy_table_pinhole(int H, int W, float fx, float fy, float cx, float cy, float *xy)
{
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            xy[2 * ((int64_t)v * W + u) + 0] = ((float)u - cx) / fx;
            xy[2 * ((int64_t)v * W + u) + 1] = ((float)v - cy) / fy;
        }
}

KPO_API void kpo_unproject_u16(const uint16_t *depth, const float *xy, int64_t n, int16_t *xyz)
{
    for (int64_t i = 0; i < n; ++i) {
        float xt = xy[2 * i], yt = xy[2 * i + 1];
        uint16_t d = depth[i];
        if (d != 0 && !isnan(xt) && !isnan(yt)) {
            float fd = (float)d;
            float px = xt * fd; px = px + 0.5f;
            float py = yt * fd; py = py + 0.5f;
            xyz[3 * i + 0] = (int16_t)(int32_t)floorf(px);
            xyz[3 * i + 1] = (int16_t)(int32_t)floorf(py);
            xyz[3 * i + 2] = (int16_t)d;
        } else {
            xyz[3 * i + 0] = 0; xyz[3 * i + 1] = 0; xyz[3 * i + 2] = 0;
        }
    }
}

/* np.median over the z column (preprocessing/data.py:170-171): sorts, mean of the two middle  */
/* values for even n, as float64.                                                               */
static int cmp_i16(const void *a, const void *b)
{
    int16_t x = *(const int16_t *)a, y = *(const int16_t *)b;
    return (x > y) - (x < y);
}
KPO_API double kpo_median_i16(const int16_t *v, int64_t n, int64_t stride)
{
    if (n <= 0) return NAN;
    int16_t *tmp = (int16_t *)malloc((size_t)n * sizeof(int16_t));
    for (int64_t i = 0; i < n; ++i) tmp[i] = v[i * stride];
    qsort(tmp, (size_t)n, sizeof(int16_t), cmp_i16);
    double m = (n & 1) ? (double)tmp[n / 2] : 0.5 * ((double)tmp[n / 2 - 1] + (double)tmp[n / 2]);
    free(tmp);
    return m;
}

/* a3 + a4: utils/io.py:23-43 (keep = x!=0 & y!=0 & z!=0, colours /255) composed with          */
/* preprocessing/data.py:165-178 (valid_pixels: all three colour channels != 0; valid_depths:  */
/* z <= median+750 | z <= median-750  ==  z <= median+750).  Order-preserving compaction.      */
/* gate_hi is median+750 computed by the caller in float64.                                     */
KPO_API int64_t kpo_rgbd_compact(const int16_t *xyz, const uint8_t *rgb, int64_t n,
                                 int use_color_mask, int use_gate, double gate_hi,
                                 real_t *pts, real_t *col, int32_t *idx)
{
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) {
        int16_t x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        int keep = (x != 0) && (y != 0) && (z != 0);
        if (use_color_mask && rgb)
            keep = keep && rgb[3 * i] != 0 && rgb[3 * i + 1] != 0 && rgb[3 * i + 2] != 0;
        if (use_gate) keep = keep && ((double)z <= gate_hi);
        if (!keep) continue;
        if (pts) { pts[3 * k] = (real_t)x; pts[3 * k + 1] = (real_t)y; pts[3 * k + 2] = (real_t)z; }
        if (col && rgb)
            for (int c = 0; c < 3; ++c) col[3 * k + c] = (real_t)((double)rgb[3 * i + c] / 255.0);
        if (idx) idx[k] = (int32_t)i;
        ++k;
    }
    return k;
}

/* ------------------------------------------------------------------------------------------ */
/* a17 / pcd.transform (preprocessing/data.py:48): p' = R p + t.  Contract AC1:                */
/*   p'_k = fma(R_k0, x, fma(R_k1, y, fma(R_k2, z, t_k)))  in fp64, stored as float32.         */
/* T is row-major 4x4.                                                                          */
/* ------------------------------------------------------------------------------------------ */
static inline void xform3(const double *T, double x, double y, double z, double *o)
{
    for (int k = 0; k < 3; ++k)
        o[k] = fma(T[4 * k + 0], x, fma(T[4 * k + 1], y, fma(T[4 * k + 2], z, T[4 * k + 3])));
}
KPO_API void kpo_transform(const real_t *pts, int64_t n, const double *T, real_t *out)
{
    for (int64_t i = 0; i < n; ++i) {
        double o[3];
        xform3(T, (double)pts[3 * i], (double)pts[3 * i + 1], (double)pts[3 * i + 2], o);
        out[3 * i] = (real_t)o[0]; out[3 * i + 1] = (real_t)o[1]; out[3 * i + 2] = (real_t)o[2];
    }
}
/* normals rotate only (Open3D PointCloud::Transform) */
KPO_API void kpo_rotate(const real_t *nrm, int64_t n, const double *T, real_t *out)
{
    for (int64_t i = 0; i < n; ++i) {
        double x = nrm[3 * i], y = nrm[3 * i + 1], z = nrm[3 * i + 2];
        for (int k = 0; k < 3; ++k)
            out[3 * i + k] = (real_t)fma(T[4 * k + 0], x, fma(T[4 * k + 1], y, T[4 * k + 2] * z));
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a7: [O3D] PointCloud::VoxelDownSample, call sites preprocessing/filtering.py:23,            */
/* registration.py:8,100,101.  origin = min_bound - v/2; index = floor((p - origin)/v);        */
/* output = per-voxel mean.  Documented deviation: output order is ascending (ix,iy,iz)        */
/* (Open3D's is unordered_map order); accumulation order inside a voxel is ascending original  */
/* index, fp64, sequential.  Normals are averaged then normalised [O3D GetAverageNormal].      */
/* Returns number of voxels, or -1 (voxel<=0) / -2 (index overflow: > 2^21 cells on an axis).  */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t key; int64_t idx; } kv_t;
static int cmp_kv(const void *a, const void *b)
{
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->key != y->key) return (x->key > y->key) - (x->key < y->key);
    return (x->idx > y->idx) - (x->idx < y->idx);
}
KPO_API int64_t kpo_voxel_downsample(const real_t *pts, const real_t *col, const real_t *nrm,
                                     int64_t n, double voxel, real_t *opts, real_t *ocol,
                                     real_t *onrm, int32_t *ocnt)
{
    if (!(voxel > 0.0)) return -1;
    if (n == 0) return 0;
    double mn[3] = { pts[0], pts[1], pts[2] };
    for (int64_t i = 1; i < n; ++i)
        for (int a = 0; a < 3; ++a) { double v = pts[3 * i + a]; if (v < mn[a]) mn[a] = v; }
    double org[3];
    for (int a = 0; a < 3; ++a) org[a] = mn[a] - voxel * 0.5;
    kv_t *kv = (kv_t *)malloc((size_t)n * sizeof(kv_t));
    for (int64_t i = 0; i < n; ++i) {
        uint64_t key = 0;
        for (int a = 0; a < 3; ++a) {
            double r = ((double)pts[3 * i + a] - org[a]) / voxel;
            double f = floor(r);
            if (!(f >= 0.0) || f >= 2097152.0) { free(kv); return -2; }
            key = (key << 21) | (uint64_t)f;
        }
        kv[i].key = key; kv[i].idx = i;
    }
    qsort(kv, (size_t)n, sizeof(kv_t), cmp_kv);
    int64_t m = 0, i = 0;
    while (i < n) {
        int64_t j = i;
        double sp[3] = {0, 0, 0}, sc[3] = {0, 0, 0}, sn[3] = {0, 0, 0};
        while (j < n && kv[j].key == kv[i].key) {
            int64_t p = kv[j].idx;
            for (int a = 0; a < 3; ++a) {
                sp[a] += (double)pts[3 * p + a];
                if (col) sc[a] += (double)col[3 * p + a];
                if (nrm) sn[a] += (double)nrm[3 * p + a];
            }
            ++j;
        }
        double c = (double)(j - i);
        for (int a = 0; a < 3; ++a) {
            opts[3 * m + a] = (real_t)(sp[a] / c);
            if (col && ocol) ocol[3 * m + a] = (real_t)(sc[a] / c);
        }
        if (nrm && onrm) {
            double nn = sqrt(fma(sn[2], sn[2], fma(sn[1], sn[1], sn[0] * sn[0])));
            for (int a = 0; a < 3; ++a) onrm[3 * m + a] = (real_t)(nn > 0 ? sn[a] / nn : sn[a]);
        }
        if (ocnt) ocnt[m] = (int32_t)(j - i);
        ++m; i = j;
    }
    free(kv);
    return m;
}

/* ------------------------------------------------------------------------------------------ */
/* Transform + fuse + voxel_down_sample of preprocessing/data.py:44-61 in one step:            */
/* cloud c is moved by T[c] (pcd.transform, :48; identity for the master), the clouds are      */
/* stacked in order (np.vstack, :55-58) and the stack is down-sampled (filter_outliers ->      */
/* voxel_down_sample, filtering.py:23).  In the reference the moved points are float64 arrays, */
/* so here every decision (min bound, voxel index) and every sum uses the fp64 value           */
/* p' = AC1(T[c], p) itself, never a float32 rounding of it -- in both storage modes.  Only    */
/* the per-voxel means are stored (real_t).  Same order rules as kpo_voxel_downsample.         */
/* ------------------------------------------------------------------------------------------ */
KPO_API int64_t kpo_fuse_voxel_downsample(int32_t count, const real_t *const *pts, const real_t *const *col, const int64_t *n,
                                          const double *T, double voxel, real_t *opts, real_t *ocol, int32_t *ocnt)
{
    if (!(voxel > 0.0)) return -1;
    int64_t total = 0;
    for (int c = 0; c < count; ++c) total += n[c];
    if (total == 0) return 0;
    double *q = (double *)malloc((size_t)total * 3 * sizeof(double));
    const real_t **cp = (const real_t **)malloc((size_t)total * sizeof(real_t *));
    int64_t w = 0;
    for (int c = 0; c < count; ++c)
        for (int64_t i = 0; i < n[c]; ++i, ++w) {
            xform3(T + 16 * c, (double)pts[c][3 * i], (double)pts[c][3 * i + 1], (double)pts[c][3 * i + 2], q + 3 * w);
            cp[w] = (col && col[c]) ? col[c] + 3 * i : NULL;
        }
    double mn[3] = { q[0], q[1], q[2] };
    for (int64_t i = 1; i < total; ++i)
        for (int a = 0; a < 3; ++a) if (q[3 * i + a] < mn[a]) mn[a] = q[3 * i + a];
    double org[3];
    for (int a = 0; a < 3; ++a) org[a] = mn[a] - voxel * 0.5;
    kv_t *kv = (kv_t *)malloc((size_t)total * sizeof(kv_t));
    for (int64_t i = 0; i < total; ++i) {
        uint64_t key = 0;
        for (int a = 0; a < 3; ++a) {
            double f = floor((q[3 * i + a] - org[a]) / voxel);
            if (!(f >= 0.0) || f >= 2097152.0) { free(kv); free(q); free(cp); return -2; }
            key = (key << 21) | (uint64_t)f;
        }
        kv[i].key = key; kv[i].idx = i;
    }
    qsort(kv, (size_t)total, sizeof(kv_t), cmp_kv);
    int64_t m = 0, i = 0;
    while (i < total) {
        int64_t j = i;
        double sp[3] = {0, 0, 0}, sc[3] = {0, 0, 0};
        while (j < total && kv[j].key == kv[i].key) {
            int64_t p = kv[j].idx;
            for (int a = 0; a < 3; ++a) {
                sp[a] += q[3 * p + a];
                if (cp[p]) sc[a] += (double)cp[p][a];
            }
            ++j;
        }
        double c = (double)(j - i);
        for (int a = 0; a < 3; ++a) {
            opts[3 * m + a] = (real_t)(sp[a] / c);
            if (ocol) ocol[3 * m + a] = (real_t)(sc[a] / c);
        }
        if (ocnt) ocnt[m] = (int32_t)(j - i);
        ++m; i = j;
    }
    free(kv); free(q); free(cp);
    return m;
}

/* ------------------------------------------------------------------------------------------ */
/* Uniform grid used to accelerate the exact neighbour searches (stand-in for Open3D's         */
/* KD-tree: any exact search returns the same neighbour distances).                             */
/* Contract AC3 (direct squared distance): d2 = fma(dz,dz, fma(dy,dy, dx*dx)), dx = xi - xj.   */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    double org[3], h;
    int dim[3];
    int64_t ncell;
    int64_t *start;     /* ncell+1 */
    int32_t *order;     /* point ids sorted by cell (stable) */
} grid_t;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void grid_build(grid_t *g, const real_t *pts, int64_t n, double target_per_cell)
{
    double mn[3] = { DBL_MAX, DBL_MAX, DBL_MAX }, mx[3] = { -DBL_MAX, -DBL_MAX, -DBL_MAX };
    for (int64_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            double v = pts[3 * i + a];
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
    double ext[3], vol = 1.0;
    for (int a = 0; a < 3; ++a) { ext[a] = mx[a] - mn[a]; if (ext[a] < 1e-9) ext[a] = 1e-9; vol *= ext[a]; }
    /* surfaces, not volumes: size cells from a 2-D density guess as well and take the larger */
    double h3 = cbrt(vol * target_per_cell / (double)(n > 0 ? n : 1));
    double area = ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2];
    double h2 = sqrt(area * target_per_cell / (double)(n > 0 ? n : 1)) * 0.5;
    double h = h3 > h2 ? h3 : h2;
    if (!(h > 0)) h = 1.0;
    for (;;) {
        int64_t tot = 1;
        for (int a = 0; a < 3; ++a) {
            double d = floor(ext[a] / h) + 1.0;
            if (d > 2000000.0) d = 2000000.0;
            g->dim[a] = (int)d; tot *= g->dim[a];
        }
        if (tot <= (int64_t)1 << 24) { g->ncell = tot; break; }
        h *= 1.26;
    }
    g->h = h;
    for (int a = 0; a < 3; ++a) g->org[a] = mn[a];
    g->start = (int64_t *)calloc((size_t)g->ncell + 1, sizeof(int64_t));
    g->order = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
    int64_t *cid = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) {
        int c[3];
        for (int a = 0; a < 3; ++a)
            c[a] = clampi((int)floor(((double)pts[3 * i + a] - g->org[a]) / h), 0, g->dim[a] - 1);
        cid[i] = ((int64_t)c[0] * g->dim[1] + c[1]) * g->dim[2] + c[2];
        g->start[cid[i] + 1]++;
    }
    for (int64_t c = 0; c < g->ncell; ++c) g->start[c + 1] += g->start[c];
    int64_t *fill = (int64_t *)malloc((size_t)g->ncell * sizeof(int64_t));
    memcpy(fill, g->start, (size_t)g->ncell * sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) g->order[fill[cid[i]]++] = (int32_t)i;
    free(fill); free(cid);
}
static void grid_free(grid_t *g) { free(g->start); free(g->order); }

static inline double dist2_direct(double ax, double ay, double az, double bx, double by, double bz)
{
    double dx = ax - bx, dy = ay - by, dz = az - bz;
    return fma(dz, dz, fma(dy, dy, dx * dx));
}

/* max-heap of (d2, idx) ordered lexicographically: keeps the k smallest pairs */
typedef struct { double d; int32_t i; } di_t;
static inline int di_less(di_t a, di_t b) { return a.d < b.d || (a.d == b.d && a.i < b.i); }
static void heap_push(di_t *h, int *sz, int k, di_t v)
{
    if (*sz < k) {
        int c = (*sz)++;
        h[c] = v;
        while (c > 0) { int p = (c - 1) / 2; if (di_less(h[p], h[c])) { di_t t = h[p]; h[p] = h[c]; h[c] = t; c = p; } else break; }
    } else if (di_less(v, h[0])) {
        h[0] = v;
        int c = 0;
        for (;;) {
            int l = 2 * c + 1, r = l + 1, b = c;
            if (l < k && di_less(h[b], h[l])) b = l;
            if (r < k && di_less(h[b], h[r])) b = r;
            if (b == c) break;
            di_t t = h[b]; h[b] = h[c]; h[c] = t; c = b;
        }
    }
}
static int cmp_di(const void *a, const void *b)
{
    di_t x = *(const di_t *)a, y = *(const di_t *)b;
    return di_less(x, y) ? -1 : (di_less(y, x) ? 1 : 0);
}

/* exact k nearest (by (d2, idx)) of query q among the gridded points; result sorted ascending.
 * If r2max >= 0 only points with d2 < r2max are returned ([O3D] SearchHybrid: knn then
 * lower_bound(radius^2), i.e. strict <).  Returns the number found. */
static int grid_knn(const grid_t *g, const real_t *pts, double qx, double qy, double qz,
                    int k, double r2max, di_t *heap)
{
    int c[3];
    double q[3] = { qx, qy, qz };
    for (int a = 0; a < 3; ++a) c[a] = clampi((int)floor((q[a] - g->org[a]) / g->h), 0, g->dim[a] - 1);
    int sz = 0;
    int maxr = g->dim[0] > g->dim[1] ? g->dim[0] : g->dim[1];
    if (g->dim[2] > maxr) maxr = g->dim[2];
    for (int r = 0; r <= maxr; ++r) {
        /* after rings 0..r-1 every point with distance < dcov is known, where dcov is the
         * distance from q to the boundary of the covered cube (clamped queries: use cell box) */
        if (r > 0) {
            double dcov = DBL_MAX;
            for (int a = 0; a < 3; ++a) {
                double lo = g->org[a] + (double)(c[a] - (r - 1)) * g->h;
                double hi = g->org[a] + (double)(c[a] + r) * g->h;
                double dl = q[a] - lo, dh = hi - q[a];
                if (c[a] - (r - 1) <= 0) dl = DBL_MAX;            /* grid edge: nothing beyond */
                if (c[a] + r >= g->dim[a]) dh = DBL_MAX;
                if (dl < dcov) dcov = dl;
                if (dh < dcov) dcov = dh;
            }
            if (dcov == DBL_MAX) break;                           /* whole grid covered */
            if (dcov < 0) dcov = 0;
            double cov2 = dcov * dcov * (1.0 - 1e-12);
            if (sz == k && heap[0].d < cov2) break;
            if (r2max >= 0 && cov2 >= r2max) break;
        }
        int x0 = c[0] - r, x1 = c[0] + r, y0 = c[1] - r, y1 = c[1] + r, z0 = c[2] - r, z1 = c[2] + r;
        for (int x = x0 < 0 ? 0 : x0; x <= (x1 >= g->dim[0] ? g->dim[0] - 1 : x1); ++x)
            for (int y = y0 < 0 ? 0 : y0; y <= (y1 >= g->dim[1] ? g->dim[1] - 1 : y1); ++y) {
                int shell_xy = (x == x0 || x == x1 || y == y0 || y == y1);
                for (int z = z0 < 0 ? 0 : z0; z <= (z1 >= g->dim[2] ? g->dim[2] - 1 : z1); ++z) {
                    if (!shell_xy && z != z0 && z != z1) continue;
                    int64_t cell = ((int64_t)x * g->dim[1] + y) * g->dim[2] + z;
                    for (int64_t s = g->start[cell]; s < g->start[cell + 1]; ++s) {
                        int32_t j = g->order[s];
                        double d = dist2_direct(qx, qy, qz, pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]);
                        if (r2max >= 0 && !(d < r2max)) continue;
                        di_t v = { d, j };
                        heap_push(heap, &sz, k, v);
                    }
                }
            }
    }
    qsort(heap, (size_t)sz, sizeof(di_t), cmp_di);
    return sz;
}

/* a8: [O3D] PointCloud::RemoveStatisticalOutliers, call sites filtering.py:24 (200, 3.0),
 * floor_removal.py:73 (50, 0.30), utils/processing.py:309.
 *   avg_i  = (sum_{ascending} sqrt(d2_j)) / k'   over the k' = min(k, N) nearest incl. itself
 *   mean   = (sum of avg_i with avg_i > 0) / N ;  std = sqrt(sum_{avg_i>0}(avg_i-mean)^2 / (N-1))
 *   keep   = avg_i > 0 && avg_i < mean + std_ratio*std ; indices ascending.
 * stats[0..2] = mean, std, threshold.  Returns kept count, -1 on invalid arguments.
 * brute != 0 uses the O(N^2) scan (validation of the grid search). */
/* second half of [O3D] RemoveStatisticalOutliers: mean / Bessel std over the per-point mean distances (sum over avg > 0,
 * divisor = all points: every point finds itself, so Open3D's valid_distances == n), keep avg > 0 && avg < mean + r std */
KPO_API int64_t kpo_sor_from_avg(const double *avg, int64_t n, double std_ratio, int32_t *keep_idx, double *stats)
{
    double sum = 0.0;
    for (int64_t i = 0; i < n; ++i) if (avg[i] > 0) sum += avg[i];
    double mean = sum / (double)n;
    double sq = 0.0;
    for (int64_t i = 0; i < n; ++i) if (avg[i] > 0) sq += (avg[i] - mean) * (avg[i] - mean);
    double sd = sqrt(sq / (double)(n - 1));
    double thr = mean + std_ratio * sd;
    if (stats) { stats[0] = mean; stats[1] = sd; stats[2] = thr; }
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; ++i)
        if (avg[i] > 0 && avg[i] < thr) { if (keep_idx) keep_idx[cnt] = (int32_t)i; ++cnt; }
    return cnt;
}

KPO_API int64_t kpo_sor(const real_t *pts, int64_t n, int k, double std_ratio, int brute,
                        int32_t *keep_idx, double *stats, double *avg_out)
{
    if (k < 1 || !(std_ratio > 0.0)) return -1;
    if (n == 0) return 0;
    int kk = (int64_t)k < n ? k : (int)n;
    double *avg = avg_out ? avg_out : (double *)malloc((size_t)n * sizeof(double));
    grid_t g;
    if (!brute) grid_build(&g, pts, n, (double)kk * 0.5 + 1.0);
#pragma omp parallel
    {
        di_t *heap = (di_t *)malloc((size_t)kk * sizeof(di_t));
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < n; ++i) {
            double qx = pts[3 * i], qy = pts[3 * i + 1], qz = pts[3 * i + 2];
            int sz = 0;
            if (brute) {
                for (int64_t j = 0; j < n; ++j) {
                    di_t v = { dist2_direct(qx, qy, qz, pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]), (int32_t)j };
                    heap_push(heap, &sz, kk, v);
                }
                qsort(heap, (size_t)sz, sizeof(di_t), cmp_di);
            } else {
                sz = grid_knn(&g, pts, qx, qy, qz, kk, -1.0, heap);
            }
            double s = 0.0;
            for (int t = 0; t < sz; ++t) s += sqrt(heap[t].d);
            avg[i] = sz > 0 ? s / (double)sz : -1.0;
        }
        free(heap);
    }
    if (!brute) grid_free(&g);
    int64_t cnt = kpo_sor_from_avg(avg, n, std_ratio, keep_idx, stats);
    if (!avg_out) free(avg);
    return cnt;
}

/* [O3D] KDTreeFlann::SearchHybrid(radius, max_nn) for every point of the cloud against itself
 * (estimate_normals, preprocessing/registration.py:9-13).  nbr: n*max_nn indices ascending by
 * (d2, idx); cnt: number found.  */
KPO_API int kpo_hybrid_knn_d2(const real_t *pts, int64_t n, double radius, int max_nn,
                              int32_t *nbr, int32_t *cnt, double *d2out);
KPO_API int kpo_hybrid_knn(const real_t *pts, int64_t n, double radius, int max_nn,
                           int32_t *nbr, int32_t *cnt)
{
    return kpo_hybrid_knn_d2(pts, n, radius, max_nn, nbr, cnt, NULL);
}
KPO_API int kpo_hybrid_knn_d2(const real_t *pts, int64_t n, double radius, int max_nn,
                              int32_t *nbr, int32_t *cnt, double *d2out)
{
    if (max_nn < 1 || !(radius > 0)) return -1;
    if (n == 0) return 0;
    int kk = (int64_t)max_nn < n ? max_nn : (int)n;
    grid_t g;
    grid_build(&g, pts, n, 8.0);
    double r2 = radius * radius;
#pragma omp parallel
    {
        di_t *heap = (di_t *)malloc((size_t)kk * sizeof(di_t));
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < n; ++i) {
            int sz = grid_knn(&g, pts, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], kk, r2, heap);
            cnt[i] = sz;
            for (int t = 0; t < sz; ++t) { nbr[i * max_nn + t] = heap[t].i; if (d2out) d2out[i * max_nn + t] = heap[t].d; }
            for (int t = sz; t < max_nn; ++t) { nbr[i * max_nn + t] = -1; if (d2out) d2out[i * max_nn + t] = 0.0; }
        }
        free(heap);
    }
    grid_free(&g);
    return 0;
}

/* [O3D] utility::ComputeCovariance over the hybrid neighbourhood (cumulant form):
 *   c[0..8] = mean of x,y,z,xx,xy,xz,yy,yz,zz (sequential, neighbour order), then
 *   cov = E[ab] - E[a]E[b].  Output 6 doubles per point: xx,xy,xz,yy,yz,zz; cnt<3 -> zeros. */
KPO_API void kpo_covariances(const real_t *pts, int64_t n, const int32_t *nbr, const int32_t *cnt,
                             int max_nn, double *cov6)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double c[9] = {0};
        int m = cnt[i];
        if (m < 3) { for (int a = 0; a < 6; ++a) cov6[6 * i + a] = 0.0; continue; }
        for (int t = 0; t < m; ++t) {
            int32_t j = nbr[i * max_nn + t];
            double x = pts[3 * j], y = pts[3 * j + 1], z = pts[3 * j + 2];
            c[0] += x; c[1] += y; c[2] += z;
            c[3] += x * x; c[4] += x * y; c[5] += x * z;
            c[6] += y * y; c[7] += y * z; c[8] += z * z;
        }
        for (int a = 0; a < 9; ++a) c[a] /= (double)m;
        cov6[6 * i + 0] = c[3] - c[0] * c[0];
        cov6[6 * i + 1] = c[4] - c[0] * c[1];
        cov6[6 * i + 2] = c[5] - c[0] * c[2];
        cov6[6 * i + 3] = c[6] - c[1] * c[1];
        cov6[6 * i + 4] = c[7] - c[1] * c[2];
        cov6[6 * i + 5] = c[8] - c[2] * c[2];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a21: [O3D] PointCloud::SegmentPlane (floor_removal.py:70: thr 30, ransac_n 30, 2000 iters). */
/* Sampling (ours, seeded; the reference's is unseeded): hypothesis h draws u32 values from     */
/* Philox(ctr=(block,h,0,0), key=(seed_lo,seed_hi)), four per block, index = (u*n)>>32,         */
/* duplicates rejected, until ransac_n distinct indices.                                        */
/* Plane: n==3 triangle normal, else the determinant least-squares fit (Appendix A).            */
/* Score: dist = |fma(a,x, fma(b,y, fma(c,z, d)))| ; inlier if dist < thr ; error = sum(dist)   */
/*        (in index order); fitness = count/N ; rmse = error/sqrt(count) [O3D quirk, recalled]. */
/* Best: higher fitness, tie -> lower rmse.  Early exit [O3D >= 0.16]: after an improvement     */
/*        break_iteration = min(log(1-p)/log(1-fitness^n), iters) (0 if fitness == 1; iters     */
/*        when fitness^n vanishes against 1 and the quotient is not a number >= 0);             */
/*        iterations with index > break_iteration are skipped.                                  */
/* Final: inliers of the best plane (ascending), plane re-fitted to them.                       */
/* ------------------------------------------------------------------------------------------ */
static void plane_from_points(const real_t *pts, const int32_t *ids, int64_t m, double pl[4])
{
    pl[0] = pl[1] = pl[2] = pl[3] = 0.0;
    if (m < 3) return;
    double cx = 0, cy = 0, cz = 0;
    for (int64_t t = 0; t < m; ++t) { int64_t j = ids ? ids[t] : t; cx += pts[3 * j]; cy += pts[3 * j + 1]; cz += pts[3 * j + 2]; }
    cx /= (double)m; cy /= (double)m; cz /= (double)m;
    double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
    for (int64_t t = 0; t < m; ++t) {
        int64_t j = ids ? ids[t] : t;
        double rx = pts[3 * j] - cx, ry = pts[3 * j + 1] - cy, rz = pts[3 * j + 2] - cz;
        xx += rx * rx; xy += rx * ry; xz += rx * rz; yy += ry * ry; yz += ry * rz; zz += rz * rz;
    }
    double det_x = yy * zz - yz * yz, det_y = xx * zz - xz * xz, det_z = xx * yy - xy * xy;
    double a, b, c;
    if (det_x > det_y && det_x > det_z) { a = det_x; b = xz * yz - xy * zz; c = xy * yz - xz * yy; }
    else if (det_y > det_z)             { a = xz * yz - xy * zz; b = det_y; c = xy * xz - yz * xx; }
    else                                { a = xy * yz - xz * yy; b = xy * xz - yz * xx; c = det_z; }
    double nn = sqrt(a * a + b * b + c * c);
    if (nn == 0.0) return;
    a /= nn; b /= nn; c /= nn;
    pl[0] = a; pl[1] = b; pl[2] = c; pl[3] = -(a * cx + b * cy + c * cz);
}
static void plane_from_triangle(const real_t *pts, const int32_t *ids, double pl[4])
{
    const real_t *p0 = pts + 3 * ids[0], *p1 = pts + 3 * ids[1], *p2 = pts + 3 * ids[2];
    double e0[3], e1[3];
    for (int a = 0; a < 3; ++a) { e0[a] = (double)p1[a] - p0[a]; e1[a] = (double)p2[a] - p0[a]; }
    double a = e0[1] * e1[2] - e0[2] * e1[1], b = e0[2] * e1[0] - e0[0] * e1[2], c = e0[0] * e1[1] - e0[1] * e1[0];
    double nn = sqrt(a * a + b * b + c * c);
    pl[0] = pl[1] = pl[2] = pl[3] = 0.0;
    if (nn == 0.0) return;
    a /= nn; b /= nn; c /= nn;
    pl[0] = a; pl[1] = b; pl[2] = c; pl[3] = -(a * p0[0] + b * p0[1] + c * p0[2]);
}
KPO_API void kpo_ransac_sample(int64_t n, int ransac_n, uint64_t seed, uint32_t h, int32_t *ids)
{
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    int got = 0;
    for (uint32_t blk = 0; got < ransac_n; ++blk) {
        uint32_t ctr[4] = { blk, h, 0u, 0u }, out[4];
        kpo_philox4x32(ctr, key, out);
        for (int w = 0; w < 4 && got < ransac_n; ++w) {
            int32_t id = (int32_t)(((uint64_t)out[w] * (uint64_t)n) >> 32);
            int dup = 0;
            for (int t = 0; t < got; ++t) if (ids[t] == id) { dup = 1; break; }
            if (!dup) ids[got++] = id;
        }
    }
}
KPO_API void kpo_plane_fit(const real_t *pts, const int32_t *ids, int64_t m, double pl[4])
{
    if (m == 3 && ids) plane_from_triangle(pts, ids, pl); else plane_from_points(pts, ids, m, pl);
}
static inline double plane_dist(const double pl[4], const real_t *p)
{
    return fabs(fma(pl[0], (double)p[0], fma(pl[1], (double)p[1], fma(pl[2], (double)p[2], pl[3]))));
}
/* per-hypothesis table hyp[h*6 + {a,b,c,d,count,error}] is optional (debug / GPU cross-check) */
KPO_API int kpo_segment_plane(const real_t *pts, int64_t n, double thr, int ransac_n, int iters,
                              double probability, uint64_t seed, double plane[4],
                              int32_t *inl_idx, int64_t *inl_count, double *hyp)
{
    if (ransac_n < 3 || n < ransac_n || !(probability > 0.0) || probability > 1.0) return -1;
    double best_fit = 0.0, best_rmse = 0.0, best_pl[4] = {0, 0, 0, 0};
    double break_it = DBL_MAX;
    int32_t *ids = (int32_t *)malloc((size_t)ransac_n * sizeof(int32_t));
    for (int it = 0; it < iters; ++it) {
        if ((double)it > break_it && !hyp) break;
        double pl[4];
        kpo_ransac_sample(n, ransac_n, seed, (uint32_t)it, ids);
        if (ransac_n == 3) plane_from_triangle(pts, ids, pl); else plane_from_points(pts, ids, ransac_n, pl);
        int zero = (pl[0] == 0 && pl[1] == 0 && pl[2] == 0 && pl[3] == 0);
        int64_t cnt = 0; double err = 0.0;
        if (!zero) {
            int64_t c = 0; double e = 0.0;
#pragma omp parallel for reduction(+ : c) schedule(static)
            for (int64_t i = 0; i < n; ++i) if (plane_dist(pl, pts + 3 * i) < thr) ++c;
            cnt = c;
            /* error summed sequentially in index order (the defined order) */
            if (cnt) for (int64_t i = 0; i < n; ++i) { double d = plane_dist(pl, pts + 3 * i); if (d < thr) e += d; }
            err = e;
        }
        if (hyp) { for (int a = 0; a < 4; ++a) hyp[6 * it + a] = pl[a]; hyp[6 * it + 4] = (double)cnt; hyp[6 * it + 5] = err; }
        if ((double)it > break_it || zero) continue;
        double fit = cnt ? (double)cnt / (double)n : 0.0;
        double rmse = cnt ? err / sqrt((double)cnt) : 0.0;
        if (fit > best_fit || (fit == best_fit && rmse < best_rmse)) {
            best_fit = fit; best_rmse = rmse; memcpy(best_pl, pl, sizeof(pl));
            if (fit < 1.0) {
                double b = log(1.0 - probability) / log(1.0 - pow(fit, (double)ransac_n));
                /* fitness^n below 2^-53: the denominator is log(1.0) = 0 and b = -inf (p = 1: NaN).  [O3D] assigns that double
                 * to a size_t -- undefined, and on x86-64 the conversion yields 2^63: the loop never breaks.  Found by the
                 * second lineage (oracle/lineage2.py); until round 4 this restatement stopped at the first hypothesis there. */
                if (!(b >= 0.0)) b = (double)iters;
                break_it = b < (double)iters ? b : (double)iters;
                break_it = floor(break_it);              /* size_t truncation in [O3D] */
            } else break_it = 0.0;
        }
    }
    free(ids);
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) if (plane_dist(best_pl, pts + 3 * i) < thr) inl_idx[k++] = (int32_t)i;
    *inl_count = k;
    plane_from_points(pts, inl_idx, k, plane);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* a14/a16: correspondence search of [O3D] registration_icp                                    */
/* (manual_pointcloud_registration.py:96-98, preprocessing/registration.py:78-84).             */
/* Source point i is first transformed by T (AC1, fp64, not rounded).  Nearest target under     */
/* contract AC2:                                                                                */
/*   K_i  = fma(s_x,s_x, fma(s_y,s_y, s_z*s_z)) + 1                                             */
/*   D_ij = fma(1, |t_j|^2, fma(s_z,-2t_z, fma(s_y,-2t_y, fma(s_x,-2t_x, K_i))))                */
/* (= d^2 + 1 > 0: the K=4 augmented inner product accumulated onto the row seed K_i as a       */
/* k-ordered fma chain, which is what the f64 MFMA computes), |t|^2 = fma(tx,tx,fma(ty,ty,tz*tz)); */
/* argmin, ties -> lowest j.  The reported d2 is the direct form AC3 of the chosen pair.        */
/* ------------------------------------------------------------------------------------------ */
static inline double nn_metric(const double s[3], const real_t *t)
{
    double tx = t[0], ty = t[1], tz = t[2];
    double t2 = fma(tx, tx, fma(ty, ty, tz * tz));
    double m = fma(s[0], s[0], fma(s[1], s[1], s[2] * s[2])) + 1.0;
    m = fma(s[0], -2.0 * tx, m);
    m = fma(s[1], -2.0 * ty, m);
    m = fma(s[2], -2.0 * tz, m);
    m = fma(1.0, t2, m);
    return m;
}
KPO_API int kpo_nn_brute(const real_t *src, int64_t n, const double *T, const real_t *tgt, int64_t m,
                         int32_t *idx, double *d2, double *metric)
{
    if (m <= 0) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double s[3];
        xform3(T, src[3 * i], src[3 * i + 1], src[3 * i + 2], s);
        double best = DBL_MAX; int64_t bj = 0;
        for (int64_t j = 0; j < m; ++j) {
            double v = nn_metric(s, tgt + 3 * j);
            if (v < best) { best = v; bj = j; }
        }
        idx[i] = (int32_t)bj;
        if (metric) metric[i] = best;
        d2[i] = dist2_direct(s[0], s[1], s[2], tgt[3 * bj], tgt[3 * bj + 1], tgt[3 * bj + 2]);
    }
    return 0;
}
/* Same answer through the grid: ring search on the direct distance with a safety margin, the
 * AC2 metric decides among everything within the margin of the best. */
KPO_API int kpo_nn_grid(const real_t *src, int64_t n, const double *T, const real_t *tgt, int64_t m,
                        int32_t *idx, double *d2, double *metric)
{
    if (m <= 0) return -1;
    grid_t g;
    grid_build(&g, tgt, m, 4.0);
    int maxr = g.dim[0] > g.dim[1] ? g.dim[0] : g.dim[1];
    if (g.dim[2] > maxr) maxr = g.dim[2];
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i) {
        double s[3];
        xform3(T, src[3 * i], src[3 * i + 1], src[3 * i + 2], s);
        int c[3];
        double outside2 = 0.0;          /* squared distance from s to the grid box (clamped query) */
        for (int a = 0; a < 3; ++a) {
            double r = floor((s[a] - g.org[a]) / g.h);
            if (r < 0) { double o = g.org[a] - s[a]; outside2 += o * o; c[a] = 0; }
            else if (r > g.dim[a] - 1) { double o = s[a] - (g.org[a] + g.dim[a] * g.h); if (o > 0) outside2 += o * o; c[a] = g.dim[a] - 1; }
            else c[a] = (int)r;
        }
        double bestm = DBL_MAX, bestd = DBL_MAX; int64_t bj = -1;
        for (int r = 0; r <= maxr; ++r) {
            if (r > 0 && bj >= 0) {
                double dcov = DBL_MAX;
                for (int a = 0; a < 3; ++a) {
                    double lo = g.org[a] + (double)(c[a] - (r - 1)) * g.h;
                    double hi = g.org[a] + (double)(c[a] + r) * g.h;
                    double dl = s[a] - lo, dh = hi - s[a];
                    if (c[a] - (r - 1) <= 0) dl = DBL_MAX;
                    if (c[a] + r >= g.dim[a]) dh = DBL_MAX;
                    if (dl < dcov) dcov = dl;
                    if (dh < dcov) dcov = dh;
                }
                if (dcov == DBL_MAX) break;
                if (dcov > 0 && bestd * (1.0 + 1e-9) + 1e-9 < dcov * dcov) break;
            }
            int x0 = c[0] - r, x1 = c[0] + r, y0 = c[1] - r, y1 = c[1] + r, z0 = c[2] - r, z1 = c[2] + r;
            for (int x = x0 < 0 ? 0 : x0; x <= (x1 >= g.dim[0] ? g.dim[0] - 1 : x1); ++x)
                for (int y = y0 < 0 ? 0 : y0; y <= (y1 >= g.dim[1] ? g.dim[1] - 1 : y1); ++y) {
                    int shell_xy = (x == x0 || x == x1 || y == y0 || y == y1);
                    for (int z = z0 < 0 ? 0 : z0; z <= (z1 >= g.dim[2] ? g.dim[2] - 1 : z1); ++z) {
                        if (!shell_xy && z != z0 && z != z1) continue;
                        int64_t cell = ((int64_t)x * g.dim[1] + y) * g.dim[2] + z;
                        for (int64_t q = g.start[cell]; q < g.start[cell + 1]; ++q) {
                            int32_t j = g.order[q];
                            double v = nn_metric(s, tgt + 3 * j);
                            if (v < bestm || (v == bestm && j < bj)) {
                                bestm = v; bj = j;
                                bestd = dist2_direct(s[0], s[1], s[2], tgt[3 * j], tgt[3 * j + 1], tgt[3 * j + 2]);
                            }
                        }
                    }
                }
        }
        (void)outside2;
        idx[i] = (int32_t)bj;
        if (metric) metric[i] = bestm;
        d2[i] = bestd;
    }
    grid_free(&g);
    return 0;
}

/* ICP accumulations for one iteration, given the correspondences (idx, d2) of the transformed
 * source: valid pair iff d2 < max_dist^2 ([O3D] SearchHybrid strict <).
 * out[0]=count, out[1]=sum d2, out[2..4]=sum s, out[5..7]=sum t, out[8..16]=sum t s^T (row-major,
 * rows = target component)  -> Umeyama needs Sigma = E[t s^T] - mu_t mu_s^T.
 * If tn (target normals) != NULL also the point-to-plane normal equations:
 * out[17..37] = upper triangle of J^T J (6x6 row-major upper), out[38..43] = J^T r,
 * with r = (s - t).n, J = [s x n, n]  ([O3D] TransformationEstimationPointToPlane). */
KPO_API void kpo_icp_accumulate(const real_t *src, int64_t n, const double *T, const real_t *tgt,
                                const real_t *tn, const int32_t *idx, const double *d2,
                                double max_dist, double *out)
{
    for (int a = 0; a < 44; ++a) out[a] = 0.0;
    double md2 = max_dist * max_dist;
    for (int64_t i = 0; i < n; ++i) {
        if (!(d2[i] < md2)) continue;
        double s[3];
        xform3(T, src[3 * i], src[3 * i + 1], src[3 * i + 2], s);
        const real_t *t = tgt + 3 * (int64_t)idx[i];
        out[0] += 1.0; out[1] += d2[i];
        for (int a = 0; a < 3; ++a) { out[2 + a] += s[a]; out[5 + a] += (double)t[a]; }
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) out[8 + 3 * a + b] += (double)t[a] * s[b];
        if (tn) {
            const real_t *nf = tn + 3 * (int64_t)idx[i];
            double nx = nf[0], ny = nf[1], nz = nf[2];
            double r = (s[0] - t[0]) * nx + (s[1] - t[1]) * ny + (s[2] - t[2]) * nz;
            double J[6] = { s[1] * nz - s[2] * ny, s[2] * nx - s[0] * nz, s[0] * ny - s[1] * nx, nx, ny, nz };
            int q = 17;
            for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) out[q++] += J[a] * J[b];
            for (int a = 0; a < 6; ++a) out[38 + a] += J[a] * r;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a11: [O3D] compute_fpfh_feature (preprocessing/registration.py:15-20), 33 = 3 x 11 bins.    */
/* Pair features (Darboux frame) exactly as Open3D's ComputePairFeatures; the acos comparison   */
/* acos|a1| > acos|a2| is written as |a1| < |a2| (acos is decreasing).  Angle binning of        */
/* f0 = atan2(w.n2, n1.n2) uses atan2 from libm; all other bins are affine in dot products.     */
/* Neighbours: hybrid search, ascending (d2, idx), the point itself (slot 0) skipped.           */
/* spfh, fpfh: (n, 33) row-major doubles.                                                       */
/* ------------------------------------------------------------------------------------------ */
static void pair_features(const double *p1, const double *n1, const double *p2, const double *n2, double f[4])
{
    double dp[3] = { p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2] };
    f[0] = f[1] = f[2] = 0.0;
    f[3] = sqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2]);
    if (f[3] == 0.0) { f[3] = 0.0; return; }
    double a[3] = { n1[0], n1[1], n1[2] }, b[3] = { n2[0], n2[1], n2[2] };
    double angle1 = (a[0] * dp[0] + a[1] * dp[1] + a[2] * dp[2]) / f[3];
    double angle2 = (b[0] * dp[0] + b[1] * dp[1] + b[2] * dp[2]) / f[3];
    if (fabs(angle1) < fabs(angle2)) {
        for (int k = 0; k < 3; ++k) { a[k] = n2[k]; b[k] = n1[k]; dp[k] = -dp[k]; }
        f[2] = -angle2;
    } else f[2] = angle1;
    double v[3] = { dp[1] * a[2] - dp[2] * a[1], dp[2] * a[0] - dp[0] * a[2], dp[0] * a[1] - dp[1] * a[0] };
    double vn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (vn == 0.0) { f[0] = f[1] = f[2] = f[3] = 0.0; return; }
    v[0] /= vn; v[1] /= vn; v[2] /= vn;
    double w[3] = { a[1] * v[2] - a[2] * v[1], a[2] * v[0] - a[0] * v[2], a[0] * v[1] - a[1] * v[0] };
    f[1] = v[0] * b[0] + v[1] * b[1] + v[2] * b[2];
    f[0] = atan2(w[0] * b[0] + w[1] * b[1] + w[2] * b[2], a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
}
static inline int bin11(double x)
{
    int h = (int)floor(x);
    return h < 0 ? 0 : (h >= 11 ? 10 : h);
}
KPO_API void kpo_fpfh(const real_t *pts, const real_t *nrm, int64_t n, const int32_t *nbr, const int32_t *cnt,
                      const double *d2, int max_nn, double *spfh, double *fpfh)
{
    memset(spfh, 0, (size_t)n * 33 * sizeof(double));
    memset(fpfh, 0, (size_t)n * 33 * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        int m = cnt[i];
        if (m <= 1) continue;
        double p1[3] = { pts[3 * i], pts[3 * i + 1], pts[3 * i + 2] }, n1[3] = { nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2] };
        double inc = 100.0 / (double)(m - 1);
        for (int t = 1; t < m; ++t) {
            int64_t j = nbr[i * max_nn + t];
            double p2[3] = { pts[3 * j], pts[3 * j + 1], pts[3 * j + 2] }, n2[3] = { nrm[3 * j], nrm[3 * j + 1], nrm[3 * j + 2] };
            double f[4];
            pair_features(p1, n1, p2, n2, f);
            spfh[33 * i + bin11(11.0 * (f[0] + M_PI) / (2.0 * M_PI))] += inc;
            spfh[33 * i + 11 + bin11(11.0 * (f[1] + 1.0) * 0.5)] += inc;
            spfh[33 * i + 22 + bin11(11.0 * (f[2] + 1.0) * 0.5)] += inc;
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        int m = cnt[i];
        if (m <= 1) continue;
        double sum[3] = { 0, 0, 0 };
        double *o = fpfh + 33 * i;
        for (int t = 1; t < m; ++t) {
            double dist = d2[i * max_nn + t];
            if (dist == 0.0) continue;
            const double *sp = spfh + 33 * (int64_t)nbr[i * max_nn + t];
            for (int j = 0; j < 33; ++j) { double val = sp[j] / dist; sum[j / 11] += val; o[j] += val; }
        }
        for (int j = 0; j < 3; ++j) if (sum[j] != 0.0) sum[j] = 100.0 / sum[j];
        for (int j = 0; j < 33; ++j) { o[j] *= sum[j / 11]; o[j] += spfh[33 * i + j]; }
    }
}

/* [O3D] feature matching of registration_ransac_based_on_feature_matching: 1-NN in the 33-D feature  */
/* space, squared distance sum_k (a_k - b_k)^2 accumulated k = 0..32 with fma, ties -> lowest index.   */
KPO_API void kpo_feature_nn(const double *fa, int64_t na, const double *fb, int64_t nb, int32_t *idx)
{
    /* metric of the matching stage (the product's MFMA chain, K = 36 augmented form): D = |a|^2, then */
    /* D = fma(a_k, -2 b_k, D) for k = 0..32, then D = fma(1, |b|^2, D); |x|^2 = fma chain x_k x_k from 0 */
    double *nb2 = (double *)malloc((size_t)(nb > 0 ? nb : 1) * sizeof(double));
    for (int64_t j = 0; j < nb; ++j) {
        double n2 = 0.0;
        for (int k = 0; k < 33; ++k) n2 = fma(fb[33 * j + k], fb[33 * j + k], n2);
        nb2[j] = n2;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < na; ++i) {
        double na2 = 0.0;
        for (int k = 0; k < 33; ++k) na2 = fma(fa[33 * i + k], fa[33 * i + k], na2);
        double best = INFINITY; int64_t bj = 0;
        for (int64_t j = 0; j < nb; ++j) {
            double d = na2;
            for (int k = 0; k < 33; ++k) d = fma(fa[33 * i + k], -2.0 * fb[33 * j + k], d);
            d = fma(1.0, nb2[j], d);
            if (d < best) { best = d; bj = j; }
        }
        idx[i] = (int32_t)bj;
    }
    free(nb2);
}

/* Umeyama without scale on 3+ pairs by Horn's quaternion method (independent of the product's Jacobi/ */
/* cross-product construction): R maximises trace(R^T S), S = sum (t-mu_t)(s-mu_s)^T.                   */
static void jacobi4(double A[4][4], double V[4][4], double w[4])
{
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 50; ++sweep) {
        double off = 0;
        for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) {
            if (A[p][q] == 0.0) continue;
            double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 4; ++k) { double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
            for (int k = 0; k < 4; ++k) { double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
            for (int k = 0; k < 4; ++k) { double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
        }
    }
    for (int i = 0; i < 4; ++i) w[i] = A[i][i];
}
static void kabsch_pairs(const double *s, const double *t, int m, double T[16])
{
    double ms[3] = {0, 0, 0}, mt[3] = {0, 0, 0};
    for (int i = 0; i < m; ++i) for (int a = 0; a < 3; ++a) { ms[a] += s[3 * i + a]; mt[a] += t[3 * i + a]; }
    for (int a = 0; a < 3; ++a) { ms[a] /= m; mt[a] /= m; }
    double S[3][3] = {{0}};
    for (int i = 0; i < m; ++i)
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) S[a][b] += (s[3 * i + a] - ms[a]) * (t[3 * i + b] - mt[b]);   /* S_ab = s_a t_b */
    double N[4][4] = {
        { S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0] },
        { S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2] },
        { S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1] },
        { S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2] } };
    double V[4][4], w[4];
    jacobi4(N, V, w);
    int b = 0;
    for (int i = 1; i < 4; ++i) if (w[i] > w[b]) b = i;
    double q0 = V[0][b], q1 = V[1][b], q2 = V[2][b], q3 = V[3][b];
    double R[3][3] = {
        { q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3, 2 * (q1 * q2 - q0 * q3), 2 * (q1 * q3 + q0 * q2) },
        { 2 * (q2 * q1 + q0 * q3), q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3, 2 * (q2 * q3 - q0 * q1) },
        { 2 * (q3 * q1 - q0 * q2), 2 * (q3 * q2 + q0 * q1), q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3 } };
    for (int k = 0; k < 16; ++k) T[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int a = 0; a < 3; ++a) {
        for (int c = 0; c < 3; ++c) T[4 * a + c] = R[a][c];
        T[4 * a + 3] = mt[a] - (R[a][0] * ms[0] + R[a][1] * ms[1] + R[a][2] * ms[2]);
    }
}
KPO_API void kpo_kabsch_pairs(const double *s, const double *t, int m, double *T) { kabsch_pairs(s, t, m, T); }

/* a13: [O3D] RegistrationRANSACBasedOnCorrespondence (preprocessing/registration.py:50-57):            */
/*   sample ransac_n correspondences (with replacement; Philox(ctr=(block,itr,1,0), key=seed), index =  */
/*   (u*|corres|)>>32), Umeyama, checkers (edge length 0.95, distance), validation = full nearest-      */
/*   neighbour correspondence search within max_dist (fitness, rmse), better-than test, est_k update    */
/*   from the inlier ratio of the correspondence set.  Iterations run in order (Open3D's run in an       */
/*   OpenMP loop in unspecified order: the reference result is not reproducible).                       */
/* stats[0] = iterations run, [1] = validations, [2] = fitness, [3] = rmse.  Returns 0, T = identity if  */
/* nothing passed.                                                                                       */
KPO_API int kpo_ransac_corres(const real_t *src, int64_t n, const real_t *tgt, int64_t m, const int32_t *corres, int64_t nc,
                              double max_dist, int ransac_n, double edge_sim, int max_iter, double confidence,
                              uint64_t seed, double *Tbest, double *stats)
{
    for (int k = 0; k < 16; ++k) Tbest[k] = (k % 5 == 0) ? 1.0 : 0.0;
    stats[0] = stats[1] = stats[2] = stats[3] = 0.0;
    if (ransac_n < 3 || nc < ransac_n || !(max_dist > 0)) return -1;
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    double best_fit = 0.0, best_rmse = 0.0;
    int est_k = max_iter, validations = 0, itr;
    int32_t *idx = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    double *d2 = (double *)malloc((size_t)n * sizeof(double));
    double *sp = (double *)malloc((size_t)ransac_n * 3 * sizeof(double)), *tp = (double *)malloc((size_t)ransac_n * 3 * sizeof(double));
    double md2 = max_dist * max_dist;
    for (itr = 0; itr < max_iter; ++itr) {
        if (itr >= est_k) break;
        int got = 0;
        int32_t pick[64];
        for (uint32_t blk = 0; got < ransac_n; ++blk) {
            uint32_t ctr[4] = { blk, (uint32_t)itr, 1u, 0u }, out[4];
            kpo_philox4x32(ctr, key, out);
            for (int w = 0; w < 4 && got < ransac_n; ++w) pick[got++] = (int32_t)(((uint64_t)out[w] * (uint64_t)nc) >> 32);
        }
        for (int q = 0; q < ransac_n; ++q)
            for (int a = 0; a < 3; ++a) { sp[3 * q + a] = src[3 * (int64_t)corres[2 * pick[q]] + a]; tp[3 * q + a] = tgt[3 * (int64_t)corres[2 * pick[q] + 1] + a]; }
        int ok = 1;
        for (int i = 0; i < ransac_n && ok; ++i)            /* CorrespondenceCheckerBasedOnEdgeLength */
            for (int j = i + 1; j < ransac_n; ++j) {
                double ds = 0, dt = 0;
                for (int a = 0; a < 3; ++a) { double e = sp[3 * i + a] - sp[3 * j + a]; ds += e * e; e = tp[3 * i + a] - tp[3 * j + a]; dt += e * e; }
                ds = sqrt(ds); dt = sqrt(dt);
                if (ds < dt * edge_sim || dt < ds * edge_sim) { ok = 0; break; }
            }
        if (!ok) continue;
        double T[16];
        kabsch_pairs(sp, tp, ransac_n, T);
        for (int q = 0; q < ransac_n && ok; ++q) {           /* CorrespondenceCheckerBasedOnDistance */
            double o[3], e2 = 0;
            xform3(T, sp[3 * q], sp[3 * q + 1], sp[3 * q + 2], o);
            for (int a = 0; a < 3; ++a) e2 += (o[a] - tp[3 * q + a]) * (o[a] - tp[3 * q + a]);
            if (sqrt(e2) > max_dist) ok = 0;
        }
        if (!ok) continue;
        ++validations;
        kpo_nn_grid(src, n, T, tgt, m, idx, d2, NULL);
        int64_t cnt = 0; double err = 0;
        for (int64_t i = 0; i < n; ++i) if (d2[i] < md2) { ++cnt; err += d2[i]; }
        double fit = cnt ? (double)cnt / (double)n : 0.0, rmse = cnt ? sqrt(err / (double)cnt) : 0.0;
        if (fit > best_fit || (fit == best_fit && rmse < best_rmse)) {
            best_fit = fit; best_rmse = rmse; memcpy(Tbest, T, sizeof(T));
            int64_t inl = 0;
            for (int64_t c = 0; c < nc; ++c) {
                double o[3], e2 = 0;
                const real_t *s = src + 3 * (int64_t)corres[2 * c], *t = tgt + 3 * (int64_t)corres[2 * c + 1];
                xform3(T, s[0], s[1], s[2], o);
                for (int a = 0; a < 3; ++a) e2 += (o[a] - t[a]) * (o[a] - t[a]);
                if (sqrt(e2) < max_dist) ++inl;
            }
            double ratio = (double)inl / (double)nc;
            double ek = log(1.0 - confidence) / log(1.0 - pow(ratio, (double)ransac_n));
            if (ek < (double)est_k) est_k = (int)ceil(ek);
        }
    }
    free(idx); free(d2); free(sp); free(tp);
    stats[0] = itr; stats[1] = validations; stats[2] = best_fit; stats[3] = best_rmse;
    return 0;
}

KPO_API int kpo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* the sampler / normaliser that follows the path.                            */
/*                                                                                            */
/* select_points_randomly (utils/processing.py:259-275) draws np.random.choice(N, k,            */
/* replace=False) from NumPy's unseeded global generator: there is no reference stream to       */
/* reproduce.  Contract used on both sides: point i gets the 64-bit key                         */
/* Philox4x32-10(ctr = (i_lo, i_hi, 'SAMP', 0), key = seed) words (1:0); the sample is the k     */
/* points with the smallest (key, i), in that order: a uniformly random ordered k-subset.       */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t key; int64_t i; } samp_t;
static int cmp_samp(const void *a, const void *b)
{
    const samp_t *x = (const samp_t *)a, *y = (const samp_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->i < y->i ? -1 : (x->i > y->i);
}
KPO_API int kpo_sample_indices(int64_t n, int64_t k, uint64_t seed, int32_t *idx)
{
    if (k < 0 || k > n) return -1;            /* np.random.choice: "Cannot take a larger sample than population" */
    samp_t *s = (samp_t *)malloc(sizeof(samp_t) * (size_t)(n > 0 ? n : 1));
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    for (int64_t i = 0; i < n; ++i) {
        uint32_t ctr[4] = { (uint32_t)i, (uint32_t)((uint64_t)i >> 32), 0x53414D50u, 0u }, out[4];
        kpo_philox4x32(ctr, key, out);
        s[i].key = ((uint64_t)out[1] << 32) | out[0];
        s[i].i = i;
    }
    qsort(s, (size_t)n, sizeof(samp_t), cmp_samp);
    for (int64_t j = 0; j < k; ++j) idx[j] = (int32_t)s[j].i;
    free(s);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* PointCloud.get_oriented_bounding_box() (utils/normalization.py:38-42,74-77,105-106;          */
/* utils/processing.py:341-344) is Open3D's OrientedBoundingBox::CreateFromPoints [O3D,          */
/* recalled]: convex hull (Qhull), then mean / covariance of the HULL VERTICES, eigenvectors by   */
/* descending eigenvalue as the columns of R (third = first x second), the axis-aligned box of   */
/* R^T (v - mean) giving centre and extent.  Qhull is not in the reference tree; its published   */
/* result for points in general position -- the set of extreme points -- is restated here by gift */
/* wrapping, with this arithmetic contract (AC5, fp64, explicit fma, shared with the device):    */
/*   wrap about the directed edge a->b of a facet (a, b, r) with outward normal                  */
/*   n = e x g (e = b - a, g = r - a), t = e x n; for candidate c, d = c - a:                    */
/*   u = t.d, w = max(-(n.d), 0), key2 = (u / |e|^2) u + w w (= |n|^2 x squared distance from the  */
/*   edge line); c is skipped as lying on that line when key2 <= 2^-80 (n.n)(d.d); key1 = u / w    */
/*   (w == 0: +-inf by the sign of u), key3 = e.d; the winner maximises (key1, key2, key3),        */
/*   lowest index on full ties; it is the next extreme point also when several hull points are    */
/*   coplanar (key2 / key3 walk the facet polygon).  New facet (b, a, c).                         */
/* PARITY: unpinned against Open3D (absent here); the vertex set is pinned against Qhull itself   */
/* (scipy.spatial.ConvexHull) in tests/test_oracle_cpu.py.                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { double k1, k2, k3; int32_t i; } wrapkey_t;
static inline int wrap_better(const wrapkey_t *x, const wrapkey_t *y)      /* x beats y */
{
    if (y->i < 0) return x->i >= 0;
    if (x->i < 0) return 0;
    if (x->k1 != y->k1) return x->k1 > y->k1;
    if (x->k2 != y->k2) return x->k2 > y->k2;
    if (x->k3 != y->k3) return x->k3 > y->k3;
    return x->i < y->i;
}
static inline double dot3f(const double a[3], const double b[3]) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }
static inline void cross3f(const double a[3], const double b[3], double o[3])
{
    o[0] = fma(a[1], b[2], -(a[2] * b[1]));
    o[1] = fma(a[2], b[0], -(a[0] * b[2]));
    o[2] = fma(a[0], b[1], -(a[1] * b[0]));
}
/* frame of one wrap: a, e, n (outward normal of the known facet), t, 1/|e|^2 */
typedef struct { double a[3], e[3], n[3], t[3], inv_e2, n2s; } wrapframe_t;
static void wrap_frame(const double a[3], const double b[3], const double r[3], wrapframe_t *f)
{
    double g[3];
    for (int k = 0; k < 3; ++k) { f->a[k] = a[k]; f->e[k] = b[k] - a[k]; g[k] = r[k] - a[k]; }
    cross3f(f->e, g, f->n);
    cross3f(f->e, f->n, f->t);
    f->inv_e2 = 1.0 / dot3f(f->e, f->e);
    f->n2s = dot3f(f->n, f->n) * 0x1p-80;
}
static inline wrapkey_t wrap_key(const wrapframe_t *f, const double c[3], int32_t i)
{
    double d[3] = { c[0] - f->a[0], c[1] - f->a[1], c[2] - f->a[2] };
    double u = dot3f(f->t, d), w = -dot3f(f->n, d);
    wrapkey_t k; k.i = i;
    if (w < 0.0) w = 0.0;
    k.k2 = fma(u * f->inv_e2, u, w * w);
    if (k.k2 <= f->n2s * dot3f(d, d)) { k.i = -1; k.k1 = k.k2 = k.k3 = 0.0; return k; }     /* on the edge line */
    k.k1 = w == 0.0 ? (u > 0.0 ? INFINITY : -INFINITY) : u / w;
    k.k3 = dot3f(f->e, d);
    return k;
}
static int32_t wrap_scan(const wrapframe_t *f, const double *pts, int64_t n, int32_t ia, int32_t ib)
{
    wrapkey_t best; best.i = -1; best.k1 = best.k2 = best.k3 = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        if (i == ia || i == ib) continue;
        wrapkey_t k = wrap_key(f, pts + 3 * i, (int32_t)i);
        if (wrap_better(&k, &best)) best = k;
    }
    return best.i;
}
/* open-addressing set of directed edges */
typedef struct { uint64_t *slot; uint64_t mask; } edgeset_t;
static inline uint64_t edge_code(int32_t a, int32_t b) { return (((uint64_t)(uint32_t)a) << 32 | (uint32_t)b) + 1; }
static inline uint64_t edge_hash(uint64_t c) { c *= 0x9E3779B97F4A7C15ull; return c ^ (c >> 29); }
static int edge_has(const edgeset_t *s, int32_t a, int32_t b)
{
    uint64_t c = edge_code(a, b);
    for (uint64_t h = edge_hash(c) & s->mask;; h = (h + 1) & s->mask) {
        if (s->slot[h] == c) return 1;
        if (s->slot[h] == 0) return 0;
    }
}
static void edge_put(edgeset_t *s, int32_t a, int32_t b)
{
    uint64_t c = edge_code(a, b);
    for (uint64_t h = edge_hash(c) & s->mask;; h = (h + 1) & s->mask) {
        if (s->slot[h] == c) return;
        if (s->slot[h] == 0) { s->slot[h] = c; return; }
    }
}
/* is_vertex: u8 [n].  Returns the number of hull vertices, or -1 when the cloud has no proper first facet
 * (fewer than 3 distinct points, or all of them on one line), -2 if the wrap did not close within 2n+64 facets.
 * pts: f64 (n,3) -- float32 clouds are widened by the caller (exact). */
KPO_API int64_t kpo_hull_vertices(const double *pts, int64_t n, uint8_t *is_vertex)
{
    memset(is_vertex, 0, (size_t)n);
    if (n < 3) return -1;
    int32_t p0 = 0;
    for (int64_t i = 1; i < n; ++i) {
        const double *a = pts + 3 * i, *b = pts + 3 * (int64_t)p0;
        if (a[0] < b[0] || (a[0] == b[0] && (a[1] < b[1] || (a[1] == b[1] && a[2] < b[2])))) p0 = (int32_t)i;
    }
    double A[3] = { pts[3 * (int64_t)p0], pts[3 * (int64_t)p0 + 1], pts[3 * (int64_t)p0 + 2] };
    /* virtual facet: the half plane {x = x0, y <= y0} bounded by the line through p0 along z */
    wrapframe_t f;
    for (int k = 0; k < 3; ++k) f.a[k] = A[k];
    f.e[0] = 0; f.e[1] = 0; f.e[2] = -1; f.n[0] = -1; f.n[1] = 0; f.n[2] = 0; f.t[0] = 0; f.t[1] = 1; f.t[2] = 0; f.inv_e2 = 1.0; f.n2s = 0x1p-80;
    int32_t c1 = wrap_scan(&f, pts, n, p0, p0);
    if (c1 < 0) return -1;
    double B[3] = { pts[3 * (int64_t)c1], pts[3 * (int64_t)c1 + 1], pts[3 * (int64_t)c1 + 2] };
    double V[3] = { A[0], A[1], A[2] - 1.0 };            /* a + e of the virtual edge: the virtual facet is (v, p0, c1) */
    wrap_frame(A, B, V, &f);
    int32_t c2 = wrap_scan(&f, pts, n, p0, c1);
    if (c2 < 0) return -1;
    int64_t max_facets = 2 * n + 64, facets = 0;
    uint64_t cap = 16; while (cap < (uint64_t)(8 * n)) cap <<= 1;
    edgeset_t es; es.slot = (uint64_t *)calloc(cap, 8); es.mask = cap - 1;
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)(2 * max_facets + 8));
    int64_t sp = 0;
#define KPO_PUSH(x, y, z) do { stack[3 * sp] = (x); stack[3 * sp + 1] = (y); stack[3 * sp + 2] = (z); ++sp; } while (0)
    /* first real facet (c1, p0, c2): all three of its edges are open */
    edge_put(&es, c1, p0); edge_put(&es, p0, c2); edge_put(&es, c2, c1);
    is_vertex[p0] = is_vertex[c1] = is_vertex[c2] = 1;
    KPO_PUSH(c1, p0, c2); KPO_PUSH(p0, c2, c1); KPO_PUSH(c2, c1, p0);
    facets = 1;
    int64_t rc = 0;
    while (sp > 0) {
        --sp;
        int32_t a = stack[3 * sp], b = stack[3 * sp + 1], r = stack[3 * sp + 2];
        if (edge_has(&es, b, a)) continue;
        if (++facets > max_facets) { rc = -2; break; }
        double pa[3], pb[3], pr[3];
        for (int k = 0; k < 3; ++k) { pa[k] = pts[3 * (int64_t)a + k]; pb[k] = pts[3 * (int64_t)b + k]; pr[k] = pts[3 * (int64_t)r + k]; }
        wrap_frame(pa, pb, pr, &f);
        int32_t c = wrap_scan(&f, pts, n, a, b);
        if (c < 0) { rc = -2; break; }
        edge_put(&es, b, a); edge_put(&es, a, c); edge_put(&es, c, b);
        is_vertex[c] = 1;
        KPO_PUSH(a, c, b); KPO_PUSH(c, b, a);
    }
#undef KPO_PUSH
    free(stack); free(es.slot);
    if (rc) return rc;
    int64_t v = 0;
    for (int64_t i = 0; i < n; ++i) v += is_vertex[i];
    return v;
}
