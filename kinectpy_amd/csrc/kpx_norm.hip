// kpx_norm.hip -- SURVEY 8f rank 3: the sampler / normaliser right after the path.
//   kpx_sample_points    select_points_randomly (utils/processing.py:259-275), seeded
//   kpx_obb_batch        PointCloud.get_oriented_bounding_box() of a batch of clouds (utils/normalization.py:38-42,
//                        74-77, 105-106; utils/processing.py:341-344)
//   kpx_normalize_batch  the three normalisations of utils/normalization.py:16-126 and obb_normalization
//                        (utils/processing.py:329-354) applied to rows of f64 triples
//
// Oriented bounding box = Open3D's CreateFromPoints: convex hull, then PCA of the hull VERTICES.  The hull vertices
// are found by gift wrapping, ONE BLOCK PER CLOUD: every wrap is a block-wide arg-max of a three-part key over all
// points; thread 0 keeps the open-edge stack and the set of directed edges in LDS (spilling to the workspace for hulls
// with more than ~2000 facets).  A wrap costs one pass over the cloud (L2 resident) plus two block barriers, a hull of
// v vertices takes 2v-4 wraps: latency-bound by construction, which is fine for what it serves -- batches of
// 4096-point training clouds run one per CU side by side.
//
// Arithmetic contract AC5 (fp64, every fma explicit): wrap about the directed edge a->b of a facet
// (a, b, r) with outward normal n = e x g (e = b - a, g = r - a), t = e x n; for a candidate c, d = c - a:
//   u = t.d,  w = max(-(n.d), 0),  key2 = (u / |e|^2) u + w w  (= |n|^2 x squared distance from the edge line);
//   c is skipped as lying on that line when key2 <= 2^-80 (n.n)(d.d);  key1 = u / w (w == 0: +-inf by the sign of u),
//   key3 = e.d;  the winner maximises (key1, key2, key3), lowest index on full ties -- it is the next extreme point
//   also when several hull points are coplanar (key2 / key3 walk the facet polygon).  New facet (b, a, c).
// The start is the lexicographically smallest point and the virtual half plane {x = x0, y <= y0} through it.
#include <hipcub/hipcub.hpp>

#include "kpx_internal.h"
#include "kpx_linalg.h"

namespace kpx {

// ---- sampler ----------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
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

__global__ __launch_bounds__(256) void sample_key_kernel(int64_t n, uint32_t seed_lo, uint32_t seed_hi, uint64_t *__restrict__ keys,
                                                         int32_t *__restrict__ vals)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t o[4];
        philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), 0x53414D50u, 0u, seed_lo, seed_hi, o);
        keys[i] = ((uint64_t)o[1] << 32) | o[0];
        vals[i] = (int32_t)i;
    }
}
__global__ __launch_bounds__(256) void sample_gather_kernel(const float *__restrict__ pts, const int32_t *__restrict__ order, int64_t k,
                                                            float *__restrict__ out_pts, int32_t *__restrict__ out_idx)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < k; j += (int64_t)gridDim.x * blockDim.x) {
        const int32_t i = order[j];
        if (out_idx) out_idx[j] = i;
        if (out_pts) {
            out_pts[3 * j] = pts[3 * (int64_t)i];
            out_pts[3 * j + 1] = pts[3 * (int64_t)i + 1];
            out_pts[3 * j + 2] = pts[3 * (int64_t)i + 2];
        }
    }
}
struct SampleScratch {
    uint64_t *keys_in, *keys_out;
    int32_t *vals_in, *vals_out;
    void *tmp;
    size_t tmp_bytes;
};
static void sample_carve(Arena &a, int64_t n, SampleScratch *s)
{
    const size_t nn = (size_t)(n > 0 ? n : 1);
    s->keys_in = a.get<uint64_t>(nn);
    s->keys_out = a.get<uint64_t>(nn);
    s->vals_in = a.get<int32_t>(nn);
    s->vals_out = a.get<int32_t>(nn);
    s->tmp_bytes = memo_bytes(9, (int64_t)nn, [&] { size_t b = 0; (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, s->keys_in, s->keys_out, s->vals_in, s->vals_out, (int)nn, 0, 64, (hipStream_t) nullptr); return b; });
    s->tmp = a.get<char>(s->tmp_bytes);
}

// ---- gift wrapping (AC5) ------------------------------------------------------------------------------
struct WrapKey {
    double k1, k2, k3;
    int32_t i;
};
__device__ __forceinline__ bool wrap_better(const WrapKey &x, const WrapKey &y)      // x beats y
{
    if (y.i < 0) return x.i >= 0;
    if (x.i < 0) return false;
    if (x.k1 != y.k1) return x.k1 > y.k1;
    if (x.k2 != y.k2) return x.k2 > y.k2;
    if (x.k3 != y.k3) return x.k3 > y.k3;
    return x.i < y.i;
}
__device__ __forceinline__ double dot3f(const double a[3], const double b[3]) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }
__device__ __forceinline__ void cross3f(const double a[3], const double b[3], double o[3])
{
    o[0] = fma(a[1], b[2], -(a[2] * b[1]));
    o[1] = fma(a[2], b[0], -(a[0] * b[2]));
    o[2] = fma(a[0], b[1], -(a[1] * b[0]));
}
struct WrapFrame {
    double a[3], e[3], n[3], t[3], inv_e2, n2s;
};
__device__ __forceinline__ void wrap_frame(const double a[3], const double b[3], const double r[3], WrapFrame *f)
{
    double g[3];
    for (int k = 0; k < 3; ++k) { f->a[k] = a[k]; f->e[k] = b[k] - a[k]; g[k] = r[k] - a[k]; }
    cross3f(f->e, g, f->n);
    cross3f(f->e, f->n, f->t);
    f->inv_e2 = 1.0 / dot3f(f->e, f->e);
    f->n2s = dot3f(f->n, f->n) * 0x1p-80;
}
__device__ __forceinline__ WrapKey shfl_xor_key(const WrapKey &k, int o)
{
    WrapKey r;
    r.k1 = __shfl_xor(k.k1, o, 64);
    r.k2 = __shfl_xor(k.k2, o, 64);
    r.k3 = __shfl_xor(k.k3, o, 64);
    r.i = __shfl_xor(k.i, o, 64);
    return r;
}
// best key of the block; valid in thread 0.  sh: one entry per wave.  wrap_better is a strict total order, so the
// result does not depend on the shape of the reduction.
// PRE = false: the caller guarantees that a barrier separates this call from the previous reader of sh.
template <bool PRE = true> __device__ __forceinline__ WrapKey block_best(WrapKey k, WrapKey *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        WrapKey t = shfl_xor_key(k, o);
        if (wrap_better(t, k)) k = t;
    }
    if (PRE) __syncthreads();
    if (lane_id() == 0) sh[wave_id()] = k;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; ++w)
            if (wrap_better(sh[w], k)) k = sh[w];
    }
    return k;
}

// block-wide maximum, valid in thread 0 (same shape as block_sum)
__device__ __forceinline__ double block_max(double v, double *sh)
{
    v = wave_max(v);
    __syncthreads();
    if (lane_id() == 0) sh[wave_id()] = v;
    __syncthreads();
    double r = v;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; ++w) r = fmax(r, sh[w]);
    }
    return r;
}

constexpr int kHullLdsSlots = 8192;      // directed-edge set in LDS: 64 KB
constexpr int kHullMigrate = 6144;       // entries at which the set moves to the workspace table
constexpr int kHullLdsStack = 4096;      // open-edge stack entries in LDS (3 ints each): 48 KB
constexpr int kHullMaxWaves = 16;

__device__ __forceinline__ uint64_t edge_code(int32_t a, int32_t b) { return ((((uint64_t)(uint32_t)a) << 32) | (uint32_t)b) + 1; }
__device__ __forceinline__ uint64_t edge_hash(uint64_t c)
{
    c *= 0x9E3779B97F4A7C15ull;
    return c ^ (c >> 29);
}
struct EdgeSet {
    uint64_t *slot;      // LDS or workspace (flat)
    uint64_t mask;
    int32_t count;
};
__device__ __forceinline__ bool edge_has(const EdgeSet &s, int32_t a, int32_t b)
{
    const uint64_t c = edge_code(a, b);
    uint64_t h = edge_hash(c) & s.mask;
    for (uint64_t probe = 0; probe <= s.mask; ++probe, h = (h + 1) & s.mask) {
        const uint64_t v = s.slot[h];
        if (v == c) return true;
        if (v == 0) return false;
    }
    return false;
}
__device__ __forceinline__ void edge_put(EdgeSet &s, int32_t a, int32_t b)
{
    const uint64_t c = edge_code(a, b);
    uint64_t h = edge_hash(c) & s.mask;
    for (uint64_t probe = 0; probe <= s.mask; ++probe, h = (h + 1) & s.mask) {
        const uint64_t v = s.slot[h];
        if (v == c) return;
        if (v == 0) { s.slot[h] = c; ++s.count; return; }
    }
}

struct HullShared {
    WrapFrame f;
    int32_t ia, ib, go, migrate, status;
    double red[kHullMaxWaves][9];
    double R[9], mean[3];
    WrapKey best[kHullMaxWaves];
};

template <class T> __device__ __forceinline__ void load_pt(const T *__restrict__ pts, int64_t i, double p[3])
{
    p[0] = (double)pts[3 * i];
    p[1] = (double)pts[3 * i + 1];
    p[2] = (double)pts[3 * i + 2];
}

// One candidate against the running best.  The quick reject skips the division for a candidate whose key1 = u / w is
// certainly below the best so far (margin 2^-40 relative, far above the rounding of the product): it could not have
// won, so the result is the arg-max of the full keys all the same.
__device__ __forceinline__ void wrap_consider(const WrapFrame &f, double cx, double cy, double cz, int32_t i, WrapKey &best)
{
    const double d[3] = { cx - f.a[0], cy - f.a[1], cz - f.a[2] };
    const double u = dot3f(f.t, d);
    double w = -dot3f(f.n, d);
    if (w < 0.0) w = 0.0;
    if (best.i >= 0 && w > 0.0) {
        const double bw = best.k1 * w;
        if (u < bw - fabs(bw) * 0x1p-40) return;
    }
    WrapKey k;
    k.k2 = fma(u * f.inv_e2, u, w * w);
    if (k.k2 <= f.n2s * dot3f(d, d)) return;          // on the edge line (or a duplicate of an end point)
    k.i = i;
    k.k1 = w == 0.0 ? (u > 0.0 ? INFINITY : -INFINITY) : u / w;
    k.k3 = dot3f(f.e, d);
    if (wrap_better(k, best)) best = k;
}

constexpr int kHullCache = 4;      // points per thread held in registers (the first kHullCache * blockDim points)
struct PtCache {
    double x[kHullCache], y[kHullCache], z[kHullCache];
};
template <class T> __device__ __forceinline__ WrapKey wrap_scan(const WrapFrame &f, const PtCache &pc, const T *__restrict__ pts, int64_t n,
                                                                int32_t ia, int32_t ib)
{
    WrapKey best;
    best.i = -1; best.k1 = best.k2 = best.k3 = 0.0;
#pragma unroll
    for (int j = 0; j < kHullCache; ++j) {
        const int64_t i = threadIdx.x + (int64_t)j * blockDim.x;
        if (i < n && i != ia && i != ib) wrap_consider(f, pc.x[j], pc.y[j], pc.z[j], (int32_t)i, best);
    }
    int64_t i = threadIdx.x + (int64_t)kHullCache * blockDim.x;
    for (; i + 3 * (int64_t)blockDim.x < n; i += 4 * (int64_t)blockDim.x) {      // four independent loads in flight
        double c[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t iq = i + (int64_t)q * blockDim.x;
            c[q][0] = (double)pts[3 * iq]; c[q][1] = (double)pts[3 * iq + 1]; c[q][2] = (double)pts[3 * iq + 2];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t iq = i + (int64_t)q * blockDim.x;
            if (iq != ia && iq != ib) wrap_consider(f, c[q][0], c[q][1], c[q][2], (int32_t)iq, best);
        }
    }
    for (; i < n; i += blockDim.x)
        if (i != ia && i != ib) wrap_consider(f, (double)pts[3 * i], (double)pts[3 * i + 1], (double)pts[3 * i + 2], (int32_t)i, best);
    return best;
}

// obb: R (row-major 9) | centre 3 | extent 3 | hull vertices (or -1: no proper first facet, -2: the wrap did not close,
// -3: flat hull).  glob_slots: cap entries (zeroed) per cloud; glob_stack: 3 * stack_cap ints per cloud.
template <class T>
__global__ __launch_bounds__(1024) void hull_obb_kernel(const T *__restrict__ pts_all, int64_t n, uint64_t *__restrict__ glob_slots_all,
                                                        uint64_t glob_cap, int32_t *__restrict__ glob_stack_all, int64_t stack_cap,
                                                        uint8_t *__restrict__ flags_all, double *__restrict__ obb_all)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *lds_slots = (uint64_t *)smem;
    int32_t *lds_stack = (int32_t *)(smem + (size_t)kHullLdsSlots * 8);
    __shared__ HullShared S;

    const int64_t cloud = blockIdx.x;
    const T *pts = pts_all + 3 * n * cloud;
    uint64_t *glob_slots = glob_slots_all + glob_cap * cloud;
    int32_t *glob_stack = glob_stack_all + 3 * stack_cap * cloud;
    uint8_t *flags = flags_all + n * cloud;
    double *obb = obb_all + 16 * cloud;
    const int tid = threadIdx.x;

    for (int i = tid; i < kHullLdsSlots; i += blockDim.x) lds_slots[i] = 0;
    if (tid == 0) { S.go = 0; S.migrate = 0; S.status = 0; }
    PtCache pc;
#pragma unroll
    for (int j = 0; j < kHullCache; ++j) {
        const int64_t i = tid + (int64_t)j * blockDim.x;
        pc.x[j] = i < n ? (double)pts[3 * i] : 0.0;
        pc.y[j] = i < n ? (double)pts[3 * i + 1] : 0.0;
        pc.z[j] = i < n ? (double)pts[3 * i + 2] : 0.0;
    }

    // p0 = lexicographically smallest point (an extreme point); ties to the lowest index
    WrapKey k;
    k.i = -1; k.k1 = k.k2 = k.k3 = 0.0;
    for (int64_t i = tid; i < n; i += blockDim.x) {
        WrapKey c;
        c.k1 = -(double)pts[3 * i]; c.k2 = -(double)pts[3 * i + 1]; c.k3 = -(double)pts[3 * i + 2]; c.i = (int32_t)i;
        if (wrap_better(c, k)) k = c;
    }
    k = block_best(k, S.best);
    // thread-0 state
    EdgeSet es;
    es.slot = lds_slots; es.mask = kHullLdsSlots - 1; es.count = 0;
    int64_t sp = 0, facets = 0;
    const int64_t max_facets = 2 * n + 64;
    int32_t p0 = -1, c1 = -1;
    double A[3] = { 0, 0, 0 };
    if (tid == 0) {
        p0 = k.i;
        if (n < 3 || p0 < 0) S.status = -1;
        else {
            load_pt(pts, p0, A);
            // virtual facet: the half plane {x = x0, y <= y0} bounded by the line through p0 along z
            for (int q = 0; q < 3; ++q) S.f.a[q] = A[q];
            S.f.e[0] = 0; S.f.e[1] = 0; S.f.e[2] = -1;
            S.f.n[0] = -1; S.f.n[1] = 0; S.f.n[2] = 0;
            S.f.t[0] = 0; S.f.t[1] = 1; S.f.t[2] = 0;
            S.f.inv_e2 = 1.0; S.f.n2s = 0x1p-80;
            S.ia = p0; S.ib = p0;
        }
    }
    __syncthreads();
    if (S.status == 0) {                                        // first edge (p0, c1)
        const WrapFrame f = S.f;
        k = wrap_scan(f, pc, pts, n, S.ia, S.ib);
        k = block_best(k, S.best);
        if (tid == 0) {
            c1 = k.i;
            if (c1 < 0) S.status = -1;
            else {
                double B[3], V[3] = { A[0], A[1], A[2] - 1.0 };
                load_pt(pts, c1, B);
                wrap_frame(A, B, V, &S.f);
                S.ia = p0; S.ib = c1;
            }
        }
        __syncthreads();
    }
    if (S.status == 0) {                                        // first facet (c1, p0, c2)
        const WrapFrame f = S.f;
        k = wrap_scan(f, pc, pts, n, S.ia, S.ib);
        k = block_best(k, S.best);
        if (tid == 0) {
            const int32_t c2 = k.i;
            if (c2 < 0) S.status = -1;
            else {
                edge_put(es, c1, p0); edge_put(es, p0, c2); edge_put(es, c2, c1);
                flags[p0] = 1; flags[c1] = 1; flags[c2] = 1;
                const int32_t first[9] = { c1, p0, c2, p0, c2, c1, c2, c1, p0 };
                for (int q = 0; q < 9; ++q) lds_stack[q] = first[q];
                sp = 3;
                facets = 1;
            }
        }
    }
    // main loop: thread 0 pops the next open edge whose other side has no facet yet, everybody wraps about it
    for (;;) {
        if (tid == 0) {
            S.go = 0;
            while (S.status == 0 && sp > 0) {
                --sp;
                const int32_t *e = sp < kHullLdsStack ? lds_stack + 3 * sp : glob_stack + 3 * (sp - kHullLdsStack);
                const int32_t a = e[0], b = e[1], r = e[2];
                if (edge_has(es, b, a)) continue;
                if (++facets > max_facets) { S.status = -2; break; }
                double pa[3], pb[3], pr[3];
                load_pt(pts, a, pa); load_pt(pts, b, pb); load_pt(pts, r, pr);
                wrap_frame(pa, pb, pr, &S.f);
                S.ia = a; S.ib = b;
                S.go = 1;
                break;
            }
            S.migrate = (es.slot == lds_slots && es.count >= kHullMigrate) ? 1 : 0;
        }
        __syncthreads();
        if (S.migrate) {                                       // move the edge set to the (zeroed) workspace table
            for (int i = tid; i < kHullLdsSlots; i += blockDim.x) {
                const uint64_t c = lds_slots[i];
                if (c == 0) continue;
                uint64_t h = edge_hash(c) & (glob_cap - 1);
                for (uint64_t probe = 0; probe < glob_cap; ++probe, h = (h + 1) & (glob_cap - 1))
                    if (atomicCAS((unsigned long long *)&glob_slots[h], 0ull, (unsigned long long)c) == 0ull) break;
            }
            __threadfence_block();
            __syncthreads();
            if (tid == 0) { es.slot = glob_slots; es.mask = glob_cap - 1; }
        }
        if (!S.go) break;
        const int32_t ia = S.ia, ib = S.ib;
        const WrapFrame f = S.f;
        k = wrap_scan(f, pc, pts, n, ia, ib);
        k = block_best<false>(k, S.best);      // the barrier at the top of the loop already separates the uses of S.best
        if (tid == 0) {
            const int32_t c = k.i;
            if (c < 0) S.status = -2;
            else {
                edge_put(es, ib, ia); edge_put(es, ia, c); edge_put(es, c, ib);
                flags[c] = 1;
                const int32_t push[6] = { ia, c, ib, c, ib, ia };
                for (int q = 0; q < 2; ++q) {
                    int32_t *e = sp < kHullLdsStack ? lds_stack + 3 * sp : glob_stack + 3 * (sp - kHullLdsStack);
                    e[0] = push[3 * q]; e[1] = push[3 * q + 1]; e[2] = push[3 * q + 2];
                    ++sp;
                }
            }
        }
        // (thread 0 goes straight on to the next pop: the barrier at the top of the loop publishes it)
    }
    __threadfence_block();
    __syncthreads();
    if (S.status != 0) {
        if (tid == 0) {
            for (int q = 0; q < 15; ++q) obb[q] = __builtin_nan("");
            obb[15] = (double)S.status;
        }
        return;
    }
    // ---- PCA of the hull vertices (cumulant form, [O3D] ComputeMeanAndCovariance) and the box in that frame
    double c9[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, cnt = 0;
    for (int64_t i = tid; i < n; i += blockDim.x) {
        if (!__builtin_nontemporal_load(&flags[i])) continue;
        const double x = (double)pts[3 * i], y = (double)pts[3 * i + 1], z = (double)pts[3 * i + 2];
        c9[0] += x; c9[1] += y; c9[2] += z;
        c9[3] += x * x; c9[4] += x * y; c9[5] += x * z; c9[6] += y * y; c9[7] += y * z; c9[8] += z * z;
        cnt += 1.0;
    }
    double *red = &S.red[0][0];
    cnt = block_sum(cnt, red);
    double nv = 0;
    if (tid == 0) nv = cnt;
    double tot[9];
    for (int q = 0; q < 9; ++q) tot[q] = block_sum(c9[q], red);
    if (tid == 0) {
        for (int q = 0; q < 9; ++q) tot[q] /= nv;
        const double cov[6] = { tot[3] - tot[0] * tot[0], tot[4] - tot[0] * tot[1], tot[5] - tot[0] * tot[2],
                                tot[6] - tot[1] * tot[1], tot[7] - tot[1] * tot[2], tot[8] - tot[2] * tot[2] };
        double w[3], V[9], R[9];
        sym3_eigen(cov, w, V);                                 // ascending; columns of V
        for (int col = 0; col < 2; ++col) {                   // descending eigenvalue; largest component positive
            double v[3] = { V[0 + (2 - col)], V[3 + (2 - col)], V[6 + (2 - col)] };
            int j = 0;
            if (fabs(v[1]) > fabs(v[j])) j = 1;
            if (fabs(v[2]) > fabs(v[j])) j = 2;
            const double s = (v[j] < 0 ? -1.0 : 1.0) / sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            for (int r = 0; r < 3; ++r) R[3 * r + col] = v[r] * s;
        }
        R[2] = R[3] * R[7] - R[6] * R[4];                       // col2 = col0 x col1
        R[5] = R[6] * R[1] - R[0] * R[7];
        R[8] = R[0] * R[4] - R[3] * R[1];
        for (int q = 0; q < 9; ++q) S.R[q] = R[q];
        for (int q = 0; q < 3; ++q) S.mean[q] = tot[q];
    }
    __syncthreads();
    double lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int64_t i = tid; i < n; i += blockDim.x) {
        if (!__builtin_nontemporal_load(&flags[i])) continue;
        const double d[3] = { (double)pts[3 * i] - S.mean[0], (double)pts[3 * i + 1] - S.mean[1], (double)pts[3 * i + 2] - S.mean[2] };
        for (int j = 0; j < 3; ++j) {
            const double q = S.R[j] * d[0] + S.R[3 + j] * d[1] + S.R[6 + j] * d[2];
            lo[j] = fmin(lo[j], q);
            hi[j] = fmax(hi[j], q);
        }
    }
    for (int j = 0; j < 3; ++j) { lo[j] = -block_max(-lo[j], red); hi[j] = block_max(hi[j], red); }
    if (tid == 0) {
        double mid[3], ext[3];
        for (int j = 0; j < 3; ++j) { mid[j] = 0.5 * (lo[j] + hi[j]); ext[j] = hi[j] - lo[j]; }
        for (int q = 0; q < 9; ++q) obb[q] = S.R[q];
        for (int r = 0; r < 3; ++r) obb[9 + r] = S.R[3 * r] * mid[0] + S.R[3 * r + 1] * mid[1] + S.R[3 * r + 2] * mid[2] + S.mean[r];
        for (int j = 0; j < 3; ++j) obb[12 + j] = ext[j];
        obb[15] = (nv < 4.0 || !(ext[2] > 0.0)) ? -3.0 : nv;
    }
}

// ---- normalisations -----------------------------------------------------------------------------------
struct Mat3 {
    double m[9];
};
// x: rows of f64 triples, `rows` per cloud; obb as written by hull_obb_kernel
__global__ __launch_bounds__(256) void normalize_kernel(const double *__restrict__ x, int64_t rows, int32_t count, const double *__restrict__ obb_all,
                                                        int mode, Mat3 M, double *__restrict__ out)
{
    const int64_t total = rows * count;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const double *obb = obb_all + 16 * (i / rows);
        const double *c = obb + 9;
        double p[3] = { x[3 * i], x[3 * i + 1], x[3 * i + 2] }, o[3];
        if (mode == KPX_NORM_OBB) {                              // (p @ M + c) / max extent
            const double L = fmax(obb[12], fmax(obb[13], obb[14]));
            for (int j = 0; j < 3; ++j) o[j] = (((p[0] * M.m[j] + p[1] * M.m[3 + j]) + p[2] * M.m[6 + j]) + c[j]) / L;
        } else {
            for (int j = 0; j < 3; ++j) p[j] -= c[j];
            if (mode == KPX_NORM_TRANSLATE) {
                for (int j = 0; j < 3; ++j) o[j] = p[j];
            } else {                                             // (p - c) @ R [@ M]
                double q[3];
                for (int j = 0; j < 3; ++j) q[j] = (p[0] * obb[j] + p[1] * obb[3 + j]) + p[2] * obb[6 + j];
                if (mode == KPX_NORM_OBB_ROT_TRANS)
                    for (int j = 0; j < 3; ++j) o[j] = (q[0] * M.m[j] + q[1] * M.m[3 + j]) + q[2] * M.m[6 + j];
                else
                    for (int j = 0; j < 3; ++j) o[j] = q[j];
            }
        }
        out[3 * i] = o[0]; out[3 * i + 1] = o[1]; out[3 * i + 2] = o[2];
    }
}

struct HullPlan {
    uint64_t glob_cap;
    int64_t stack_cap;
};
static HullPlan hull_plan(int64_t n)
{
    HullPlan p;
    p.glob_cap = 1024;
    while (p.glob_cap < (uint64_t)(8 * (n > 0 ? n : 1))) p.glob_cap <<= 1;
    p.stack_cap = 2 * (2 * n + 64) + 8;
    return p;
}
struct HullScratch {
    uint64_t *slots;
    int32_t *stack;
    uint8_t *flags;
};
static void hull_carve(Arena &a, int32_t count, int64_t n, HullScratch *s)
{
    const HullPlan p = hull_plan(n);
    const size_t c = (size_t)(count > 0 ? count : 1);
    s->slots = a.get<uint64_t>(c * p.glob_cap);
    s->stack = a.get<int32_t>(c * 3 * (size_t)p.stack_cap);
    s->flags = a.get<uint8_t>(c * (size_t)(n > 0 ? n : 1));
}

}  // namespace kpx

using namespace kpx;

KPX_EXPORT size_t kpx_sample_workspace_bytes(int64_t n)
{
    Arena a(nullptr, 0);
    SampleScratch s;
    sample_carve(a, n, &s);
    return a.off;
}
KPX_EXPORT int kpx_sample_points(const float *pts, int64_t n, int64_t k, uint64_t seed, float *out_pts, int32_t *out_idx, void *ws,
                                 size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(n >= 0 && k >= 0, "kpx_sample_points: negative size");
    KPX_REQUIRE(k <= n, "kpx_sample_points: cannot take a larger sample than population without replacement (%lld > %lld)", (long long)k,
                (long long)n);
    KPX_REQUIRE(n < ((int64_t)1 << 31), "kpx_sample_points: more than 2^31-1 points");
    if (k == 0) return KPX_OK;
    KPX_REQUIRE(ws && (out_idx || (out_pts && pts)), "kpx_sample_points: null pointer");
    hipStream_t st = (hipStream_t)stream;
    Arena a(ws, ws_bytes);
    SampleScratch s;
    sample_carve(a, n, &s);
    KPX_ARENA_CHECK(a);
    hipLaunchKernelGGL(sample_key_kernel, dim3((unsigned)(cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256))), dim3(256), 0, st, n, (uint32_t)seed,
                       (uint32_t)(seed >> 32), s.keys_in, s.vals_in);
    size_t bytes = s.tmp_bytes;
    KPX_HIP(hipcub::DeviceRadixSort::SortPairs(s.tmp, bytes, s.keys_in, s.keys_out, s.vals_in, s.vals_out, (int)n, 0, 64, st));
    hipLaunchKernelGGL(sample_gather_kernel, dim3((unsigned)(cdiv(k, 256) > 4096 ? 4096 : cdiv(k, 256))), dim3(256), 0, st, pts, s.vals_out, k,
                       out_pts, out_idx);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

KPX_EXPORT size_t kpx_obb_workspace_bytes(int32_t count, int64_t n)
{
    Arena a(nullptr, 0);
    HullScratch s;
    hull_carve(a, count, n, &s);
    return a.off;
}
KPX_EXPORT int kpx_obb_batch(const void *pts, int32_t pts_f64, int32_t count, int64_t n, double *d_obb, uint8_t *d_is_vertex, void *ws,
                             size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(count >= 0 && n >= 0, "kpx_obb_batch: negative size");
    KPX_REQUIRE(n < ((int64_t)1 << 31), "kpx_obb_batch: more than 2^31-1 points per cloud");
    if (count == 0) return KPX_OK;
    KPX_REQUIRE(d_obb && ws && (pts || n == 0), "kpx_obb_batch: null pointer");
    hipStream_t st = (hipStream_t)stream;
    Arena a(ws, ws_bytes);
    HullScratch s;
    hull_carve(a, count, n, &s);
    KPX_ARENA_CHECK(a);
    const HullPlan p = hull_plan(n);
    uint8_t *flags = d_is_vertex ? d_is_vertex : s.flags;
    KPX_HIP(hipMemsetAsync(s.slots, 0, (size_t)count * p.glob_cap * 8, st));
    KPX_HIP(hipMemsetAsync(flags, 0, (size_t)count * (size_t)(n > 0 ? n : 1), st));
    const size_t lds = (size_t)kHullLdsSlots * 8 + (size_t)kHullLdsStack * 12;
    const int threads = 1024;      // the LDS tables allow one block per CU anyway
    if (pts_f64) {
        static bool once = false;
        if (!once) { KPX_HIP(hipFuncSetAttribute((const void *)hull_obb_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
        hipLaunchKernelGGL(hull_obb_kernel<double>, dim3((unsigned)count), dim3(threads), lds, st, (const double *)pts, n, s.slots, p.glob_cap,
                           s.stack, p.stack_cap, flags, d_obb);
    } else {
        static bool once = false;
        if (!once) { KPX_HIP(hipFuncSetAttribute((const void *)hull_obb_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
        hipLaunchKernelGGL(hull_obb_kernel<float>, dim3((unsigned)count), dim3(threads), lds, st, (const float *)pts, n, s.slots, p.glob_cap,
                           s.stack, p.stack_cap, flags, d_obb);
    }
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

KPX_EXPORT int kpx_normalize_batch(const double *x, int32_t count, int64_t rows, const double *d_obb, int32_t mode, const double *h_M,
                                   double *out, void *stream)
{
    KPX_REQUIRE(count >= 0 && rows >= 0, "kpx_normalize_batch: negative size");
    KPX_REQUIRE(mode >= KPX_NORM_OBB && mode <= KPX_NORM_OBB_ROT, "kpx_normalize_batch: unknown mode %d", (int)mode);
    if (count == 0 || rows == 0) return KPX_OK;
    KPX_REQUIRE(x && d_obb && out, "kpx_normalize_batch: null pointer");
    KPX_REQUIRE(h_M || (mode != KPX_NORM_OBB && mode != KPX_NORM_OBB_ROT_TRANS), "kpx_normalize_batch: this mode needs the constant matrix");
    Mat3 M;
    for (int q = 0; q < 9; ++q) M.m[q] = h_M ? h_M[q] : (q % 4 == 0 ? 1.0 : 0.0);
    const int64_t total = rows * count;
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256))), dim3(256), 0, (hipStream_t)stream, x,
                       rows, count, d_obb, (int)mode, M, out);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}
