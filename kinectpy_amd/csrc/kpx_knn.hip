// kpx_knn.hip -- exact neighbour searches on a uniform grid:
//   a8  PointCloud.remove_statistical_outlier (filtering.py:24, floor_removal.py:73, utils/processing.py:309)
//       estimate_normals(KDTreeSearchParamHybrid) (preprocessing/registration.py:9-13)
// One thread per query walks Chebyshev rings of cells around its own cell and keeps the k smallest
// squared distances in a per-thread max-heap held in LDS (layout [slot][thread]: bank-conflict free
// when the threads of a wave touch the same slot).  The search is exact: a ring loop stops only once
// the k-th best distance is below the distance to the boundary of the covered cube.
// Squared distance (contract AC3): d2 = fma(dz,dz, fma(dy,dy, dx*dx)), differences in fp64.
#include <hipcub/hipcub.hpp>

#include "kpx_gridknn.h"
#include "kpx_morton.h"
#include "kpx_radix.h"
#include "kpx_linalg.h"

namespace kpx {

// ---- grid construction ------------------------------------------------------------------------------
// Cell size of the grid: from the bounding box (volume and area heuristics), then -- refine -- corrected once by the occupancy an
// average POINT saw in the first binning (sum c^2 / n: isolated outliers each own a cell and would drag a per-cell mean down);
// occupancy ~ h^2.3 for surfaces with some thickness.  Pure function of (bbox, n, target, cell_cap, sumsq): every block of the
// binning kernel evaluates it for itself (no parameter kernel in front of it), block 0 stores the result for the later kernels.
__device__ __forceinline__ void grid_dims(const double ext[3], double &h, int32_t cell_cap, int dim[3])
{
    for (;;) {
        double tot = 1.0;
        for (int a = 0; a < 3; ++a) {
            double d = floor(ext[a] / h) + 1.0;
            if (d > 1000000.0) d = 1000000.0;
            dim[a] = (int)d; tot *= d;
        }
        if (tot <= (double)cell_cap) break;
        h *= 1.26;
    }
}
__device__ __forceinline__ GridParams grid_params_of(const double *__restrict__ bbox, int64_t n, double target, int32_t cell_cap,
                                                     const unsigned long long *__restrict__ sumsq)
{
    GridParams gp;
    double ext[3], vol = 1.0;
    for (int a = 0; a < 3; ++a) { ext[a] = bbox[3 + a] - bbox[a]; if (!(ext[a] > 1e-9)) ext[a] = 1e-9; vol *= ext[a]; }
    const double nn = (double)(n > 0 ? n : 1);
    const double h3 = cbrt(vol * target / nn);
    const double area = ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2];
    const double h2 = sqrt(area * target / nn) * 0.5;           // clouds are surfaces: size cells by area too
    double h = h3 > h2 ? h3 : h2;
    if (!(h > 0.0)) h = 1.0;
    int dim[3];
    grid_dims(ext, h, cell_cap, dim);
    if (sumsq) {
        const double occ = (double)*sumsq / nn;
        double f = pow(target / (occ > 1.0 ? occ : 1.0), 1.0 / 2.3);
        f = f < 0.125 ? 0.125 : (f > 8.0 ? 8.0 : f);
        h *= f;
        grid_dims(ext, h, cell_cap, dim);
    }
    gp.h = h;
    for (int a = 0; a < 3; ++a) { gp.org[a] = bbox[a]; gp.dim[a] = dim[a]; }
    gp.ncell = dim[0] * dim[1] * dim[2];
    return gp;
}
__global__ __launch_bounds__(256) void grid_occupancy_kernel(const uint32_t *__restrict__ cell_count, const GridParams *__restrict__ gp,
                                                             unsigned long long *__restrict__ sumsq)
{
    __shared__ unsigned long long sh[4];
    const int ncell = gp->ncell;
    unsigned long long c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += gridDim.x * blockDim.x) {
        unsigned long long v = cell_count[i];
        c += v * v;
    }
    c = block_sum(c, sh);
    if (threadIdx.x == 0 && c) atomicAdd(sumsq, c);
}
__global__ __launch_bounds__(256) void grid_cell_kernel(const float *__restrict__ pts, int64_t n, const double *__restrict__ bbox, double target,
                                                        int32_t cell_cap, const unsigned long long *__restrict__ sumsq, GridParams *gp,
                                                        uint32_t *__restrict__ keys, int32_t *__restrict__ vals, uint32_t *__restrict__ cell_count)
{
    __shared__ GridParams sg;
    if (threadIdx.x == 0) {
        sg = grid_params_of(bbox, n, target, cell_cap, sumsq);
        if (blockIdx.x == 0) *gp = sg;
    }
    __syncthreads();
    const GridParams g = sg;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int cx = cell_coord(pts[3 * i], g.org[0], g.h, g.dim[0]);
        int cy = cell_coord(pts[3 * i + 1], g.org[1], g.h, g.dim[1]);
        int cz = cell_coord(pts[3 * i + 2], g.org[2], g.h, g.dim[2]);
        uint32_t cell = (uint32_t)((cx * g.dim[1] + cy) * g.dim[2] + cz);
        keys[i] = cell;
        vals[i] = (int32_t)i;
        atomicAdd(&cell_count[cell], 1u);
    }
}
__global__ __launch_bounds__(256) void grid_gather_kernel(const float *__restrict__ pts, int64_t n, const int32_t *__restrict__ order,
                                                          float *__restrict__ sorted_pts)
{
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = order[s];
        sorted_pts[3 * s] = pts[3 * i]; sorted_pts[3 * s + 1] = pts[3 * i + 1]; sorted_pts[3 * s + 2] = pts[3 * i + 2];
    }
}
// Cell order without a radix sort (a sort of 30k keys is 9-13 launches; this is 2): every point takes the next free slot of its
// cell's range (atomic cursor: the order inside a cell is whatever the hardware made it) ...
__global__ __launch_bounds__(256) void grid_place_kernel(const uint32_t *__restrict__ cell_of, int64_t n, const uint32_t *__restrict__ cell_start,
                                                         uint32_t *__restrict__ cursor, int32_t *__restrict__ slot_idx)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t c = cell_of[i];
        slot_idx[cell_start[c] + atomicAdd(&cursor[c], 1u)] = (int32_t)i;
    }
}
// ... and one thread per point then finds its rank among the points of its cell (how many of them have a lower index: the
// cell's range is a handful of entries, read by neighbouring threads together) and moves itself there: ascending point index
// inside every cell, i.e. exactly the order a stable sort by cell gives -- deterministic, identical to the sorted build.
__global__ __launch_bounds__(256) void grid_rank_kernel(const float *__restrict__ pts, int64_t n, const uint32_t *__restrict__ cell_of,
                                                        const uint32_t *__restrict__ cell_start, const int32_t *__restrict__ slot_idx,
                                                        int32_t *__restrict__ sorted_idx, float *__restrict__ sorted_pts)
{
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        const int32_t i = slot_idx[s];
        const uint32_t c = cell_of[i];
        const int64_t s0 = cell_start[c], s1 = cell_start[c + 1];
        int rank = 0;
        for (int64_t t = s0; t < s1; ++t) rank += slot_idx[t] < i ? 1 : 0;
        const int64_t d = s0 + rank;
        sorted_idx[d] = i;
        sorted_pts[3 * d] = pts[3 * (int64_t)i]; sorted_pts[3 * d + 1] = pts[3 * (int64_t)i + 1]; sorted_pts[3 * d + 2] = pts[3 * (int64_t)i + 2];
    }
}

// Exclusive prefix sums of the cell counts in ONE launch (decoupled look-back over tiles of 2048 counters, kpx_common.h) instead of
// the vendor scan's two launches and its host-side set-up: with four frames in flight the frame's stream sat idle 50-65 us in front
// of that scan (profiles/r05/overlap_timeline_native_stream.txt).  states: one cleared 64-bit word per tile.
constexpr int kGridScanItems = 8, kGridScanTile = 256 * kGridScanItems;
constexpr int kGridScanWords = 2 * ((kGridMaxCells + 1 + kGridScanTile - 1) / kGridScanTile) + 2;
__global__ __launch_bounds__(256) void grid_scan_kernel(const uint32_t *__restrict__ counts, int64_t len, uint32_t *__restrict__ out,
                                                        unsigned long long *__restrict__ states)
{
    __shared__ int sh[256 / 64 + 2];
    const int64_t base = (int64_t)blockIdx.x * kGridScanTile + (int64_t)threadIdx.x * kGridScanItems;
    uint32_t v[kGridScanItems];
    int c = 0;
#pragma unroll
    for (int k = 0; k < kGridScanItems; ++k) { v[k] = base + k < len ? counts[base + k] : 0u; c += (int)v[k]; }
    int tot;
    const int ex = block_excl_scan(c, sh, &tot);
    __syncthreads();
    const int before = lookback_exclusive(states, blockIdx.x, tot, sh + 256 / 64 + 1);
    uint32_t run = (uint32_t)(before + ex);
#pragma unroll
    for (int k = 0; k < kGridScanItems; ++k) {
        if (base + k < len) out[base + k] = run;
        run += v[k];
    }
}

int grid_build(const float *pts, int64_t n, double target_per_cell, Arena &a, Grid *g, hipStream_t st)
{
    const size_t nn = (size_t)(n > 0 ? n : 1);
    g->params = a.get<GridParams>(1);
    // ONE cleared region per build: [16 spare words for the caller's counters | sum of squares | counts of the first binning |
    // counts of the definitive binning] (each binning has its own counters so that nothing is cleared in between)
    uint32_t *zero = a.get<uint32_t>(3 * ((size_t)kGridMaxCells + 1) + 32 + kGridExtraWords + kGridScanWords);
    g->cell_start = a.get<uint32_t>((size_t)kGridMaxCells + 1);
    g->sorted_pts = a.get<float>(nn * 3);
    g->sorted_idx = a.get<int32_t>(nn);
    uint32_t *keys_in = a.get<uint32_t>(nn), *keys_out = a.get<uint32_t>(nn);
    g->sorted_keys = keys_out;
    int32_t *vals_in = a.get<int32_t>(nn);
    double *part = a.get<double>((size_t)kBboxBlocks * 6 + 8);
    size_t sort_bytes = 0, scan_bytes = 0;
    sort_bytes = memo_bytes(2, (int64_t)nn, [&] { size_t b = 0; (void)sort_pairs(nullptr, b, keys_in, keys_out, vals_in, g->sorted_idx, (int64_t)nn, 22, st); return b; });
    scan_bytes = memo_bytes(3, 0, [&] { size_t b = 0; (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, zero, g->cell_start, kGridMaxCells + 1, st); return b; });
    char *tmp = a.get<char>(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
    RadixScratch rx{};
    const bool own_sort = (int64_t)nn > 65536 && (int64_t)nn <= kRadixMaxPairs;     // the sort build's sizes that kpx_radix.h serves
    // carved for every size (capped at the sort's limit): the workspace a caller sized for a worst-case count then also holds any
    // smaller cloud's build -- bytes(n) is monotonic (kpx_frame_step sizes one scratch region from the worst case)
    radix_carve(a, (int64_t)nn <= kRadixMaxPairs ? (int64_t)nn : kRadixMaxPairs, &rx);
    if (a.dry) return KPX_OK;
    KPX_ARENA_CHECK(a);
    double *bbox = part + (size_t)kBboxBlocks * 6;
    int rc = bbox_f32(pts, n, bbox, part, st);
    if (rc) return rc;
    // The counters and the scan cover every cell the grid may have.  Small clouds get a smaller ceiling than the 4M the
    // workspace holds (16 cells per point, at least 64k): clearing and scanning 16 MB twice per build cost ~50 us, more than
    // the neighbour search of a 30k-point cloud; a grid that would need more cells just gets coarser (results do not depend on h).
    int32_t cell_cap = 65536;
    while (cell_cap < kGridMaxCells && (int64_t)cell_cap < 16 * n) cell_cap <<= 1;
    g->spare = reinterpret_cast<int32_t *>(zero);
    unsigned long long *sumsq = reinterpret_cast<unsigned long long *>(zero + 16);
    g->extra = reinterpret_cast<int32_t *>(zero + 32);
    unsigned long long *scan_states = reinterpret_cast<unsigned long long *>(zero + 32 + kGridExtraWords);
    uint32_t *count1 = zero + 32 + kGridExtraWords + kGridScanWords, *count2 = count1 + (size_t)cell_cap + 1, *cursor = count2 + (size_t)cell_cap + 1;
    KPX_HIP(hipMemsetAsync(zero, 0, (32 + kGridExtraWords + kGridScanWords + 3 * ((size_t)cell_cap + 1)) * sizeof(uint32_t), st));
    int nb = (int)(cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256));
    // first binning from the bounding-box heuristic, one round of occupancy feedback, then the definitive binning
    hipLaunchKernelGGL(grid_cell_kernel, dim3(nb), dim3(256), 0, st, pts, n, bbox, target_per_cell, cell_cap, (const unsigned long long *)nullptr, g->params,
                       keys_in, vals_in, count1);
    hipLaunchKernelGGL(grid_occupancy_kernel, dim3(1024), dim3(256), 0, st, count1, g->params, sumsq);
    hipLaunchKernelGGL(grid_cell_kernel, dim3(nb), dim3(256), 0, st, pts, n, bbox, target_per_cell, cell_cap, (const unsigned long long *)sumsq, g->params,
                       keys_in, vals_in, count2);
    static const bool vendor_scan = [] { const char *e = getenv("KPX_GRID_SCAN"); return e && e[0] == '0'; }();      // A/B switch
    if (vendor_scan) KPX_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, scan_bytes, count2, g->cell_start, cell_cap + 1, st));
    else
        hipLaunchKernelGGL(grid_scan_kernel, dim3((unsigned)cdiv((int64_t)cell_cap + 1, kGridScanTile)), dim3(256), 0, st, count2, (int64_t)cell_cap + 1, g->cell_start,
                           scan_states);
    // The rank pass reads a cell's whole range per point: quadratic in the cell's population.  Cells are sized for 6-96 points, but
    // exact duplicates cannot be split by any grid, so the counting build is kept to frame-sized clouds (where the launch count is
    // what matters and an all-duplicates input costs at most 65536^2 range reads, ~1 s); larger clouds take the radix sort, whose
    // cost does not depend on the data.  KPX_GRID_SORT=1 forces the sort (A/B runs).
    static const bool force_sort = [] { const char *e = getenv("KPX_GRID_SORT"); return e && e[0] == '1'; }();
    const bool by_sort = force_sort || n > 65536;
    if (by_sort) {
        if (own_sort) {
            rc = radix_sort_pairs_u32(rx, keys_in, keys_out, vals_in, g->sorted_idx, n, 22, st);
            if (rc) return rc;
        } else {
            KPX_HIP(sort_pairs(tmp, sort_bytes, keys_in, keys_out, vals_in, g->sorted_idx, n, 22, st));
        }
        hipLaunchKernelGGL(grid_gather_kernel, dim3(nb), dim3(256), 0, st, pts, n, g->sorted_idx, g->sorted_pts);
    } else {
        hipLaunchKernelGGL(grid_place_kernel, dim3(nb), dim3(256), 0, st, keys_in, n, g->cell_start, cursor, vals_in);
        hipLaunchKernelGGL(grid_rank_kernel, dim3(nb), dim3(256), 0, st, pts, n, keys_in, g->cell_start, vals_in, g->sorted_idx, g->sorted_pts);
    }
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

// ---- a8 SOR ------------------------------------------------------------------------------------------
// One WAVE per query; the mean needs no identities: sum of sqrt over the selected set, ties at the k-th value counted
// k - (#smaller) times.  Queries whose candidates exceed the LDS buffer are listed for the next pass.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sor_wave_kernel(const GridParams *__restrict__ gp, const uint32_t *__restrict__ cell_start,
                                                              const float *__restrict__ spts, const int32_t *__restrict__ sidx,
                                                              int64_t q0, int64_t q1, int k, int cap, double *__restrict__ avg,
                                                              const int32_t *__restrict__ in_list, const int32_t *__restrict__ in_count,
                                                              int32_t *__restrict__ fb_list, int32_t *__restrict__ fb_count)
{
    // queries = the cell-sorted positions [q0, q1) (all of them: 0, n; a slab of the grid order for kpx_sor_partial).
    // sidx != NULL: avg is indexed by the caller's point index; sidx == NULL: by sorted position relative to q0.
    extern __shared__ __align__(16) double lds[];
    __shared__ uint32_t run_s0[WAVES][64];
    __shared__ int32_t run_off[WAVES][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ __align__(16) uint32_t knn_hist[WAVES][kKnnBuckets];
    const WaveKnnScratch sc{ lds + (size_t)wave * cap, nullptr, run_s0[wave], run_off[wave], cap, knn_hist[wave] };
    const GridParams g = *gp;
    const int64_t nq = in_list ? (int64_t)*in_count : q1 - q0;   // in_list: the queries an earlier pass could not hold
    for (int64_t e = (int64_t)blockIdx.x * WAVES + wave; e < nq; e += (int64_t)gridDim.x * WAVES) {
        const int64_t s = in_list ? (int64_t)in_list[e] : q0 + e;
        const double q[3] = { (double)spts[3 * s], (double)spts[3 * s + 1], (double)spts[3 * s + 2] };
        WaveKnnResult res;
        if (!wave_knn_select<false>(g, cell_start, spts, q, k, INFINITY, sc, res)) {
            if (lane == 0) fb_list[atomicAdd(fb_count, 1)] = (int32_t)s;
            continue;
        }
        double sum = 0.0;
        for (int t = lane; t < res.m; t += 64) {
            const double d = sc.vals[t];
            if ((unsigned long long)__double_as_longlong(d) <= res.thr) sum += sqrt(d);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const double total = sum - (double)(res.cnt - res.kk) * sqrt(res.top);      // ties beyond the k-th slot
        if (lane == 0) avg[sidx ? (int64_t)sidx[s] : s - q0] = res.kk > 0 ? total / (double)res.kk : -1.0;
        wave_lds_fence();
    }
}

// ---- a8 SOR, the cell's queries together (round 4) -------------------------------------------------------------------------
// The wave-per-query kernel above re-gathers the 27-cell block for every query -- binary search over the cell runs, loads, 64-lane
// ballots per bisection step -- although the queries of a cell share that block.  Here a wave takes 16 consecutive cell-sorted
// queries; every group of L lanes takes one of the CELLS among them (64 / L cells side by side: their loads are in flight together),
// stages the cell's 27-cell block ONCE -- nine contiguous runs of the cell-sorted points, one per (x, y) column -- in its LDS region
// and answers the cell's queries one after the other, each from the two words of the S squared distances a lane holds in REGISTERS
// (lane gl: candidates gl, gl + L, ...; AC3, fp64; d^2 >= 0, so the IEEE patterns order like the values).  (Keeping the candidates
// themselves in registers across the cell's queries was tried: the compiler hoists their float -> double conversions and spills.)  Queries the 3 x 3 x 3 block does not cover get the 5 x 5 x 5 block the same way.  The k-th smallest is found by
// bisection on the high words (a 32-bit compare + add per candidate and step, the count folded over the L lanes by DPP row
// rotations; it ends as soon as a pivot has exactly k patterns below it), the low words are looked at only when several candidates
// share the k-th high word.  Exactness is the ring walk's rule: the query is answered here only if its k-th candidate lies inside
// the distance the block covers (block_cover2), everything else -- blocks that do not cover, blocks with more than L x S points,
// ties far beyond k -- goes to the wave-per-query passes through a list.  The block, its order (column by column, cell-sorted
// position inside a column) and with it every sum depend on the query's CELL alone, not on how the queries were dealt out: the
// slab calls of the sharded filter (kpx_sor_partial) give bitwise the values of the one-GPU call.
// Roof: per query 6 S L fp64 flops for the distances and ~(2 S + 8) 32-bit operations per bisection step on L lanes -- vector
// ALU work on LDS-resident operands (12 B per candidate staged once per cell and read G x per query group); DESIGN.md section 5.
__device__ __forceinline__ int dppi(int v, int ctrl_sel)
{
    switch (ctrl_sel) {
    case 1: return __builtin_amdgcn_update_dpp(v, v, 0x121, 0xF, 0xF, false);
    case 2: return __builtin_amdgcn_update_dpp(v, v, 0x122, 0xF, 0xF, false);
    case 4: return __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, false);
    default: return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);
    }
}
// reductions over the L lanes of a query group (L = 16: one DPP row; 32, 64: the rows are folded through the LDS crossbar);
// integer results are the same in every lane of the group
template <int L> __device__ __forceinline__ int group_sum_i(int v)
{
    v += dppi(v, 1); v += dppi(v, 2); v += dppi(v, 4); v += dppi(v, 8);
    if (L >= 32) v += __shfl_xor(v, 16, 64);
    if (L >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
template <int L> __device__ __forceinline__ unsigned group_min_u(unsigned v)
{
    unsigned o;
    o = (unsigned)dppi((int)v, 1); v = o < v ? o : v; o = (unsigned)dppi((int)v, 2); v = o < v ? o : v;
    o = (unsigned)dppi((int)v, 4); v = o < v ? o : v; o = (unsigned)dppi((int)v, 8); v = o < v ? o : v;
    if (L >= 32) { o = (unsigned)__shfl_xor((int)v, 16, 64); v = o < v ? o : v; }
    if (L >= 64) { o = (unsigned)__shfl_xor((int)v, 32, 64); v = o < v ? o : v; }
    return v;
}
template <int L> __device__ __forceinline__ unsigned group_max_u(unsigned v)
{
    unsigned o;
    o = (unsigned)dppi((int)v, 1); v = o > v ? o : v; o = (unsigned)dppi((int)v, 2); v = o > v ? o : v;
    o = (unsigned)dppi((int)v, 4); v = o > v ? o : v; o = (unsigned)dppi((int)v, 8); v = o > v ? o : v;
    if (L >= 32) { o = (unsigned)__shfl_xor((int)v, 16, 64); v = o > v ? o : v; }
    if (L >= 64) { o = (unsigned)__shfl_xor((int)v, 32, 64); v = o > v ? o : v; }
    return v;
}
#ifndef KPX_SOR_CELL_MINB8
#define KPX_SOR_CELL_MINB8 6
#endif
#ifndef KPX_SOR_CELL_MINB
#define KPX_SOR_CELL_MINB 4
#endif
#ifndef KPX_SOR_CELL_MINW
#define KPX_SOR_CELL_MINW 4
#endif
#ifdef KPX_SOR_CELL_STATS
// diagnostic build only (tools/sor_stats.py): wave clocks per phase, summed over the waves
__device__ unsigned long long g_sor_stats[16];
#define KPX_STAT_CLK(var) const long long var = (long long)clock64()
#define KPX_STAT_ADD(slot, v) do { if (lane == 0) atomicAdd(&g_sor_stats[slot], (unsigned long long)(v)); } while (0)
#else
#define KPX_STAT_CLK(var)
#define KPX_STAT_ADD(slot, v)
#endif
constexpr int kCellBuckets = 256;             // buckets of the counting selection (round 5)
// doubles of LDS per group of L lanes: bucket counts, selection buffer, run table (32 starts, 32 offsets), the block's points; even, so
// that every group's region starts on 16 bytes
__host__ __device__ constexpr size_t sor_cell_group_doubles(int kbuf, int cap)
{
    return ((size_t)kCellBuckets / 2 + (size_t)kbuf + 32 + (size_t)(cap * 3 + 1) / 2 + 1) & ~(size_t)1;
}
constexpr int kCellQueries = 16;              // consecutive cell-sorted queries per wave and trip
constexpr int kCellTieRoom = 32;              // candidates tied with the k-th beyond k the selection buffer still holds
template <int L, int S, int W>
__global__ __launch_bounds__(64 * W, (S > 16 ? 2 : (S > 8 ? 3 : KPX_SOR_CELL_MINW))) void sor_cell_kernel(
    const GridParams *__restrict__ gp, const uint32_t *__restrict__ cell_start, const float *__restrict__ spts, const int32_t *__restrict__ sidx,
    int64_t q0, int64_t q1, int k, int kbuf, double *__restrict__ avg, int32_t *__restrict__ fb_list, int32_t *__restrict__ fb_count)
{
    constexpr int CAP = L * S, G = 64 / L;
    extern __shared__ __align__(16) double lds[];
    // per wave and group: a selection buffer of kbuf doubles, the run table of the group's block (32 starts, 32 offsets) and the block's
    // points (CAP x 3 floats)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane / L, gl = lane % L;
    const size_t per_group = sor_cell_group_doubles(kbuf, CAP);
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds + ((size_t)wave * G + grp) * per_group);       // 256 bucket counts (16-byte aligned)
    double *selbuf = reinterpret_cast<double *>(hist + kCellBuckets);
    uint32_t *run_s0 = reinterpret_cast<uint32_t *>(selbuf + kbuf);
    int32_t *run_off = reinterpret_cast<int32_t *>(run_s0 + 32);
    float *cand = reinterpret_cast<float *>(selbuf + kbuf + 32);
    const GridParams g = *gp;
    const int64_t nq = q1 - q0;
    KPX_STAT_CLK(t_wave0);
    for (int64_t chunk = (int64_t)blockIdx.x * W + wave; chunk * kCellQueries < nq; chunk += (int64_t)gridDim.x * W) {
        const int64_t base = q0 + chunk * kCellQueries;
        const int nqc = (int)(q1 - base < kCellQueries ? q1 - base : kCellQueries);
        // lanes 0 .. nqc-1: the chunk's queries and their cells; a cell's queries are contiguous (cell-sorted)
        double myq[3] = { 0.0, 0.0, 0.0 };
        int myc[3] = { 0, 0, 0 };
        int64_t mycell = -1;
        if (lane < nqc) {
            const float *qp = spts + 3 * (base + lane);
#pragma unroll
            for (int a = 0; a < 3; ++a) { myq[a] = (double)qp[a]; myc[a] = cell_coord(myq[a], g.org[a], g.h, g.dim[a]); }
            mycell = ((int64_t)myc[0] * g.dim[1] + myc[1]) * g.dim[2] + myc[2];
        }
        const int64_t prevcell = __shfl_up(mycell, 1, 64);
        unsigned segs = (unsigned)__builtin_amdgcn_ballot_w64(lane < nqc && (lane == 0 || mycell != prevcell));     // bit i: query i opens a cell
        unsigned fbmask = 0u;                                    // queries of this chunk that go to the wave-per-query passes
        while (segs != 0u) {                                     // G cells at a time, one per group of L lanes
            int pos = 0, len = 0;
            {
                unsigned tt = segs;
#pragma unroll
                for (int gg = 0; gg < G; ++gg) {
                    if (tt != 0u) {
                        const int p0 = __builtin_ctz(tt);
                        tt &= tt - 1u;
                        const int p1 = tt ? __builtin_ctz(tt) : nqc;
                        if (gg == grp) { pos = p0; len = p1 - p0; }
                    }
                }
                segs = tt;
            }
            const int cx = __shfl(myc[0], pos, 64), cy = __shfl(myc[1], pos, 64), cz = __shfl(myc[2], pos, 64);
            unsigned open = len > 0 ? (1u << len) - 1u : 0u;     // the cell's queries still unanswered (bit = query pos + i)
            unsigned fbbits = 0u;
            // the 3 x 3 x 3 block first; the queries it does not cover get the 5 x 5 x 5 block (as long as it fits the lanes)
            for (int rad = 1; rad <= 2; ++rad) {
                bool active = open != 0u;
                if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
                KPX_STAT_CLK(t_st0);
                const int side = 2 * rad + 1, nruns = side * side;
                // the block's runs (one per (x, y) column, ascending), two per lane where the group has fewer lanes than runs
                uint32_t s0[2] = { 0u, 0u };
                int ln[2] = { 0, 0 };
#pragma unroll
                for (int pss = 0; pss < 2; ++pss) {
                    const int rr = gl + pss * L;
                    if (active && rr < nruns) {
                        const int x = cx + rr / side - rad, y = cy + rr % side - rad;
                        if (x >= 0 && x < g.dim[0] && y >= 0 && y < g.dim[1]) {
                            const int64_t col = ((int64_t)x * g.dim[1] + y) * g.dim[2];
                            const int za = cz - rad < 0 ? 0 : cz - rad, zb = cz + rad >= g.dim[2] ? g.dim[2] - 1 : cz + rad;
                            s0[pss] = cell_start[col + za];
                            ln[pss] = (int)(cell_start[col + zb + 1] - s0[pss]);
                        }
                    }
                }
                int inc0 = ln[0], inc1 = ln[1];
#pragma unroll
                for (int o = 1; o < L; o <<= 1) {
                    const int t0 = __shfl_up(inc0, o, L), t1 = __shfl_up(inc1, o, L);
                    if (gl >= o) { inc0 += t0; inc1 += t1; }
                }
                const int tot0 = __shfl(inc0, L - 1, L);
                const int m = tot0 + __shfl(inc1, L - 1, L);
                if (m > CAP) active = false;                     // more points than the group's lanes hold: the queries stay open
                if (active) {
                    if (gl < 32) { run_s0[gl] = s0[0]; run_off[gl] = gl < nruns ? inc0 - ln[0] : INT_MAX; }
                    if (L < 32 && gl + L < 32) { run_s0[gl + L] = s0[1]; run_off[gl + L] = gl + L < nruns ? tot0 + inc1 - ln[1] : INT_MAX; }
                }
                wave_lds_fence();
                const int mmax = [&] { int v = active ? m : 0;
#pragma unroll
                                       for (int o = 32; o > 0; o >>= 1) { const int t2 = __shfl_xor(v, o, 64); v = t2 > v ? t2 : v; }
                                       return v; }();
                // the block's points -> the group's LDS region (lane gl brings in candidates gl, gl + L, ...: all loads of a lane in flight together)
#pragma unroll
                for (int sl = 0; sl < S; ++sl) {
                    const int jn = sl * L + gl;
                    if (sl * L < mmax && active && jn < m) {
                        int lo_r = 0, hi_r = nruns - 1;          // last run with off <= jn
                        while (lo_r < hi_r) {
                            const int mid = (lo_r + hi_r + 1) >> 1;
                            if (run_off[mid] <= jn) lo_r = mid; else hi_r = mid - 1;
                        }
                        const float *pp = spts + 3 * (int64_t)(run_s0[lo_r] + (uint32_t)(jn - run_off[lo_r]));
                        cand[3 * jn] = pp[0]; cand[3 * jn + 1] = pp[1]; cand[3 * jn + 2] = pp[2];
                    }
                }
                wave_lds_fence();
                KPX_STAT_CLK(t_st1);
                KPX_STAT_ADD(8, t_st1 - t_st0); KPX_STAT_ADD(7, 1); KPX_STAT_ADD(6, mmax);
                unsigned todo = active ? open : 0u;
                while (__builtin_amdgcn_ballot_w64(todo != 0u) != 0ull) {       // one open query of every group's cell per trip
                    KPX_STAT_CLK(t_q0);
                    const bool have = todo != 0u;
                    const int qrel = have ? __builtin_ctz(todo) : 0;
                    todo &= todo - 1u;
                    const int qi = pos + qrel;
                    const double q[3] = { __shfl(myq[0], qi, 64), __shfl(myq[1], qi, 64), __shfl(myq[2], qi, 64) };
                    const int c[3] = { cx, cy, cz };
                    unsigned hi[S], lo[S];
#pragma unroll
                    for (int sl = 0; sl < S; ++sl) {
                        hi[sl] = 0xFFFFFFFFu; lo[sl] = 0xFFFFFFFFu;
                        const int jn = sl * L + gl;
                        if (sl * L < mmax && have && jn < m) {
                            const double dx = q[0] - (double)cand[3 * jn], dy = q[1] - (double)cand[3 * jn + 1], dz = q[2] - (double)cand[3 * jn + 2];
                            const unsigned long long pat = (unsigned long long)__double_as_longlong(fma(dz, dz, fma(dy, dy, dx * dx)));
                            hi[sl] = (unsigned)(pat >> 32); lo[sl] = (unsigned)pat;
                        }
                    }
                    // inside the covered distance: at least kk candidates strictly nearer than the block's cover (the ring walk's rule)
                    const double cov2 = block_cover2(g, q, c, rad);
                    const bool whole = cov2 == INFINITY;
                    const unsigned long long cpat = (unsigned long long)__double_as_longlong(cov2);
                    const unsigned ch = (unsigned)(cpat >> 32), cl = (unsigned)cpat;
                    int cnt = 0;
#pragma unroll
                    for (int sl = 0; sl < S; ++sl) if (sl * L < mmax) cnt += (hi[sl] < ch || (hi[sl] == ch && lo[sl] < cl)) ? 1 : 0;
                    cnt = group_sum_i<L>(cnt);
                    const int kk = whole ? (m < k ? m : k) : k;
                    bool ok = have && cnt >= kk;
                    // phase 1: the k-th smallest HIGH word
                    unsigned vmin = 0xFFFFFFFFu, vmax = 0u;
#pragma unroll
                    for (int sl = 0; sl < S; ++sl) if (sl * L < mmax) { vmin = hi[sl] < vmin ? hi[sl] : vmin; const unsigned hv = hi[sl] == 0xFFFFFFFFu ? 0u : hi[sl]; vmax = hv > vmax ? hv : vmax; }
                    unsigned a = group_min_u<L>(vmin), b = group_max_u<L>(vmax);
                    bool exact = false;                          // a pivot with exactly kk patterns at or below it was met
                    bool run = ok && a < b;
                    KPX_STAT_CLK(t_q1);
                    KPX_STAT_ADD(9, t_q1 - t_q0);
#ifdef KPX_SOR_CELL_BISECT
                    while (__builtin_amdgcn_ballot_w64(run) != 0ull) {
                        const unsigned mid = a + ((b - a) >> 1);
                        int c1 = 0;
#pragma unroll
                        for (int sl = 0; sl < S; ++sl) if (sl * L < mmax) c1 += hi[sl] <= mid ? 1 : 0;
                        c1 = group_sum_i<L>(c1);
                        if (run) {
                            if (c1 == kk) { a = b = mid; exact = true; }
                            else if (c1 > kk) b = mid;
                            else a = mid + 1;
                        }
                        run = ok && a < b;
                    }
#else
                    // Round 5: by COUNTING instead of bisecting (a bisection step costs 2 S + 8 instructions and the high words span
                    // ~2^23 values: ~24 steps).  The window [a, a + range] that holds the k-th is cut into <= 256 buckets of 2^shift;
                    // every candidate inside adds one to its bucket (LDS atomics), lane gl sums its 256 / L buckets, a scan over the
                    // group finds the bucket of the k-th, and the next level looks inside that bucket only.  It ends when a bucket is
                    // one value wide, or -- nearly always after two levels -- when the k-th is the LAST of its bucket: then the bucket's
                    // upper edge is a pivot with exactly kk patterns at or below it, the bisection's `exact` case.
                    {
                        constexpr int BPL = kCellBuckets / L;            // buckets per lane
                        unsigned range = b - a;
                        int need_k = kk;                                 // rank of the k-th inside the window, 1-based
                        while (__builtin_amdgcn_ballot_w64(run) != 0ull) {
                            KPX_STAT_ADD(13, 1);
                            const int shift = range >= 256u ? 24 - __builtin_clz(range) : 0;      // range >> shift <= 255
                            uint4 *hz = reinterpret_cast<uint4 *>(hist + gl * BPL);
#pragma unroll
                            for (int w = 0; w < BPL / 4; ++w) hz[w] = make_uint4(0u, 0u, 0u, 0u);
                            wave_lds_fence();
#pragma unroll
                            for (int sl = 0; sl < S; ++sl)
                                if (sl * L < mmax) {
                                    const unsigned rel = hi[sl] - a;
                                    if (run && rel <= range) __hip_atomic_fetch_add(hist + (rel >> shift), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                            wave_lds_fence();
                            unsigned hv[BPL];
#pragma unroll
                            for (int w = 0; w < BPL / 4; ++w) { const uint4 t4 = hz[w]; hv[4 * w] = t4.x; hv[4 * w + 1] = t4.y; hv[4 * w + 2] = t4.z; hv[4 * w + 3] = t4.w; }
                            int lsum = 0;
#pragma unroll
                            for (int w = 0; w < BPL; ++w) lsum += (int)hv[w];
                            int incl = lsum;
#pragma unroll
                            for (int o = 1; o < L; o <<= 1) { const int t2 = __shfl_up(incl, o, L); if (gl >= o) incl += t2; }
                            const int excl = incl - lsum;
                            const bool mine = run && need_k > excl && need_k <= incl;
                            int B = -1, cb = 0, hb = 0;
                            if (mine) {
                                int cum = excl;
#pragma unroll
                                for (int w = 0; w < BPL; ++w) {
                                    if (B < 0 && need_k <= cum + (int)hv[w]) { B = gl * BPL + w; cb = cum; hb = (int)hv[w]; }
                                    cum += (int)hv[w];
                                }
                            }
                            const unsigned long long bm = __builtin_amdgcn_ballot_w64(mine);
                            const unsigned long long gm = L == 64 ? bm : ((bm >> (grp * L)) & ((1ull << (L & 63)) - 1ull));
                            const int owner = gm ? __builtin_ctzll(gm) : 0;
                            B = __shfl(B, owner, L); cb = __shfl(cb, owner, L); hb = __shfl(hb, owner, L);
                            if (run) {
                                if (B < 0) { ok = false; run = false; }                       // cannot happen (cnt >= kk); the query would go to the lists
                                else if (need_k - cb == hb) {                                 // the k-th is the last of its bucket
                                    a = a + (((unsigned)B + 1u) << shift) - 1u; exact = true; run = false;
                                } else if (shift == 0) { a = a + (unsigned)B; run = false; }  // one value wide: THE k-th high word
                                else { need_k -= cb; a += (unsigned)B << shift; range = (1u << shift) - 1u; }
                            }
                        }
                    }
#endif
                    const unsigned P = a;
                    unsigned Q = 0xFFFFFFFFu;
                    int n_le = kk;
                    if (__builtin_amdgcn_ballot_w64(ok && !exact) != 0ull) {
                        int c_less = 0, c_eq = 0;
#pragma unroll
                        for (int sl = 0; sl < S; ++sl) if (sl * L < mmax) { c_less += hi[sl] < P ? 1 : 0; c_eq += hi[sl] == P ? 1 : 0; }
                        c_less = group_sum_i<L>(c_less); c_eq = group_sum_i<L>(c_eq);
                        n_le = c_less + c_eq;
                        // phase 2 (several candidates share the k-th high word and not all of them fit): the low words decide
                        const int r2 = kk - c_less;
                        const bool need = ok && !exact && n_le > kk;
                        if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
                            unsigned lmin = 0xFFFFFFFFu, lmax = 0u;
#pragma unroll
                            for (int sl = 0; sl < S; ++sl) if (hi[sl] == P) { lmin = lo[sl] < lmin ? lo[sl] : lmin; lmax = lo[sl] > lmax ? lo[sl] : lmax; }
                            unsigned a2 = group_min_u<L>(lmin), b2 = group_max_u<L>(lmax);
                            bool run2 = need && a2 < b2;
                            while (__builtin_amdgcn_ballot_w64(run2) != 0ull) {
                                const unsigned mid = a2 + ((b2 - a2) >> 1);
                                int c2 = 0;
#pragma unroll
                                for (int sl = 0; sl < S; ++sl) c2 += (hi[sl] == P && lo[sl] <= mid) ? 1 : 0;
                                c2 = group_sum_i<L>(c2);
                                if (run2) { if (c2 >= r2) b2 = mid; else a2 = mid + 1; }
                                run2 = need && a2 < b2;
                            }
                            int c3 = 0;
#pragma unroll
                            for (int sl = 0; sl < S; ++sl) c3 += (hi[sl] == P && lo[sl] <= a2) ? 1 : 0;
                            c3 = group_sum_i<L>(c3);
                            if (need) { Q = a2; n_le = c_less + c3; }
                        }
                        if (exact) n_le = kk;
                    }
                    KPX_STAT_CLK(t_q2);
                    KPX_STAT_ADD(10, t_q2 - t_q1);
                    // the selected patterns (all <= (P, Q)), in the block's own order -> selection buffer -> sum of square roots
                    const bool tied_out = ok && n_le > kk + kCellTieRoom;     // ties far beyond k (duplicates, lattices): the wave-per-query passes
                    if (tied_out) ok = false;
                    int mine = 0;
#pragma unroll
                    for (int sl = 0; sl < S; ++sl) if (sl * L < mmax) mine += (hi[sl] < P || (hi[sl] == P && lo[sl] <= Q)) ? 1 : 0;
                    int before = mine;                            // exclusive prefix over the group's lanes
#pragma unroll
                    for (int o = 1; o < L; o <<= 1) { const int t2 = __shfl_up(before, o, L); if (gl >= o) before += t2; }
                    before -= mine;
                    if (ok) {
                        int w = before;
#pragma unroll
                        for (int sl = 0; sl < S; ++sl)
                            if (sl * L < mmax && (hi[sl] < P || (hi[sl] == P && lo[sl] <= Q)))
                                selbuf[w++] = __longlong_as_double((long long)(((unsigned long long)hi[sl] << 32) | lo[sl]));
                    }
                    wave_lds_fence();
                    double acc = 0.0;
                    if (ok)
                        for (int t = gl; t < n_le; t += L) acc += sqrt(selbuf[t]);
                    // fixed tree over the group's lanes; lane 0 of the group holds THE sum
#pragma unroll
                    for (int o = L / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, L);
                    if (ok && gl == 0) {
                        const int64_t sq = base + qi;
                        const double top = __longlong_as_double((long long)(((unsigned long long)P << 32) | Q));
                        const double total = n_le > kk ? acc - (double)(n_le - kk) * sqrt(top) : acc;
                        avg[sidx ? (int64_t)sidx[sq] : sq - q0] = kk > 0 ? total / (double)kk : -1.0;
                    }
                    if (ok || tied_out) open &= ~(1u << qrel);   // settled, or straight to the list
                    if (tied_out) fbbits |= 1u << qi;
                    wave_lds_fence();
                    KPX_STAT_CLK(t_q3);
                    KPX_STAT_ADD(11, t_q3 - t_q2);
                    KPX_STAT_ADD(rad == 1 ? 0 : 1, __builtin_popcountll(__builtin_amdgcn_ballot_w64(ok && gl == 0)));
                }
            }
            fbbits |= open << pos;                               // neither block settled them
            // the groups' lists -> the chunk's
            unsigned fbv = fbbits;
            if (L < 64) fbv |= (unsigned)__shfl_xor((int)fbv, 32, 64);
            if (L < 32) fbv |= (unsigned)__shfl_xor((int)fbv, 16, 64);
            fbmask |= (unsigned)__builtin_amdgcn_readfirstlane((int)fbv);
        }
        if (fbmask != 0u) {                                      // ONE counter update per chunk (same-address atomics serialise)
            int fb_base = 0;
            if (lane == 0) fb_base = atomicAdd(fb_count, __builtin_popcount(fbmask));
            fb_base = __shfl(fb_base, 0, 64);
            if (lane < 16 && ((fbmask >> lane) & 1u)) fb_list[fb_base + __builtin_popcount(fbmask & ((1u << lane) - 1u))] = (int32_t)(base + lane);
        }
    }
#ifdef KPX_SOR_CELL_STATS
    KPX_STAT_ADD(12, (long long)clock64() - t_wave0);
#endif
}

// ---- a8 SOR, large k: a BLOCK per 64 cell-sorted queries (round 5) -------------------------------------------------------------------
// The kernel above gives every wave its own 16 queries and its own copy of their cells' blocks: at k = 200 (filter_outliers' default,
// preprocessing/data.py:61) that is 27 KB of LDS per wave -- five waves per CU, 247 registers -- and a cell of ~80 queries is staged by
// five or six different waves; the phase clocks (tools/sor_stats.py) put 31 % of the wave time into staging and most of the rest into
// dependent LDS / cross-lane round trips nothing overlaps.  Here four waves share what they can: a block takes 64 consecutive cell-sorted
// queries, stages the 27-cell block of each CELL among them once with all 256 threads (a fixed-depth run search per point: the loads of a
// thread's eight points are in flight together), and its waves answer the cell's queries side by side -- wave w the w-th, (w + 4)-th, ...
// -- each from the squared distances its lanes hold in registers (lane l: candidates l, l + 64, ...).  36 KB of LDS per block at k = 200:
// sixteen waves per CU.  The arithmetic is that of sor_cell_kernel<64, S, .>: the k-th high word by counting into 256 buckets (its window
// starts at the smallest NON-ZERO high word and ends at the cover's), low words only for ties; the selected patterns go into the selection
// buffer in the block's own order (a ballot per row of 64 candidates places them) and the sum of their square roots is a fixed tree --
// every mean depends on the query's cell alone, not on how the queries were dealt out.  Queries the 3 x 3 x 3 block does not cover get the 5 x 5 x 5 block; what neither settles
// goes to the wave-per-query passes through the list.
constexpr int kBlkQueries = 64, kBlkWaves = 4;
#ifndef KPX_SOR_BLOCK_MINB
#define KPX_SOR_BLOCK_MINB 4
#endif
template <int S>
__global__ __launch_bounds__(64 * kBlkWaves, KPX_SOR_BLOCK_MINB) void sor_block_kernel(
    const GridParams *__restrict__ gp, const uint32_t *__restrict__ cell_start, const float *__restrict__ spts, const int32_t *__restrict__ sidx,
    int64_t q0, int64_t q1, int k, int kbuf, double *__restrict__ avg, int32_t *__restrict__ fb_list, int32_t *__restrict__ fb_count)
{
    constexpr int CAP = 64 * S, W = kBlkWaves, T = 64 * kBlkWaves;
    extern __shared__ __align__(16) double lds[];
    __shared__ uint32_t run_s0[32];
    __shared__ int32_t run_off[32];
    __shared__ float qf[3][kBlkQueries];
    __shared__ int32_t qc[3][kBlkQueries];
    __shared__ int32_t qcell[kBlkQueries], qsidx[kBlkQueries];
    __shared__ uint8_t qstate[kBlkQueries];              // 0 open, 1 settled, 2 ties beyond the buffer (straight to the list)
    __shared__ int32_t s_m;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float *cand = reinterpret_cast<float *>(lds);                                                   // CAP x 3 floats
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds + (size_t)(CAP * 3) / 2) + (size_t)wave * kCellBuckets;
    double *selbuf = lds + (size_t)(CAP * 3) / 2 + (size_t)W * kCellBuckets / 2 + (size_t)wave * kbuf;
    const GridParams g = *gp;
    const int64_t nq = q1 - q0;
    for (int64_t chunk = blockIdx.x; chunk * kBlkQueries < nq; chunk += gridDim.x) {
        const int64_t base = q0 + chunk * kBlkQueries;
        const int nqc = (int)(q1 - base < kBlkQueries ? q1 - base : kBlkQueries);
        if (tid < nqc) {
            const float *qp = spts + 3 * (base + tid);
            int c[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { const float v = qp[a]; qf[a][tid] = v; c[a] = cell_coord((double)v, g.org[a], g.h, g.dim[a]); qc[a][tid] = c[a]; }
            qcell[tid] = (int32_t)(((int64_t)c[0] * g.dim[1] + c[1]) * g.dim[2] + c[2]);
            qsidx[tid] = sidx ? sidx[base + tid] : (int32_t)(base + tid - q0);
            qstate[tid] = 0;
        }
        __syncthreads();
        // the chunk's cells: a cell's queries are contiguous (cell-sorted); every wave derives the same mask
        const int mycell = lane < nqc ? qcell[lane] : -1;
        const int prevcell = __shfl_up(mycell, 1, 64);
        unsigned long long segs = __builtin_amdgcn_ballot_w64(lane < nqc && (lane == 0 || mycell != prevcell));
        while (segs != 0ull) {
            const int pos = __builtin_ctzll(segs);
            segs &= segs - 1ull;
            const int len = (segs ? __builtin_ctzll(segs) : nqc) - pos;
            const int cx = qc[0][pos], cy = qc[1][pos], cz = qc[2][pos];
            for (int rad = 1; rad <= 2; ++rad) {
                if (rad == 2 && __builtin_amdgcn_ballot_w64(lane < len && qstate[pos + lane] == 0) == 0ull) break;       // block-uniform: qstate is settled
                const int side = 2 * rad + 1, nruns = side * side;
                if (wave == 0) {                                 // the block's runs, one per (x, y) column, ascending
                    uint32_t s0 = 0u;
                    int ln = 0;
                    if (lane < nruns) {
                        const int x = cx + lane / side - rad, y = cy + lane % side - rad;
                        if (x >= 0 && x < g.dim[0] && y >= 0 && y < g.dim[1]) {
                            const int64_t col = ((int64_t)x * g.dim[1] + y) * g.dim[2];
                            const int za = cz - rad < 0 ? 0 : cz - rad, zb = cz + rad >= g.dim[2] ? g.dim[2] - 1 : cz + rad;
                            s0 = cell_start[col + za];
                            ln = (int)(cell_start[col + zb + 1] - s0);
                        }
                    }
                    const int incl = wave_incl_scan(ln);
                    if (lane < 32) { run_s0[lane] = s0; run_off[lane] = lane < nruns ? incl - ln : INT_MAX; }
                    if (lane == 63) s_m = incl;
                }
                __syncthreads();
                const int m = s_m;
                if (m <= CAP) {
                    // the block's points -> LDS: thread t brings in points t, t + 256, ...; the run of a point by a fixed-depth search
#pragma unroll
                    for (int tr = 0; tr < CAP / T; ++tr) {
                        const int jn = tr * T + tid;
                        if (jn < m) {
                            int r = 0;
#pragma unroll
                            for (int st = 16; st > 0; st >>= 1) { const int cr = r + st; if (cr < nruns && run_off[cr] <= jn) r = cr; }
                            const float *pp = spts + 3 * (int64_t)(run_s0[r] + (uint32_t)(jn - run_off[r]));
                            cand[3 * jn] = pp[0]; cand[3 * jn + 1] = pp[1]; cand[3 * jn + 2] = pp[2];
                        }
                    }
                }
                __syncthreads();
                if (m <= CAP) {
                    const int c[3] = { cx, cy, cz };
                    for (int qi = pos + wave; qi < pos + len; qi += W) {
                        if (qstate[qi] != 0) continue;           // wave-uniform
                        const double q[3] = { (double)qf[0][qi], (double)qf[1][qi], (double)qf[2][qi] };
                        // inside the covered distance: at least kk candidates strictly nearer than the block's cover (the ring walk's rule);
                        // d^2 >= 0, so the IEEE patterns order like the values
                        const double cov2 = block_cover2(g, q, c, rad);
                        const bool whole = cov2 == INFINITY;
                        const unsigned ch = (unsigned)((unsigned long long)__double_as_longlong(cov2) >> 32);
                        unsigned hi[S], lo[S];
                        int cnt = 0, zeros = 0;                  // zeros: distances with a zero high word (the query itself, its duplicates)
                        unsigned vmin = 0xFFFFFFFFu;             // smallest non-zero high word
                        int vmax = -1;                           // largest high word (an empty slot's 0xFFFFFFFF is -1)
#pragma unroll
                        for (int sl = 0; sl < S; ++sl) {
                            hi[sl] = 0xFFFFFFFFu; lo[sl] = 0xFFFFFFFFu;
                            const int jn = sl * 64 + lane;
                            if (sl * 64 < m && jn < m) {
                                const double dx = q[0] - (double)cand[3 * jn], dy = q[1] - (double)cand[3 * jn + 1], dz = q[2] - (double)cand[3 * jn + 2];
                                const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                                const unsigned long long pat = (unsigned long long)__double_as_longlong(d2);
                                hi[sl] = (unsigned)(pat >> 32); lo[sl] = (unsigned)pat;
                                cnt += d2 < cov2 ? 1 : 0;
                                zeros += hi[sl] == 0u ? 1 : 0;
                                const unsigned nz = hi[sl] == 0u ? 0xFFFFFFFFu : hi[sl];
                                vmin = nz < vmin ? nz : vmin;
                                vmax = (int)hi[sl] > vmax ? (int)hi[sl] : vmax;
                            }
                        }
                        cnt = group_sum_i<64>(cnt);
                        zeros = group_sum_i<64>(zeros);
                        const int kk = whole ? (m < k ? m : k) : k;
                        bool ok = cnt >= kk;                     // wave-uniform from here on
                        // The window of the k-th high word: from the smallest NON-ZERO one (the query is its own nearest candidate: with
                        // zero in the window the first level's buckets are eight binades wide and hold everything else in two or three
                        // of them) up to the largest, or the cover's -- at least kk candidates lie strictly inside the cover
                        unsigned a = group_min_u<64>(vmin);
                        unsigned b = group_max_u<64>((unsigned)(vmax < 0 ? 0 : vmax));
                        if (!whole && b > ch) b = ch;
                        bool exact = false;                      // a pivot with exactly kk patterns at or below it was met
                        {
                            constexpr int BPL = kCellBuckets / 64;
                            int need_k = kk - zeros;             // rank of the k-th among the candidates at or above the window's start
                            if (need_k <= 0) a = b = 0u;         // the k-th is one of the zero distances
                            unsigned range = b - a;
                            bool run = ok && a < b;
                            while (run) {
                                const int shift = range >= 256u ? 24 - __builtin_clz(range) : 0;
                                uint4 *hz = reinterpret_cast<uint4 *>(hist + lane * BPL);
                                hz[0] = make_uint4(0u, 0u, 0u, 0u);
                                wave_lds_fence();
#pragma unroll
                                for (int sl = 0; sl < S; ++sl)
                                    if (sl * 64 < m) {
                                        const unsigned rel = hi[sl] - a;
                                        if (rel <= range) __hip_atomic_fetch_add(hist + (rel >> shift), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    }
                                wave_lds_fence();
                                const uint4 t4 = hz[0];
                                const int hv[BPL] = { (int)t4.x, (int)t4.y, (int)t4.z, (int)t4.w };
                                const int lsum = hv[0] + hv[1] + hv[2] + hv[3];
                                const int incl = wave_incl_scan(lsum), excl = incl - lsum;
                                const bool mine = need_k > excl && need_k <= incl;
                                int B = -1, cb = 0, hb = 0;
                                if (mine) {
                                    int cum = excl;
#pragma unroll
                                    for (int w = 0; w < BPL; ++w) {
                                        if (B < 0 && need_k <= cum + hv[w]) { B = lane * BPL + w; cb = cum; hb = hv[w]; }
                                        cum += hv[w];
                                    }
                                }
                                const unsigned long long bm = __builtin_amdgcn_ballot_w64(mine);
                                const int owner = bm ? __builtin_ctzll(bm) : 0;
                                B = __shfl(B, owner, 64); cb = __shfl(cb, owner, 64); hb = __shfl(hb, owner, 64);
                                // (the window's end may cut the last bucket -- the cover's high word at the first level: nothing beyond it was counted)
                                const unsigned wend = a + range, bstart = a + ((unsigned)(B < 0 ? 0 : B) << shift), bspan = (1u << shift) - 1u;
                                const unsigned bend = wend - bstart < bspan ? wend : bstart + bspan;
                                if (B < 0) { ok = false; run = false; }                       // cannot happen (cnt >= kk)
                                else if (need_k - cb == hb) { a = bend; exact = true; run = false; }     // the k-th is the last of its bucket
                                else if (shift == 0) { a = bstart; run = false; }                        // one value wide: THE k-th high word
                                else { need_k -= cb; a = bstart; range = bend - bstart; }
                            }
                        }
                        const unsigned P = a;
                        unsigned Q = 0xFFFFFFFFu;
                        int n_le = kk;
                        if (ok && !exact) {
                            int c_less = 0, c_eq = 0;
#pragma unroll
                            for (int sl = 0; sl < S; ++sl) if (sl * 64 < m) { c_less += hi[sl] < P ? 1 : 0; c_eq += hi[sl] == P ? 1 : 0; }
                            c_less = group_sum_i<64>(c_less); c_eq = group_sum_i<64>(c_eq);
                            n_le = c_less + c_eq;
                            // several candidates share the k-th high word and not all of them fit: the low words decide
                            const int r2 = kk - c_less;
                            if (n_le > kk) {
                                unsigned lmin = 0xFFFFFFFFu, lmax = 0u;
#pragma unroll
                                for (int sl = 0; sl < S; ++sl) if (hi[sl] == P) { lmin = lo[sl] < lmin ? lo[sl] : lmin; lmax = lo[sl] > lmax ? lo[sl] : lmax; }
                                unsigned a2 = group_min_u<64>(lmin), b2 = group_max_u<64>(lmax);
                                while (a2 < b2) {
                                    const unsigned mid = a2 + ((b2 - a2) >> 1);
                                    int c2 = 0;
#pragma unroll
                                    for (int sl = 0; sl < S; ++sl) c2 += (hi[sl] == P && lo[sl] <= mid) ? 1 : 0;
                                    c2 = group_sum_i<64>(c2);
                                    if (c2 >= r2) b2 = mid; else a2 = mid + 1;
                                }
                                int c3 = 0;
#pragma unroll
                                for (int sl = 0; sl < S; ++sl) c3 += (hi[sl] == P && lo[sl] <= a2) ? 1 : 0;
                                c3 = group_sum_i<64>(c3);
                                Q = a2; n_le = c_less + c3;
                            }
                        }
                        const bool tied_out = ok && n_le > kk + kCellTieRoom;     // ties far beyond k (duplicates, lattices): the wave-per-query passes
                        if (tied_out) ok = false;
                        if (ok) {
                            // the selected patterns (all <= (P, Q)), lane by lane -> selection buffer -> sum of square roots
                            // in the block's own order (candidate 0, 1, 2, ...): a ballot per row of 64 candidates places them
                            int wbase = 0;
#pragma unroll
                            for (int sl = 0; sl < S; ++sl)
                                if (sl * 64 < m) {
                                    const bool sel = hi[sl] < P || (hi[sl] == P && lo[sl] <= Q);
                                    const unsigned long long bm = __builtin_amdgcn_ballot_w64(sel);
                                    if (sel)
                                        selbuf[wbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u))] =
                                            __longlong_as_double((long long)(((unsigned long long)hi[sl] << 32) | lo[sl]));
                                    wbase += __builtin_popcountll(bm);
                                }
                            wave_lds_fence();
                            double acc = 0.0;
                            for (int t = lane; t < n_le; t += 64) acc += sqrt(selbuf[t]);
#pragma unroll
                            for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
                            if (lane == 0) {
                                const double top = __longlong_as_double((long long)(((unsigned long long)P << 32) | Q));
                                const double total = n_le > kk ? acc - (double)(n_le - kk) * sqrt(top) : acc;
                                avg[qsidx[qi]] = kk > 0 ? total / (double)kk : -1.0;
                            }
                            wave_lds_fence();
                        }
                        if (lane == 0 && (ok || tied_out)) qstate[qi] = ok ? 1 : 2;
                    }
                }
                __syncthreads();                                 // the states are written and the staged block may be replaced
            }
        }
        // what neither block settled: ONE counter update per chunk
        if (wave == 0) {
            const bool fb = lane < nqc && qstate[lane] != 1;
            const unsigned long long fm = __builtin_amdgcn_ballot_w64(fb);
            if (fm != 0ull) {
                int fb_base = 0;
                if (lane == 0) fb_base = atomicAdd(fb_count, __builtin_popcountll(fm));
                fb_base = __shfl(fb_base, 0, 64);
                if (fb) fb_list[fb_base + __builtin_popcountll(fm & ((1ull << lane) - 1ull))] = (int32_t)(base + lane);
            }
        }
        __syncthreads();
    }
}

// Exact ring walk, one thread per query (queries in cell order: neighbouring threads walk neighbouring cells).
// list != NULL: only the queries list[0 .. *list_count) (the wave kernel's overflow list).
__global__ void sor_knn_kernel(const GridParams *__restrict__ gp, const uint32_t *__restrict__ cell_start,
                               const float *__restrict__ spts, const int32_t *__restrict__ sidx, int64_t q0, int64_t q1, int k,
                               double *__restrict__ avg, const int32_t *__restrict__ list, const int32_t *__restrict__ list_count,
                               double *__restrict__ gheap)
{
    // gheap != NULL (k beyond what LDS holds): the thread's heap lives in the workspace, strided by the grid's thread count
    extern __shared__ __align__(16) double lds[];
    const GridParams g = *gp;
    const int64_t total = list ? (int64_t)*list_count : q1 - q0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = list ? (int64_t)list[t] : q0 + t;
        HeapD heap{ gheap ? gheap + (size_t)blockIdx.x * blockDim.x + threadIdx.x : lds + threadIdx.x, gheap ? (int)(gridDim.x * blockDim.x) : (int)blockDim.x, k, 0 };
        grid_knn_scan(g, cell_start, spts, (const int32_t *)nullptr, (double)spts[3 * s], (double)spts[3 * s + 1],
                      (double)spts[3 * s + 2], -1.0, heap);
        double sum = 0.0;
        for (int e = 0; e < heap.sz; ++e) sum += sqrt(heap.h[e * heap.stride]);
        avg[sidx ? (int64_t)sidx[s] : s - q0] = heap.sz > 0 ? sum / (double)heap.sz : -1.0;
    }
}

// mean / std exactly as [O3D]: mean = sum(avg>0)/n ; std = sqrt(sum_{avg>0}(avg-mean)^2/(n-1))
__global__ __launch_bounds__(256) void sor_sum_kernel(const double *__restrict__ avg, int64_t n, const double *__restrict__ stats,
                                                      int pass, double *__restrict__ part)
{
    __shared__ double sh[4];
    double mean = pass ? stats[0] : 0.0, acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v = avg[i];
        if (v > 0.0) acc += pass ? (v - mean) * (v - mean) : v;
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ void sor_final_kernel(const double *__restrict__ part, int nb, int64_t n, double std_ratio, int pass, double *stats)
{
    if (threadIdx.x || blockIdx.x) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += part[b];
    if (pass == 0) stats[0] = s / (double)n;
    else { stats[1] = sqrt(s / (double)(n - 1)); stats[2] = stats[0] + std_ratio * stats[1]; }
}
struct SorPred {
    const double *avg; const double *stats;
    __device__ bool operator()(int64_t i, int) const { double v = avg[i]; return v > 0.0 && v < stats[2]; }
};
struct IdxEmit {
    int32_t *idx;
    __device__ void operator()(int64_t i, int, int32_t dst) const { idx[dst] = (int32_t)i; }
};
// keep list AND the kept rows of up to two (n,3) attribute arrays: `cl, ind = remove_statistical_outlier(...)` without a host round trip
struct SorGather {
    const float *a0, *a1;
    float *o0, *o1;
};
// the kept rows, gathered by a launch of its own whose length is read on the DEVICE (the keep list's count): inside the compaction's
// emit the eight items of a thread gathered one after the other (17 us for 23k points against 6 + 4 for compaction + this kernel)
// d_count: in device memory (every thread reads it); count_out: the caller's count word, which may be host-visible pinned memory
__global__ __launch_bounds__(256) void sor_gather_kernel(SorGather g, const int32_t *__restrict__ idx, const int32_t *__restrict__ d_count,
                                                         int32_t *__restrict__ count_out)
{
    const int64_t n = *d_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) *count_out = (int32_t)n;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = idx[k];
        if (g.a0) { g.o0[3 * k] = g.a0[3 * s]; g.o0[3 * k + 1] = g.a0[3 * s + 1]; g.o0[3 * k + 2] = g.a0[3 * s + 2]; }
        if (g.a1) { g.o1[3 * k] = g.a1[3 * s]; g.o1[3 * k + 1] = g.a1[3 * s + 1]; g.o1[3 * k + 2] = g.a1[3 * s + 2]; }
    }
}

static int sor_block_threads(int k)
{
    // k * threads * 8 B of LDS per block; keep >= 2 blocks per CU where k allows
    if (k <= 32) return 256;
    if (k <= 80) return 128;
    return 64;
}

// The four launches below (partial sums, fold, partial squared deviations, fold) as ONE block for clouds it can walk in a few
// microseconds: the same block partition (thread t of virtual block b adds items b*2048 + t, +256, ... in the same order, the
// virtual blocks' sums are folded in the same order), so the statistics are bit-identical to the multi-launch form.
__global__ __launch_bounds__(1024) void sor_stats_small_kernel(const double *__restrict__ avg, int64_t n, int nb, double std_ratio, double *__restrict__ stats)
{
    __shared__ double part[64];               // nb <= 64 virtual blocks of 256 threads x 8 items
    __shared__ double sh[4][4];
    __shared__ double s_mean;
    const int vb = threadIdx.x >> 8, t = threadIdx.x & 255;          // 4 virtual blocks at a time
    for (int pass = 0; pass < 2; ++pass) {
        const double mean = pass ? s_mean : 0.0;
        for (int b0 = 0; b0 < nb; b0 += 4) {
            const int b = b0 + vb;
            double acc = 0.0;
            if (b < nb)
                for (int64_t i0 = (int64_t)b * 256 + t; i0 < n; i0 += 8 * (int64_t)nb * 256) {      // 8 loads in flight, added in the same order
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int64_t i = i0 + (int64_t)u * nb * 256;
                        v[u] = i < n ? avg[i] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (v[u] > 0.0) acc += pass ? (v[u] - mean) * (v[u] - mean) : v[u];
                }
            // block_sum of a 256-thread block: wave sums, then the four waves in order
            acc = wave_sum(acc);
            if ((t & 63) == 0) sh[vb][t >> 6] = acc;
            __syncthreads();
            if (t == 0 && b < nb) part[b] = ((sh[vb][0] + sh[vb][1]) + sh[vb][2]) + sh[vb][3];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int b = 0; b < nb; ++b) s += part[b];
            if (pass == 0) { stats[0] = s / (double)n; s_mean = stats[0]; }
            else { stats[1] = sqrt(s / (double)(n - 1)); stats[2] = stats[0] + std_ratio * stats[1]; }
        }
        __syncthreads();
    }
}

// statistics over avg (caller's point order) + ascending keep list -- shared by kpx_sor and kpx_sor_finish, so that the
// sharded filter folds the very same reduction tree over the very same array as the one-GPU call
static int sor_stats_compact(const double *avg, int64_t n, double std_ratio, double *part, int32_t *counts, int32_t *keep_idx,
                             int32_t *d_count, double *d_stats, hipStream_t st, const SorGather *ga = nullptr, bool state_is_clear = false)
{
    int nb = (int)(cdiv(n, 256 * 8) < 1 ? 1 : (cdiv(n, 256 * 8) > 1024 ? 1024 : cdiv(n, 256 * 8)));
    if (nb <= 64) {
        hipLaunchKernelGGL(sor_stats_small_kernel, dim3(1), dim3(1024), 0, st, avg, n, nb, std_ratio, d_stats);
    } else {
        for (int pass = 0; pass < 2; ++pass) {
            hipLaunchKernelGGL(sor_sum_kernel, dim3(nb), dim3(256), 0, st, avg, n, d_stats, pass, part);
            hipLaunchKernelGGL(sor_final_kernel, dim3(1), dim3(1), 0, st, part, nb, n, std_ratio, pass, d_stats);
        }
    }
    KPX_LAUNCH_CHECK();
    int32_t *dev_count = reinterpret_cast<int32_t *>(part + 1023);              // the gather reads the count on the device
    const int rc = compact(SorPred{ avg, d_stats }, IdxEmit{ keep_idx }, n, 1, counts, ga ? dev_count : d_count, st, state_is_clear);
    if (rc || !ga) return rc;
    const int64_t gb = cdiv(n, 256);
    hipLaunchKernelGGL(sor_gather_kernel, dim3((unsigned)(gb < 1 ? 1 : (gb > 4096 ? 4096 : gb))), dim3(256), 0, st, *ga, keep_idx, dev_count, d_count);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

__global__ __launch_bounds__(256) void sor_unsort_kernel(const double *__restrict__ avg_sorted, const int32_t *__restrict__ order, int64_t n,
                                                         double *__restrict__ avg)
{
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) avg[order[s]] = avg_sorted[s];
}

// q0 <= q1: only the queries at cell-sorted positions [q0, q1) are searched (kpx_sor_partial: avg_out is indexed by sorted
// position - q0 and the statistics / keep list are skipped); full == true: the whole filter (kpx_sor)
static int sor_impl(const float *pts, int64_t n, int k, double std_ratio, int32_t *keep_idx, int32_t *d_count, double *d_stats,
                    double *d_avg, Arena &a, hipStream_t st, bool full = true, int64_t q0 = 0, int64_t q1 = 0, int32_t *d_order = nullptr,
                    const SorGather *ga = nullptr)
{
    Grid g;
    if (full) { q0 = 0; q1 = n; }
    int kk = (int64_t)k < n ? k : (int)(n > 0 ? n : 1);
    // cell occupancy (as seen by a point) ~0.4 k: the 27-cell block then holds ~10 k candidates and usually covers the
    // k-th neighbour
    // (round 4: with the cell's queries answered together -- sor_cell_kernel -- a query that the 27-cell block does not cover costs a second,
    // wave-per-query search, so the cells are sized to cover the k-th neighbour of nearly every query: r_k ~ sqrt(k / (pi rho)) against
    // h = sqrt(occ / rho) on a surface of density rho)
    static const double occ_factor = [] { const char *e = getenv("KPX_SOR_OCC"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 0.4; }();
    double occ = occ_factor * (double)kk;
    occ = occ < 6.0 ? 6.0 : (occ > 240.0 ? 240.0 : occ);
    int rc = grid_build(pts, n, occ, a, &g, st);
    if (rc) return rc;
    double *avg = a.get<double>((size_t)(n > 0 ? n : 1));
    double *part = a.get<double>(1024);
    int32_t *counts = a.get<int32_t>((size_t)compact_ws_ints(n));
    int32_t *fb_list = a.get<int32_t>((size_t)(n > 0 ? n : 1) + 1);
    int32_t *fb_list2 = a.get<int32_t>((size_t)(n > 0 ? n : 1) + 1);
    int32_t *fb_list0 = a.get<int32_t>((size_t)(n > 0 ? n : 1) + 1);
    // k beyond the LDS heaps (KPX_SOR_LDS_K): the last pass's per-thread heaps live here (fewer blocks as k grows: <= 256 MB)
    const bool global_heap = kk > KPX_SOR_LDS_K;
    int heap_blocks = 256;
    while (global_heap && heap_blocks > 8 && (size_t)heap_blocks * 64 * (size_t)kk * sizeof(double) > ((size_t)256 << 20)) heap_blocks >>= 1;
    double *gheap = global_heap ? a.get<double>((size_t)heap_blocks * 64 * (size_t)kk) : nullptr;
    if (a.dry) return KPX_OK;
    KPX_ARENA_CHECK(a);
    if (d_avg) avg = d_avg;
    const int32_t *out_idx = full ? g.sorted_idx : nullptr;
    const int64_t nq = q1 - q0;
    int32_t *fb_count = g.spare, *fb_count2 = g.spare + 1, *fb_count0 = g.spare + 2;       // cleared by the grid build
    const int threads = sor_block_threads(kk);
    const size_t lds = global_heap ? 0 : (size_t)kk * threads * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        KPX_HIP(hipFuncSetAttribute((const void *)sor_knn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        KPX_HIP(hipFuncSetAttribute((const void *)sor_wave_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        KPX_HIP(hipFuncSetAttribute((const void *)sor_wave_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    {
        ProfScope prof(KPX_PROF_SOR_KNN, 12.0 * (double)n + 8.0 * (double)n, st);     // read points, write mean distances
        const int32_t *none = nullptr;
        // pass 0: the queries of a cell together (sor_cell_kernel): L lanes per query, L x S candidates at most; what it cannot settle
        // goes to the wave-per-query passes below.  KPX_SOR_CELL=0: every query starts at pass 1.
        // Measured (profiles/r04/sor_cell_kernel.txt): the cell kernel wins where a query's candidates are many -- k = 200 on a 221k-point
        // fused cloud 4.39 -> 3.63 ms -- and roughly breaks even below (k = 20 at 259k points 1.04 -> 0.94 ms, k = 50 1.72 -> 1.92, a
        // frame-sized 26k-point cloud 0.17 -> 0.23): there a third of the filter's time is the 5 % of the queries that sit in sparse
        // space (their k-th neighbour lies many cells away: the wave-per-query passes' re-gathers), and the selection itself is
        // instruction-bound either way.  So: KPX_SOR_CELL unset = the cell kernel from k > 64 on, 1 = always, 0 = never.
        static const int cell_mode = [] { const char *e = getenv("KPX_SOR_CELL"); return e ? (e[0] == '0' ? 0 : 2) : 1; }();
        // (round 5: the block-per-64-queries kernel from k > 32 on; at k <= 32 the wave-per-query pass -- now selecting by counting -- is as fast)
        const bool cell_on = (cell_mode == 2 || (cell_mode == 1 && kk > 32)) && kk <= 1024;       // (its largest block holds 2048 candidates)
        const int32_t *list0 = nullptr, *count0 = nullptr;
        if (cell_on && nq > 0) {
            const int kbuf = kk + kCellTieRoom;
            const int64_t chunks = cdiv(nq, kCellQueries);
            auto bytes = [&](int cap, int groups, int waves) { return (size_t)waves * groups * sor_cell_group_doubles(kbuf, cap) * 8; };
#define KPX_SOR_CELL_LAUNCH(LL, SS, WW)                                                                                          \
            hipLaunchKernelGGL((sor_cell_kernel<LL, SS, WW>), dim3((unsigned)(cdiv(chunks, WW) > 32768 ? 32768 : cdiv(chunks, WW))), dim3(64 * WW), \
                               bytes(LL * SS, 64 / LL, WW), st, g.params, g.cell_start, g.sorted_pts, out_idx, q0, q1, kk, kbuf, avg, fb_list0, fb_count0)
            static const int small_cap = [] { const char *e = getenv("KPX_SOR_CELL_SMALL"); return e ? atoi(e) : 1; }();
            static const bool wave_cells = [] { const char *e = getenv("KPX_SOR_BLOCK"); return e && e[0] == '0'; }();
            static const int block_mink = [] { const char *e = getenv("KPX_SOR_BLOCK_MINK"); return e ? atoi(e) : 33; }();
            // a block per 64 queries, its four waves sharing the staged cell blocks (round 5), from k = block_mink on (KPX_SOR_BLOCK_MINK;
            // KPX_SOR_BLOCK=0: never); below it the wave-per-16-queries forms
            const int64_t bchunks = cdiv(nq, kBlkQueries);
#define KPX_SOR_BLOCK_LAUNCH(SS)                                                                                                 \
            hipLaunchKernelGGL(sor_block_kernel<SS>, dim3((unsigned)(bchunks > 16384 ? 16384 : bchunks)), dim3(64 * kBlkWaves),      \
                               ((size_t)(64 * SS * 3) / 2 + (size_t)kBlkWaves * kCellBuckets / 2 + (size_t)kBlkWaves * kbuf) * 8, st, g.params, g.cell_start, \
                               g.sorted_pts, out_idx, q0, q1, kk, kbuf, avg, fb_list0, fb_count0)
            if (!wave_cells && kk >= block_mink) {
                if (kk <= 32) KPX_SOR_BLOCK_LAUNCH(4);
                else if (kk <= 64) KPX_SOR_BLOCK_LAUNCH(8);
                else if (kk <= 128) KPX_SOR_BLOCK_LAUNCH(16);
                else KPX_SOR_BLOCK_LAUNCH(32);
            }
            else if (kk <= 24 && small_cap) KPX_SOR_CELL_LAUNCH(16, 8, 4);
            else if (kk <= 32) KPX_SOR_CELL_LAUNCH(16, 16, 4);
            else if (kk <= 64) KPX_SOR_CELL_LAUNCH(32, 16, 4);
            else if (kk <= 128) KPX_SOR_CELL_LAUNCH(64, 16, 4);
            else KPX_SOR_CELL_LAUNCH(64, 32, 1);
#undef KPX_SOR_BLOCK_LAUNCH
#undef KPX_SOR_CELL_LAUNCH
            list0 = fb_list0; count0 = fb_count0;
        }
        // pass 1: one wave per query, 1024-candidate buffer (many waves per CU)
        // (round 5: k beyond half a buffer skips the pass whose buffer cannot hold k candidates plus their surroundings -- the queries
        // go on to the next one through the same lists)
        const int cap1 = kk <= 32 ? 512 : 1024;               // small k: smaller buffers, more waves per CU
        const bool pass1 = kk <= 512, pass2 = kk <= 4096;
        if (nq > 0 && pass1)
            hipLaunchKernelGGL(sor_wave_kernel<4>, dim3((unsigned)(cdiv(nq, 4) > (list0 ? 2048 : 8192) ? (list0 ? 2048 : 8192) : cdiv(nq, 4))), dim3(256), (size_t)4 * cap1 * 8, st,
                               g.params, g.cell_start, g.sorted_pts, out_idx, q0, q1, kk, cap1, avg, list0, count0, fb_list, fb_count);
        // pass 2: the queries whose block did not fit (isolated points next to a dense sheet), 8192-candidate buffer
        const int32_t *list1 = pass1 ? fb_list : list0, *count1 = pass1 ? fb_count : count0;      // what pass 2 searches: pass 1's leftovers, or its input
        if (nq > 0 && pass2)
            hipLaunchKernelGGL(sor_wave_kernel<1>, dim3(2048), dim3(64), (size_t)8192 * 8, st, g.params, g.cell_start, g.sorted_pts,
                               out_idx, q0, q1, kk, 8192, avg, list1, count1, fb_list2, fb_count2);
        // pass 3: whatever is left: thread-per-query ring walk with a k-heap (in LDS; in the workspace for k > KPX_SOR_LDS_K)
        const int32_t *list2 = pass2 ? fb_list2 : list1, *count2 = pass2 ? fb_count2 : count1;
        if (nq > 0)
            hipLaunchKernelGGL(sor_knn_kernel, dim3(global_heap ? heap_blocks : 256), dim3(global_heap ? 64 : threads), lds, st, g.params, g.cell_start, g.sorted_pts,
                               out_idx, q0, q1, kk, avg, list2, count2, gheap);
        (void)none;
    }
    if (!full) {
        if (d_order) KPX_HIP(hipMemcpyAsync(d_order, g.sorted_idx, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        KPX_LAUNCH_CHECK();
        return KPX_OK;
    }
    // the grid build cleared g.extra together with its counters: the compaction's tile states live there when they fit
    const bool fits = compact_ws_ints(n) <= kGridExtraWords;
    return sor_stats_compact(avg, n, std_ratio, part, fits ? g.extra : counts, keep_idx, d_count, d_stats, st, ga, fits);
}

// ---- estimate_normals ---------------------------------------------------------------------------------
__global__ void normals_kernel(const GridParams *__restrict__ gp, const uint32_t *__restrict__ cell_start,
                               const float *__restrict__ spts, const int32_t *__restrict__ sidx, const float *__restrict__ pts,
                               int64_t n, int k, double r2, float *__restrict__ normals, const int32_t *__restrict__ list,
                               const int32_t *__restrict__ list_count, double *__restrict__ gheap, int32_t *__restrict__ gix)
{
    // gheap / gix != NULL (max_nn beyond what LDS holds): the thread's (d^2, index) heap lives in the workspace
    extern __shared__ __align__(16) double lds[];
    const int64_t total = list ? (int64_t)*list_count : n;
    const GridParams g = *gp;
    int32_t *ilds = reinterpret_cast<int32_t *>(lds + (size_t)k * blockDim.x);
    const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int gstride = (int)(gridDim.x * blockDim.x);
    // one query per thread; list != NULL: only the queries list[0 .. *list_count) (the wave kernel's leftovers)
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = list ? (int64_t)list[t] : t;
        HeapDI heap{ gheap ? gheap + gt : lds + threadIdx.x, gheap ? gix + gt : ilds + threadIdx.x, gheap ? gstride : (int)blockDim.x, k, 0 };
        grid_knn_scan(g, cell_start, spts, sidx, (double)spts[3 * s], (double)spts[3 * s + 1], (double)spts[3 * s + 2], r2, heap);
        const int64_t me = sidx[s];
        double nx = 0.0, ny = 0.0, nz = 1.0;
        if (heap.sz >= 3) {
            double c[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            for (int e = 0; e < heap.sz; ++e) {
                int64_t j = heap.ix[e * heap.stride];
                double x = pts[3 * j], y = pts[3 * j + 1], z = pts[3 * j + 2];
                c[0] += x; c[1] += y; c[2] += z;
                c[3] += x * x; c[4] += x * y; c[5] += x * z; c[6] += y * y; c[7] += y * z; c[8] += z * z;
            }
            const double m = (double)heap.sz;
#pragma unroll
            for (int q = 0; q < 9; ++q) c[q] /= m;
            double cov[6] = { c[3] - c[0] * c[0], c[4] - c[0] * c[1], c[5] - c[0] * c[2],
                              c[6] - c[1] * c[1], c[7] - c[1] * c[2], c[8] - c[2] * c[2] };
            double w[3], V[9];
            sym3_eigen(cov, w, V);
            nx = V[0]; ny = V[3]; nz = V[6];                       // eigenvector of the smallest eigenvalue
            double nn = sqrt(nx * nx + ny * ny + nz * nz);
            if (nn > 0.0) { nx /= nn; ny /= nn; nz /= nn; } else { nx = 0.0; ny = 0.0; nz = 1.0; }
        }
        normals[3 * me] = (float)nx; normals[3 * me + 1] = (float)ny; normals[3 * me + 2] = (float)nz;
    }
}

// One wave per query: wave_knn_select with identities, then the covariance of the selected neighbours.  Ties at the
// k-th distance are resolved like the (d^2, index) heap of normals_kernel: the lowest original indices stay.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void normals_wave_kernel(const GridParams *__restrict__ gp, const uint32_t *__restrict__ cell_start,
                                                                  const float *__restrict__ spts, const int32_t *__restrict__ sidx,
                                                                  int64_t n, int k, int cap, double r2, double *__restrict__ covbuf,
                                                                  int32_t *__restrict__ fb_list, int32_t *__restrict__ fb_count)
{
    extern __shared__ __align__(16) double lds[];
    __shared__ uint32_t run_s0[WAVES][64];
    __shared__ int32_t run_off[WAVES][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t *posbase = reinterpret_cast<uint32_t *>(lds + (size_t)WAVES * cap);
    __shared__ __align__(16) uint32_t knn_hist[WAVES][kKnnBuckets];
    const WaveKnnScratch sc{ lds + (size_t)wave * cap, posbase + (size_t)wave * cap, run_s0[wave], run_off[wave], cap, knn_hist[wave] };
    const GridParams g = *gp;
    for (int64_t s = (int64_t)blockIdx.x * WAVES + wave; s < n; s += (int64_t)gridDim.x * WAVES) {
        const double q[3] = { (double)spts[3 * s], (double)spts[3 * s + 1], (double)spts[3 * s + 2] };
        WaveKnnResult res;
        if (!wave_knn_select<true>(g, cell_start, spts, q, k, r2, sc, res)) {
            if (lane == 0) { fb_list[atomicAdd(fb_count, 1)] = (int32_t)s; covbuf[10 * s + 9] = -1.0; }
            continue;
        }
        const int32_t idx_thr = wave_knn_tie_threshold(sc, res, sidx);
        double c[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int t = lane; t < res.m; t += 64) {
            if (wave_knn_is_selected(sc, res, sidx, idx_thr, t)) {
                const float *pp = spts + 3 * (int64_t)sc.pos[t];
                const double x = pp[0], y = pp[1], z = pp[2];
                c[0] += x; c[1] += y; c[2] += z;
                c[3] += x * x; c[4] += x * y; c[5] += x * z; c[6] += y * y; c[7] += y * z; c[8] += z * z;
            }
        }
#pragma unroll
        for (int a9 = 0; a9 < 9; ++a9)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c[a9] += __shfl_xor(c[a9], o, 64);
        // the 3x3 eigen-problem is serial work: it runs one thread per query in normals_eigen_kernel
        if (lane == 0) {
            double *o = covbuf + 10 * s;
#pragma unroll
            for (int a9 = 0; a9 < 9; ++a9) o[a9] = c[a9];
            o[9] = (double)res.kk;
        }
        wave_lds_fence();
    }
}
// sums -> covariance -> eigenvector of the smallest eigenvalue; one thread per (cell-sorted) query.  count < 0: the
// query was handed to the heap walk, which writes its normal itself.
__global__ __launch_bounds__(256) void normals_eigen_kernel(const double *__restrict__ covbuf, const int32_t *__restrict__ sidx, int64_t n,
                                                            float *__restrict__ normals)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const double *in = covbuf + 10 * s;
    const double m = in[9];
    if (m < 0.0) return;
    double nx = 0.0, ny = 0.0, nz = 1.0;
    if (m >= 3.0) {
        double c[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) c[q] = in[q] / m;
        double cov[6] = { c[3] - c[0] * c[0], c[4] - c[0] * c[1], c[5] - c[0] * c[2],
                          c[6] - c[1] * c[1], c[7] - c[1] * c[2], c[8] - c[2] * c[2] };
        double w[3], V[9];
        sym3_eigen(cov, w, V);
        nx = V[0]; ny = V[3]; nz = V[6];
        const double nn = sqrt(nx * nx + ny * ny + nz * nz);
        if (nn > 0.0) { nx /= nn; ny /= nn; nz /= nn; } else { nx = 0.0; ny = 0.0; nz = 1.0; }
    }
    const int64_t me = sidx[s];
    normals[3 * me] = (float)nx; normals[3 * me + 1] = (float)ny; normals[3 * me + 2] = (float)nz;
}

static int normals_impl(const float *pts, int64_t n, double radius, int max_nn, float *normals, Arena &a, hipStream_t st)
{
    Grid g;
    int kk = (int64_t)max_nn < n ? max_nn : (int)(n > 0 ? n : 1);
    int rc = grid_build(pts, n, 8.0, a, &g, st);
    if (rc) return rc;
    int32_t *fb_list = a.get<int32_t>((size_t)(n > 0 ? n : 1) + 1);
    double *covbuf = a.get<double>((size_t)(n > 0 ? n : 1) * 10);
    const bool global_heap = kk > KPX_NORMALS_LDS_NN;            // the fall-back heaps in the workspace (<= 256 MB of distances)
    int heap_blocks = 256;
    while (global_heap && heap_blocks > 8 && (size_t)heap_blocks * 64 * (size_t)kk * sizeof(double) > ((size_t)256 << 20)) heap_blocks >>= 1;
    double *gheap = global_heap ? a.get<double>((size_t)heap_blocks * 64 * (size_t)kk) : nullptr;
    int32_t *gix = global_heap ? a.get<int32_t>((size_t)heap_blocks * 64 * (size_t)kk) : nullptr;
    if (a.dry) return KPX_OK;
    KPX_ARENA_CHECK(a);
    int32_t *fb_count = g.spare;                                  // cleared by the grid build
    const int threads = global_heap ? 64 : (kk <= 48 ? 128 : 64);
    const size_t lds = global_heap ? 0 : (size_t)kk * threads * (sizeof(double) + sizeof(int32_t));
    static bool attr_set = false;
    if (!attr_set) {
        KPX_HIP(hipFuncSetAttribute((const void *)normals_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        KPX_HIP(hipFuncSetAttribute((const void *)normals_wave_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    // pass 1: one wave per query, 512- or 1024-candidate buffer; pass 2: the few queries that did not fit, thread-per-query heap walk
    // (max_nn > 512: the wave kernel's 1024-candidate buffer cannot hold a neighbourhood and its surroundings -- every query takes the heap walk)
    const int cap = kk <= 48 ? 512 : 1024;
    const bool wave_pass = kk <= 512;
    if (wave_pass) {
        hipLaunchKernelGGL(normals_wave_kernel<4>, dim3((unsigned)(cdiv(n, 4) > 8192 ? 8192 : cdiv(n, 4))), dim3(256),
                           (size_t)4 * cap * (sizeof(double) + sizeof(uint32_t)), st, g.params, g.cell_start, g.sorted_pts, g.sorted_idx, n, kk,
                           cap, radius * radius, covbuf, fb_list, fb_count);
        hipLaunchKernelGGL(normals_eigen_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, covbuf, g.sorted_idx, n, normals);
    }
    hipLaunchKernelGGL(normals_kernel, dim3(global_heap ? heap_blocks : 256), dim3(threads), lds, st, g.params, g.cell_start, g.sorted_pts, g.sorted_idx, pts, n, kk,
                       radius * radius, normals, wave_pass ? fb_list : (const int32_t *)nullptr, wave_pass ? fb_count : (const int32_t *)nullptr, gheap, gix);
    KPX_LAUNCH_CHECK();
    return KPX_OK;
}

}  // namespace kpx

using namespace kpx;

#ifdef KPX_SOR_CELL_STATS
KPX_EXPORT int kpx_debug_sor_stats(uint64_t *h_out16)
{
    unsigned long long *p = nullptr;
    static const unsigned long long zero[16] = { 0 };
    KPX_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(g_sor_stats)));
    KPX_HIP(hipMemcpy(h_out16, p, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    KPX_HIP(hipMemcpy(p, zero, sizeof(zero), hipMemcpyHostToDevice));
    return KPX_OK;
}
#endif
KPX_EXPORT size_t kpx_sor_workspace_bytes(int64_t n, int32_t nb_neighbors)
{
    Arena a(nullptr, 0);
    sor_impl(nullptr, n, nb_neighbors < 1 ? 1 : nb_neighbors, 1.0, nullptr, nullptr, nullptr, nullptr, a, nullptr);
    return a.off;
}
KPX_EXPORT int kpx_sor(const float *pts, int64_t n, int32_t nb_neighbors, double std_ratio, int32_t *keep_idx,
                       int32_t *d_count, double *d_stats, double *d_avg, void *ws, size_t ws_bytes, void *stream)
{
    // [O3D] "Illegal input parameters, the number of neighbors and standard deviation ratio must be positive"
    KPX_REQUIRE(nb_neighbors >= 1 && std_ratio > 0.0, "remove_statistical_outlier: nb_neighbors and std_ratio must be positive");
    KPX_REQUIRE(nb_neighbors <= KPX_SOR_MAX_K, "remove_statistical_outlier: nb_neighbors > %d is not supported", KPX_SOR_MAX_K);
    KPX_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "kpx_sor: bad size");
    KPX_REQUIRE(d_count && d_stats && ws, "kpx_sor: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { KPX_HIP(hipMemsetAsync(d_count, 0, sizeof(int32_t), st)); return KPX_OK; }
    KPX_REQUIRE(pts && keep_idx, "kpx_sor: null pointer");
    Arena a(ws, ws_bytes);
    return sor_impl(pts, n, nb_neighbors, std_ratio, keep_idx, d_count, d_stats, d_avg, a, st);
}

KPX_EXPORT int kpx_sor_select(const float *pts, const float *attr, int64_t n, int32_t nb_neighbors, double std_ratio, float *out_pts,
                              float *out_attr, int32_t *keep_idx, int32_t *d_count, double *d_stats, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(nb_neighbors >= 1 && std_ratio > 0.0, "remove_statistical_outlier: nb_neighbors and std_ratio must be positive");
    KPX_REQUIRE(nb_neighbors <= KPX_SOR_MAX_K, "remove_statistical_outlier: nb_neighbors > %d is not supported", KPX_SOR_MAX_K);
    KPX_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "kpx_sor_select: bad size");
    KPX_REQUIRE(d_count && d_stats && ws, "kpx_sor_select: null pointer");
    KPX_REQUIRE(!attr || out_attr, "kpx_sor_select: attributes without an output array");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { KPX_HIP(hipMemsetAsync(d_count, 0, sizeof(int32_t), st)); return KPX_OK; }
    KPX_REQUIRE(pts && keep_idx && out_pts, "kpx_sor_select: null pointer");
    Arena a(ws, ws_bytes);
    const SorGather ga{ pts, attr, out_pts, out_attr };
    return sor_impl(pts, n, nb_neighbors, std_ratio, keep_idx, d_count, d_stats, nullptr, a, st, true, 0, 0, nullptr, &ga);
}

KPX_EXPORT int kpx_sor_partial(const float *pts, int64_t n, int32_t nb_neighbors, int64_t q_begin, int64_t q_end, double *d_avg_sorted,
                               int32_t *d_order, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(nb_neighbors >= 1, "remove_statistical_outlier: nb_neighbors and std_ratio must be positive");
    KPX_REQUIRE(nb_neighbors <= KPX_SOR_MAX_K, "remove_statistical_outlier: nb_neighbors > %d is not supported", KPX_SOR_MAX_K);
    KPX_REQUIRE(n >= 1 && n < ((int64_t)1 << 31), "kpx_sor_partial: bad size");
    KPX_REQUIRE(q_begin >= 0 && q_begin <= q_end && q_end <= n, "kpx_sor_partial: bad query range");
    KPX_REQUIRE(pts && ws && (d_avg_sorted || q_begin == q_end), "kpx_sor_partial: null pointer");
    Arena a(ws, ws_bytes);
    return sor_impl(pts, n, nb_neighbors, 1.0, nullptr, nullptr, nullptr, d_avg_sorted, a, (hipStream_t)stream, false, q_begin, q_end, d_order);
}
KPX_EXPORT size_t kpx_sor_finish_workspace_bytes(int64_t n)
{
    Arena a(nullptr, 0);
    a.get<double>((size_t)(n > 0 ? n : 1));
    a.get<double>(1024);
    a.get<int32_t>((size_t)compact_ws_ints(n));
    return a.off;
}
KPX_EXPORT int kpx_sor_finish(const double *d_avg_sorted, const int32_t *d_order, int64_t n, double std_ratio, int32_t *keep_idx,
                              int32_t *d_count, double *d_stats, double *d_avg, void *ws, size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(std_ratio > 0.0, "remove_statistical_outlier: nb_neighbors and std_ratio must be positive");
    KPX_REQUIRE(n >= 1 && n < ((int64_t)1 << 31), "kpx_sor_finish: bad size");
    KPX_REQUIRE(d_avg_sorted && d_order && keep_idx && d_count && d_stats && ws, "kpx_sor_finish: null pointer");
    hipStream_t st = (hipStream_t)stream;
    Arena a(ws, ws_bytes);
    double *avg = a.get<double>((size_t)n);
    double *part = a.get<double>(1024);
    int32_t *counts = a.get<int32_t>((size_t)compact_ws_ints(n));
    KPX_ARENA_CHECK(a);
    if (d_avg) avg = d_avg;
    hipLaunchKernelGGL(sor_unsort_kernel, dim3((unsigned)(cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256))), dim3(256), 0, st, d_avg_sorted, d_order, n, avg);
    return sor_stats_compact(avg, n, std_ratio, part, counts, keep_idx, d_count, d_stats, st);
}

KPX_EXPORT size_t kpx_normals_workspace_bytes(int64_t n, int32_t max_nn)
{
    Arena a(nullptr, 0);
    normals_impl(nullptr, n, 1.0, max_nn < 1 ? 1 : max_nn, nullptr, a, nullptr);
    return a.off;
}
KPX_EXPORT int kpx_estimate_normals(const float *pts, int64_t n, double radius, int32_t max_nn, float *normals, void *ws,
                                    size_t ws_bytes, void *stream)
{
    KPX_REQUIRE(radius > 0.0 && max_nn >= 1, "estimate_normals: radius and max_nn must be positive");
    KPX_REQUIRE(max_nn <= KPX_NORMALS_MAX_NN, "estimate_normals: max_nn > %d is not supported", KPX_NORMALS_MAX_NN);
    KPX_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "kpx_estimate_normals: bad size");
    if (n == 0) return KPX_OK;
    KPX_REQUIRE(pts && normals && ws, "kpx_estimate_normals: null pointer");
    Arena a(ws, ws_bytes);
    return normals_impl(pts, n, radius, max_nn, normals, a, (hipStream_t)stream);
}
