// kpx_icprows.h -- one ICP iteration of SEVERAL registrations, one wave per 64 sorted rows (included by kpx_icp.hip only; round 5).
//
// icp_iter_body gives every 16-row tile a wave and every 64 rows a block of four: right for the first iterations, where every tile
// is searched, and wasteful afterwards -- from iteration ~10 on 95-99.9 % of the rows carry a certificate ("Certificates", kpx_icp.hip),
// a block then lives ~8.6 us of which the search is 0.3, and only 16 of a wave's 64 lanes do the per-row work (transform, bound,
// certificate test, the 44 products of the update sums).  With four frames in flight the chip's wave slots x a block's life time IS the
// frame rate, so here
//   * a block is ONE wave and owns 64 consecutive sorted rows, one row per lane in the prologue (rows, previous partners and their
//     coordinates / normals, certificates: coalesced loads; AC1, the partner's AC2 value, the certificate test) and in the pair epilogue
//     (AC3, the 17 / 44 contributions);
//   * the wave sweeps only those of its four 16-row tiles that hold an uncertified row, one after the other, through the same
//     sweep_wave (kpx_nnlocal.h) on the tile's row records in LDS -- partners are those of icp_iter_body bit for bit;
//   * the sums follow the contract stated in icp_iter_body: per tile the balanced tree of four DPP butterfly steps
//     (row16_tree_sum == tile_tree16), the four tile partials added exactly in fixed point, one pair of returning atomics per sum;
//   * the block that draws the registration's last ticket performs the update (icp_finish_wave), as in icp_iter_body's ticket mode;
//   * every problem of the launch carries its OWN target operands and its OWN iteration number: the registrations of several frames in
//     flight share one launch per tick (kpx_stream, kpx_frame.hip), each at the iteration it has reached.
#pragma once

namespace kpx {

constexpr int kRowsBatchMax = 16;                  // registrations per launch (kernel arguments: 16 x 176 B)
constexpr int kRowsBlock = 64;
struct RowsProblem {
    // source side (sorted-row order; written by icp_batch_init_kernel and by the iterations themselves)
    const float *src_sorted;
    int32_t *idx_sorted;
    float *ptgt_sorted;
    uint32_t *cert;
    double *thist;
    double *light_key;
    const double *sbbox;
    IcpState *state;
    unsigned long long *ring;
    double *result;
    unsigned long long *progress;
    unsigned long long tag;
    // target side (nn_local_prep_kernel)
    const float *tgt, *tn;
    const double *Bs;
    const int32_t *orig;
    const float *tile_box, *group_box;
    const double *tbbox;
    int64_t n;
    int32_t n_groups, k;
    uint32_t block0, blocks;
};
struct RowsArgs {
    RowsProblem p[kRowsBatchMax];
    int32_t count;
};

constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;
// sum over the 16 lanes of a DPP row, the same value (bit for bit: addition commutes) in all of them: tile_tree16's order
__device__ __forceinline__ double row16_tree_sum(double v)
{
    v += dpp_f64<kDppXor1>(v);
    v += dpp_f64<kDppXor2>(v);
    v += dpp_f64<kDppHalfMirror>(v);
    v += dpp_f64<kDppMirror>(v);
    return v;
}

template <int MODE, int R>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KPX_ICP_WPE, KPX_ICP_WPE))) void icp_rows_kernel(RowsArgs args, double max_d2, int max_iter, double rel_fit,
                                                                                                                     double rel_rmse, unsigned long long *__restrict__ tile_visits,
                                                                                                                     int light, CertPolicy pol)
{
    constexpr int NACC = MODE == 1 ? kAcc : 17;
    static_assert(R == 16 || R == 32 || R == 64, "rows per wave");
    int pi = 0;
#pragma unroll
    for (int c = 1; c < kRowsBatchMax; ++c) pi += (c < args.count && blockIdx.x >= args.p[c].block0) ? 1 : 0;
    const RowsProblem &P = args.p[pi];
    const unsigned bid = blockIdx.x - P.block0;
    const int k = P.k;
    IcpState *const st = P.state;
    if (st->done) return;

    // LDS: 10.3 KB per wave (12 waves per CU).  The sweep's records are COMPACT: slot = the row's rank among the block's rows that are
    // searched; what the pair epilogue needs of every row is parked by lane.
    __shared__ IcpState s_state;
    __shared__ double s_sums[kAcc];
    __shared__ FinishScratch s_tail;
    __shared__ double rowd[kRowsBlock][kRowStride];     // by slot: x, y, z under this iteration's transform, (the sweep's bound), K, bound / result value
    __shared__ float rowf[kLRows][kRowFStride];         // float32 mirror of the 16 rows being swept (written by the sweep)
    __shared__ float rowk[kRowsBlock][11];              // by lane: previous partner's coordinates, normal, index; the row's own coordinates
    __shared__ int32_t rowi[kRowsBlock];                // by slot: partner (bound going in, result coming out)
    __shared__ uint32_t rowc[kRowsBlock];               // by slot: certificate word (the sweep writes the new one)
    __shared__ int32_t rowm[kRowsBlock];                // by slot: 1 = the row is certified (searched only by the self-check)
    __shared__ int32_t lists[kLScratch];                // the sweep's scratch; afterwards the four tile partials of the sums
    static_assert(kLScratch * sizeof(int32_t) >= 4 * kAcc * sizeof(double), "the tile partials reuse the sweep's scratch");
    double (*s_part)[kAcc] = reinterpret_cast<double (*)[kAcc]>(lists);

    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int64_t last = P.n - 1;
    const int64_t row = (int64_t)bid * R + lane;
    const bool valid = lane < R && row <= last;
    const int64_t r = valid ? row : last;
    const bool certs = (light & 2) != 0, use_light = (light & 1) != 0, cert_check = (light & 4) != 0;
    const double t2max = target_t2max(P.tbbox);

    // everything that does not depend on this iteration's transform is requested first
    float my_src[3], my_pt[3] = { 0.0f, 0.0f, 0.0f }, my_nrm[3] = { 0.0f, 0.0f, 0.0f };
    int32_t my_prev = -1;
    uint32_t my_cert = 0u;
#pragma unroll
    for (int a = 0; a < 3; ++a) my_src[a] = P.src_sorted[3 * r + a];
    if (k > 0) {
        my_prev = P.idx_sorted[r];
#pragma unroll
        for (int a = 0; a < 3; ++a) my_pt[a] = P.ptgt_sorted[3 * r + a];
        if (certs) my_cert = P.cert[r];
    }
    const double key_in = use_light ? P.light_key[bid] : 0.0;
    if (lane < (int)(sizeof(IcpState) / sizeof(double))) reinterpret_cast<double *>(&s_state)[lane] = reinterpret_cast<const double *>(st)[lane];
    GroupPre gpre;
    group_pre_load(gpre, P.group_box, P.n_groups, lane);
    if (MODE == 1 && k > 0) {
        const float *np_ = P.tn + 3 * (int64_t)(my_prev > 0 ? my_prev : 0);
#pragma unroll
        for (int a = 0; a < 3; ++a) my_nrm[a] = np_[a];
    }
    // (certificates) the transform of the iteration the row was last searched in: its position then is recomputed from it, exactly
    double Th[12];
    const int kc = (int)(my_cert & 63u);
    const float Lc = __uint_as_float(my_cert & ~63u);
    if (certs && k > 0 && Lc > 0.0f) {
#pragma unroll
        for (int e = 0; e < 12; ++e) Th[e] = P.thist[12 * kc + e];
    } else {
#pragma unroll
        for (int e = 0; e < 12; ++e) Th[e] = 0.0;
    }
    const double *Tk = st->T;
    unsigned long long *const ticket = P.ring + kAccSet;

    // LightSkip: nothing of this block can have come within reach since it was last swept -> straight to the ticket
    const bool skip = use_light && key_in > 0.0 && (st->motion + st->reach) * (1.0 + 1e-6) + 1e-6 < key_in;
    unsigned visited_total = 0u;
    if (!skip) {
        const double c_reach = certs ? st->reach : 0.0;
        double c_skin = 0.0;
        if (certs && k > 0) {
            const double lm = st->last_motion, md = sqrt(max_d2);
            if (lm <= (double)pol.calm * md) c_skin = fmin(fmax((double)pol.factor * lm, (double)pol.smin * md), (double)pol.smax * md);
        }
        bool my_active = valid, my_certd = false;
        unsigned long long act64, certd64;
        int n_act;
        {
            double s[3];
            xform_row(Tk, my_src, s);
            const double seed = row_seed(s);
            double bv = INFINITY;
            int32_t bj = INT_MAX;
            if (k > 0 && my_prev >= 0) {
                const double tx = my_pt[0], ty = my_pt[1], tz = my_pt[2];
                const double t2 = fma(tx, tx, fma(ty, ty, tz * tz));
                double d = fma(s[0], -2.0 * tx, seed);
                d = fma(s[1], -2.0 * ty, d);
                d = fma(s[2], -2.0 * tz, d);
                bv = fma(1.0, t2, d);
                bj = my_prev;
            }
            const double clamp = (max_d2 + 1.0) * (1.0 + 9.31322574615478515625e-10) + ldexp(seed + t2max + 1.0, -38);
            if (!(bv <= clamp)) { bv = clamp; bj = INT_MAX; }
            double rb0 = bv - 1.0;
            if (certs) {
                double d1 = c_reach;
                if (bj != INT_MAX) {
                    const double dx = s[0] - (double)my_pt[0], dy = s[1] - (double)my_pt[1], dz = s[2] - (double)my_pt[2];
                    d1 = sqrt(fma(dz, dz, fma(dy, dy, dx * dx))) * (1.0 + 1e-12);
                }
                const bool keeps = (my_prev >= 0) == (bj != INT_MAX);
                double pc[3] = { 0.0, 0.0, 0.0 };
                if (Lc > 0.0f) {
                    const double x = my_src[0], y = my_src[1], z = my_src[2];
#pragma unroll
                    for (int a = 0; a < 3; ++a) pc[a] = fma(Th[4 * a], x, fma(Th[4 * a + 1], y, fma(Th[4 * a + 2], z, Th[4 * a + 3])));
                }
                const double ex = s[0] - pc[0], ey = s[1] - pc[1], ez = s[2] - pc[2];
                const double moved = sqrt(fma(ez, ez, fma(ey, ey, ex * ex))) * (1.0 + 1e-12);
                const bool certd = Lc > 0.0f && keeps && (d1 + moved) * (1.0 + 1e-6) + 1e-6 < (double)Lc;
                my_active = (!certd || cert_check) && valid;
                my_certd = certd && valid;
                if (certd && !cert_check) rb0 = -1.0;
                else if (c_skin > 0.0) { const double rr = d1 + c_skin; rb0 = fmax(rb0, rr * rr); }
            }
            act64 = __builtin_amdgcn_ballot_w64(my_active);
            certd64 = __builtin_amdgcn_ballot_w64(my_certd);
            n_act = __builtin_popcountll(act64);
            if (my_active) {
                const int slot = __builtin_popcountll(act64 & lt);
                rowd[slot][0] = s[0]; rowd[slot][1] = s[1]; rowd[slot][2] = s[2];
                rowd[slot][3] = rb0;
                rowd[slot][4] = seed; rowd[slot][5] = bv;
                rowi[slot] = bj;
                rowc[slot] = my_cert;
                rowm[slot] = my_certd ? 1 : 0;
            }
            if (lane >= n_act) {                               // padding of the last 16-row tile (and beyond): rows that take no part
                rowd[lane][0] = 0.0; rowd[lane][1] = 0.0; rowd[lane][2] = 0.0;
                rowd[lane][3] = -1.0;
                rowd[lane][4] = 1.0; rowd[lane][5] = 2.0;
                rowi[lane] = INT_MAX;
                rowc[lane] = 0u;
                rowm[lane] = 1;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) { rowk[lane][a] = my_pt[a]; rowk[lane][3 + a] = my_nrm[a]; rowk[lane][7 + a] = my_src[a]; }
            rowk[lane][6] = __int_as_float(my_prev);
        }
        wave_lds_fence();

        // the rows to be searched, 16 at a time (a block whose 64 rows all are: its four tiles as they stand)
        bool light_blk = n_act == R;                 // (LightSkip speaks for ALL rows of the block)
        double gap2_blk = INFINITY;
#pragma unroll 1
        for (int v0 = 0; v0 < n_act; v0 += kLRows) {
            const int left = n_act - v0;
            const unsigned act_mask = left >= kLRows ? 0xFFFFu : ((1u << left) - 1u);
            const int lane_s = opaque_i((int)threadIdx.x), q_s = lane_s >> 4, j_s = lane_s & 15;
            WaveRows w;
            w.rows = &rowd[v0][0];
            w.rowsf = &rowf[0][0];
            w.a = q_s < 3 ? rowd[v0 + j_s][q_s] : 1.0;
#pragma unroll
            for (int rr4 = 0; rr4 < 4; ++rr4) {
                const int rr = v0 + q_s + 4 * rr4;
                w.seed[rr4] = rowd[rr][4];
                w.best[rr4] = rowd[rr][5];
                w.bcol[rr4] = rowi[rr];
                w.rb0[rr4] = rowd[rr][3];
            }
            w.skin = c_skin;
            w.act_mask = act_mask;
            w.light_gap2 = -1.0;
            w.dbg = nullptr;
            wave_lds_fence();                                 // rowd[..][3] is the sweep's own slot from here on
            const unsigned long long swept = sweep_wave<true, true, true>(w, P.Bs, P.orig, P.tile_box, P.group_box, P.n_groups, t2max, lists, &gpre);
            visited_total += (unsigned)(swept & 0xFFFFu);
            const int lane_p = opaque_i((int)threadIdx.x), q_p = lane_p >> 4, j_p = lane_p & 15;
            if (certs && cert_check && j_p == 0) {
#pragma unroll
                for (int rr4 = 0; rr4 < 4; ++rr4) {
                    const int rr = q_p + 4 * rr4;
                    if (((act_mask >> rr) & 1u) != 0u && rowm[v0 + rr] != 0 && w.bcol[rr4] != rowi[v0 + rr]) {
                        if (atomicAdd(&g_cert_check[2], 1ull) == 0ull) {
                            g_cert_check[3] = (unsigned long long)k; g_cert_check[4] = (unsigned long long)((int64_t)bid * R + v0 + rr);
                            g_cert_check[5] = (unsigned long long)(unsigned)rowi[v0 + rr]; g_cert_check[6] = (unsigned long long)(unsigned)w.bcol[rr4];
                            g_cert_check[7] = (unsigned long long)(rowc[v0 + rr] & ~63u);
                        }
                    }
                }
            }
            wave_lds_fence();
            if (j_p == 0) {
#pragma unroll
                for (int rr4 = 0; rr4 < 4; ++rr4) {
                    const int rr = q_p + 4 * rr4;
                    rowi[v0 + rr] = w.bcol[rr4];
                    // new keys for the rows that were searched: L^2 = min(final culling bound, runner-up among the multiplied columns), both on d^2
                    if (certs && ((act_mask >> rr) & 1u) != 0u && rowm[v0 + rr] == 0) {
                        typedef unsigned uu2 __attribute__((ext_vector_type(2)));
                        const uu2 pat = { 0u, w.sec[rr4] };
                        const double d2nd = w.sec[rr4] == 0xFFFFFFFFu ? INFINITY : __builtin_bit_cast(double, pat) - 1.0 - w.eps_out;
                        const double l2 = fmin(w.rb_out[rr4], d2nd);
                        const uint32_t lb = l2 > 0.0 && k < kCertHist ? (__float_as_uint(f32_down(sqrt(l2) * (1.0 - 1e-7))) & ~63u) : 0u;
                        rowc[v0 + rr] = lb > 63u ? (lb | (uint32_t)k) : 0u;
                    }
                }
            }
            if (w.light_gap2 >= 0.0) gap2_blk = fmin(gap2_blk, w.light_gap2);
            else light_blk = false;
            wave_lds_fence();
        }
        const int lane_e = opaque_i((int)threadIdx.x);
        const unsigned long long lt_e = (1ull << lane_e) - 1ull;
        if (certs && cert_check && lane_e == 0) {
            if (bid == 0 && g_cert_check[2] == 0ull) {
                g_cert_check[3] = (unsigned long long)k; g_cert_check[4] = __builtin_bit_cast(unsigned long long, st->last_motion);
                g_cert_check[5] = __builtin_bit_cast(unsigned long long, st->motion); g_cert_check[6] = __builtin_bit_cast(unsigned long long, c_skin);
            }
            atomicAdd(&g_cert_check[0], (unsigned long long)__builtin_popcountll(certd64));
            atomicAdd(&g_cert_check[1], (unsigned long long)__builtin_popcountll(act64 & ~certd64));
        }
        if (tile_visits && lane_e == 0 && visited_total) atomicAdd(tile_visits + (blockIdx.x & (kVisitSlots - 1)), (unsigned long long)visited_total);

        // the chosen pairs: direct distance (AC3), contribution to the sums -- one row per lane
        double c[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a) c[a] = 0.0;
        const int64_t row_e = (int64_t)bid * R + lane_e;
        if (lane_e < R && row_e <= last) {
            const bool was_act = ((act64 >> lane_e) & 1ull) != 0ull;
            const int slot_e = __builtin_popcountll(act64 & lt_e);
            const int32_t prev_j = __float_as_int(rowk[lane_e][6]);
            // a row that was not searched keeps what it came with (certified: its partner, or none)
            const int32_t bj = was_act ? rowi[slot_e] : (prev_j >= 0 ? prev_j : INT_MAX);
            const bool none = bj < 0 || bj == INT_MAX;
            const int32_t out_j = none ? -1 : bj;
            const bool changed = k == 0 || out_j != prev_j;
            if (changed) P.idx_sorted[row_e] = out_j;
            const bool searched = was_act && ((certd64 >> lane_e) & 1ull) == 0ull;
            if (certs && searched) P.cert[row_e] = rowc[slot_e];
            if (!none) {
                const float sf[3] = { rowk[lane_e][7], rowk[lane_e][8], rowk[lane_e][9] };
                double s[3];
                xform_row(Tk, sf, s);
                float tf[3] = { rowk[lane_e][0], rowk[lane_e][1], rowk[lane_e][2] }, nf[3] = { rowk[lane_e][3], rowk[lane_e][4], rowk[lane_e][5] };
                if (changed) {
                    const float *tp = P.tgt + 3 * (int64_t)bj;
#pragma unroll
                    for (int e = 0; e < 3; ++e) tf[e] = tp[e];
                    if (MODE == 1) {
                        const float *np_ = P.tn + 3 * (int64_t)bj;
#pragma unroll
                        for (int e = 0; e < 3; ++e) nf[e] = np_[e];
                    }
#pragma unroll
                    for (int e = 0; e < 3; ++e) P.ptgt_sorted[3 * row_e + e] = tf[e];
                }
                const double t[3] = { (double)tf[0], (double)tf[1], (double)tf[2] };
                const double dx = s[0] - t[0], dy = s[1] - t[1], dz = s[2] - t[2];
                const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                if (d2 < max_d2) {
                    c[0] = 1.0; c[1] = d2;
#pragma unroll
                    for (int e = 0; e < 3; ++e) { c[2 + e] = s[e]; c[5 + e] = t[e]; }
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int e = 0; e < 3; ++e) c[8 + 3 * a + e] = t[a] * s[e];
                    if (MODE == 1) {
                        const double nx = nf[0], ny = nf[1], nz = nf[2];
                        const double res = (s[0] - t[0]) * nx + (s[1] - t[1]) * ny + (s[2] - t[2]) * nz;
                        const double J[6] = { s[1] * nz - s[2] * ny, s[2] * nx - s[0] * nz, s[0] * ny - s[1] * nx, nx, ny, nz };
                        int slot = 17;
#pragma unroll
                        for (int a = 0; a < 6; ++a)
#pragma unroll
                            for (int e = a; e < 6; ++e) c[slot++] = J[a] * J[e];
#pragma unroll
                        for (int a = 0; a < 6; ++a) c[38 + a] = J[a] * res;
                    }
                }
            }
        }
        // per tile the balanced tree (every lane of the tile's DPP row ends with the tile's partial), then the four partials exactly
#pragma unroll
        for (int a = 0; a < NACC; ++a) c[a] = row16_tree_sum(c[a]);
        wave_lds_fence();                                     // (the partials take the place of the sweep's scratch)
        if ((lane_e & 15) == 0) {
#pragma unroll
            for (int a = 0; a < NACC; ++a) s_part[lane_e >> 4][a] = c[a];
        }
        wave_lds_fence();
        if (lane_e < NACC) {
            unsigned long long lo = 0ull, hi = 0ull;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                unsigned long long l, h;
                fixed_split(s_part[t][lane_e], l, h);
                fixed_accumulate(lo, hi, l, h);
            }
            fixed_add_words_performed(P.ring + (((int64_t)(bid & (kAccCopies - 1)) * kAcc + lane_e) * kFixedWords), lo, hi);
        } else if (lane_e == NACC && (act64 & ~certd64) != 0ull) {
            // how much of the registration is still searched (see icp_iter_body): one returning add per block, eight words per registration
            const unsigned long long back = __hip_atomic_fetch_add(ticket + kSearchedWord + (bid & 7u), (unsigned long long)__builtin_popcountll(act64 & ~certd64), __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(back));
        }
        if (use_light && lane_e == 0) P.light_key[bid] = light_blk ? st->motion + sqrt(gap2_blk) * (1.0 - 1e-9) : 0.0;
    }

    // "The last block finishes the job" (icp_iter_body, ticket mode): every add above has RETURNED, the wave waits for all of them, then
    // draws its ticket; the block that draws the registration's last one reads the totals with device-coherent loads, clears them and
    // performs the update.  The state is written with plain stores: its readers are the blocks of the NEXT launch.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned tk = 0u;
    if (threadIdx.x == 0) tk = (unsigned)__hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tk = (unsigned)__builtin_amdgcn_readfirstlane((int)tk);
    if (tk != P.blocks - 1u) return;
#if KPX_ICP_ACQ_FENCE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    const int lane_w = opaque_i((int)threadIdx.x);
    if (lane_w < kAcc) s_sums[lane_w] = lane_w < NACC ? fixed_total_coherent(P.ring, lane_w) : 0.0;
    const unsigned long long n_searched = searched_take(ticket, lane_w);
    for (int e = lane_w; e < kAccSet; e += 64) __hip_atomic_store(P.ring + e, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane_w == 0) __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wave_lds_fence();
    const double *tbbox_w = P.tbbox;
    asm volatile("" : "+s"(tbbox_w));
    const double t2max_w = target_t2max(tbbox_w);
    icp_finish_wave(s_sums, P.n, MODE, k, max_iter, rel_fit, rel_rmse, &s_state, P.result, s_tail, lane_w,
                    LightSkip{ use_light ? P.sbbox : (const double *)nullptr, max_d2, t2max_w });
    wave_lds_fence();
    if (lane_w < (int)(sizeof(IcpState) / sizeof(double))) reinterpret_cast<double *>(st)[lane_w] = reinterpret_cast<const double *>(&s_state)[lane_w];
    if (certs && k + 1 < kCertHist && lane_w < 12) P.thist[12 * (k + 1) + lane_w] = s_state.T[lane_w];     // what iteration k + 1 transforms with
    if (lane_w == 0 && P.progress)
        __hip_atomic_store(P.progress, P.tag | progress_searched(n_searched, P.n) | ((unsigned long long)(s_state.done ? 1 : 0) << 32) | (unsigned long long)(unsigned)(k + 1),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace kpx
