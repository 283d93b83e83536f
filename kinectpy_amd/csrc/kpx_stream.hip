// kpx_stream.hip -- several frames of a stream in flight, scheduled natively (round 5).
//
// The reference's frame loop is sequential (preprocessing/data.py:35-61: load -> segment -> cloud -> transform -> fuse ->
// filter_outliers, one frame after the other); a frame here is a chain of ~100 short, mostly latency-bound dispatches with four
// read-backs, and consecutive frames are independent, so `depth` of them run side by side.  Rounds 2-4 did that from Python
// (pipeline.FrameStream: a ThreadPoolExecutor, futures, the GIL between a frame's end and the next one's start): under four frames in
// flight the stream of a slot then sat idle ~400 us between two frames -- 14 % of a frame's 2.8 ms (profiles/r05/
// overlap_timeline_python_framestream.txt) -- and a 20-step timing window (12 ms) moved with every interpreter hiccup.  Here the
// slots are C++ worker threads inside the library: one HIP stream and one slice of the caller's workspace each, a frame handed over
// by kpx_stream_submit (returns at once) and collected in submission order by kpx_stream_pop.  The interpreter only hands frames
// over and takes results.
//
// One GPU: a worker runs kpx_frame_step / kpx_frame_step_host.  Several GPUs (comms != NULL: one communicator per slot): a worker
// runs kpx_frame_step_sharded and the stream owns the kpx_order that gives the collectives of the frames in flight one issue order
// on every rank; a frame that outgrew its messages (KPX_RETRY on every rank alike) is submitted again inside kpx_stream_pop -- the
// same point of the caller's program on every rank, as pipeline.FrameStream.pop did.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "kpx_internal.h"

namespace kpx {

enum : int { kJobEmpty = 0, kJobQueued = 1, kJobRunning = 2, kJobDone = 3 };

// a frame handed over by kpx_stream_submit, and what came of it
struct StreamJob {
    std::atomic<int> state{ kJobEmpty };
    const void *depth = nullptr, *rgb = nullptr;
    int host_input = 0;
    float *out_pts = nullptr, *out_col = nullptr;
    int64_t frame = -1;                // (several GPUs) the frame's number in the stream's kpx_order
    int rc = KPX_OK;
    char err[512] = { 0 };
    int32_t count = 0;
    double T[16 * 16] = { 0.0 };
    int32_t info[64] = { 0 };
};
struct StreamSlot {
    std::thread th;
    hipStream_t st = nullptr;
    void *ws = nullptr;
    size_t ws_bytes = 0;
    kpx_comm *comm = nullptr;
    int index = 0;
};

}  // namespace kpx

// One GPU: the frames wait in ONE queue of 2 x depth jobs and whichever worker is free takes the oldest -- results are popped in
// submission order, and a slot whose frame finished early does not sit idle until the frames in front of it have been collected
// (measured with one job per slot: ~350 us of a 2.8 ms frame, profiles/r05/overlap_timeline_native_one_job_per_slot.txt).  Several
// GPUs: job j belongs to slot j % depth on EVERY rank (the slot's communicator carries its collectives), one job per slot.
struct kpx_stream {
    int dev = 0, depth = 1, cap = 1, sensors = 1, fused_filter = 0;
    int64_t n_px = 0;
    const float *xy = nullptr;
    std::vector<double> init;
    kpx_frame_params prm;
    std::vector<std::unique_ptr<kpx::StreamSlot>> slots;
    std::vector<std::unique_ptr<kpx::StreamJob>> jobs;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::atomic<uint64_t> submitted{ 0 };
    uint64_t claimed = 0, popped = 0;      // claimed: under mu
    std::atomic<bool> quit{ false };
    kpx_order *order = nullptr;
    bool sharded = false;
    kpx::IcpEngine *engine = nullptr;      // the device's ICP engine (kpx_icp.hip): the frames' registrations in one chain of launches
    std::atomic<uint64_t> frames_done{ 0 };
};

namespace kpx {

// a short spin before the futex: frames are a fraction of a millisecond apart, a futex wake-up costs tens of microseconds
template <class Pred> static void spin_then_wait(std::mutex &mu, std::condition_variable &cv, Pred pred)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (pred()) return;
        __builtin_ia32_pause();
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
    }
    std::unique_lock<std::mutex> lock(mu);
    cv.wait(lock, pred);
}

static void stream_worker(kpx_stream *S, StreamSlot *s)
{
    (void)hipSetDevice(S->dev);
    icp_engine_attach(S->engine);
    uint64_t mine = (uint64_t)s->index;                       // (several GPUs) the next job of this slot
    for (;;) {
        StreamJob *j = nullptr;
        for (;;) {
            spin_then_wait(S->mu, S->cv_work, [&] {
                if (S->quit.load(std::memory_order_acquire)) return true;
                if (S->sharded) return S->jobs[(size_t)(mine % (uint64_t)S->cap)]->state.load(std::memory_order_acquire) == kJobQueued;
                return S->submitted.load(std::memory_order_acquire) > __atomic_load_n(&S->claimed, __ATOMIC_ACQUIRE);
            });
            if (S->quit.load(std::memory_order_acquire)) return;
            if (S->sharded) {
                j = S->jobs[(size_t)(mine % (uint64_t)S->cap)].get();
                break;
            }
            std::lock_guard<std::mutex> lock(S->mu);
            if (S->submitted.load(std::memory_order_acquire) > S->claimed) {      // (another worker may have been faster)
                j = S->jobs[(size_t)(S->claimed % (uint64_t)S->cap)].get();
                __atomic_store_n(&S->claimed, S->claimed + 1, __ATOMIC_RELEASE);
                break;
            }
        }
        j->state.store(kJobRunning, std::memory_order_relaxed);
        int rc;
        if (S->sharded)
            rc = kpx_frame_step_sharded(s->comm, S->order, j->frame, static_cast<const uint16_t *>(j->depth), static_cast<const uint8_t *>(j->rgb), j->host_input, S->xy,
                                        S->n_px, S->sensors, S->init.data(), &S->prm, S->fused_filter, j->out_pts, j->out_col, &j->count, j->T, j->info, s->ws,
                                        s->ws_bytes, s->st);
        else if (j->host_input)
            rc = kpx_frame_step_host(static_cast<const uint16_t *>(j->depth), static_cast<const uint8_t *>(j->rgb), S->xy, S->n_px, S->sensors, S->init.data(), &S->prm,
                                     j->out_pts, j->out_col, &j->count, j->T, j->info, s->ws, s->ws_bytes, s->st);
        else
            rc = kpx_frame_step(static_cast<const uint16_t *>(j->depth), static_cast<const uint8_t *>(j->rgb), S->xy, S->n_px, S->sensors, S->init.data(), &S->prm,
                                j->out_pts, j->out_col, &j->count, j->T, j->info, s->ws, s->ws_bytes, s->st);
        if (rc < 0) {
            snprintf(j->err, sizeof(j->err), "%s", kpx_last_error());
            (void)hipStreamSynchronize(s->st);             // leave nothing of a failed frame in flight
        }
        if (S->sharded) kpx_order_finish(S->order, j->frame);      // stages the frame did not use (or did not reach) are passed
        j->rc = rc;
        S->frames_done.fetch_add(1, std::memory_order_relaxed);
        if (S->sharded && rc != KPX_RETRY) mine += (uint64_t)S->depth;      // (a frame to be run again stays this slot's next job)
        {
            std::lock_guard<std::mutex> lock(S->mu);
            j->state.store(kJobDone, std::memory_order_release);
        }
        S->cv_done.notify_all();
    }
}

static void queue_job(kpx_stream *S, StreamJob *j)
{
    if (S->sharded) (void)kpx_order_submit(S->order, &j->frame);
    {
        std::lock_guard<std::mutex> lock(S->mu);
        j->state.store(kJobQueued, std::memory_order_release);
    }
    S->cv_work.notify_all();
}

}  // namespace kpx

using namespace kpx;

KPX_EXPORT size_t kpx_stream_workspace_bytes(int32_t sensors, int32_t rank, int32_t world, int64_t n_px, int32_t depth)
{
    if (sensors < 1 || n_px < 1 || depth < 1 || depth > 16 || world < 1 || rank < 0 || rank >= world) return 0;
    const size_t per = world > 1 ? kpx_frame_step_sharded_workspace_bytes(sensors, rank, world, n_px, 1) : kpx_frame_step_host_workspace_bytes(sensors, n_px);
    if (!per) return 0;
    return (size_t)depth * ((per + 255) & ~(size_t)255);
}

KPX_EXPORT int kpx_stream_create(const float *xy_table, int64_t n_px, int32_t sensors, const double *h_init, const kpx_frame_params *prm, int32_t depth,
                                 kpx_comm *const *comms, int32_t fused_filter, void *ws, size_t ws_bytes, kpx_stream **out)
{
    KPX_REQUIRE(out && xy_table && prm && ws, "kpx_stream_create: null pointer");
    KPX_REQUIRE(sensors >= 1 && sensors <= 16 && n_px > 0 && depth >= 1 && depth <= 16, "kpx_stream_create: 1 .. 16 sensors, 1 .. 16 frames in flight");
    KPX_REQUIRE(sensors == 1 || h_init, "kpx_stream_create: initial transforms missing");
    int rank = 0, world = 1;
    if (comms) {
        for (int i = 0; i < depth; ++i) KPX_REQUIRE(comms[i], "kpx_stream_create: one communicator per frame slot");
        rank = kpx_comm_rank(comms[0]);
        world = kpx_comm_world(comms[0]);
    }
    const size_t need = kpx_stream_workspace_bytes(sensors, rank, world, n_px, depth);
    KPX_REQUIRE(need > 0, "kpx_stream_create: bad shape");
    if (ws_bytes < need) return fail(KPX_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", need, ws_bytes);
    std::unique_ptr<kpx_stream> S(new kpx_stream());
    KPX_HIP(hipGetDevice(&S->dev));
    S->depth = depth; S->sensors = sensors; S->n_px = n_px; S->xy = xy_table; S->prm = *prm; S->fused_filter = fused_filter;
    S->sharded = comms != nullptr;
    S->cap = S->sharded ? depth : 2 * depth;
    S->init.assign((size_t)16 * (sensors > 1 ? sensors - 1 : 1), 0.0);
    if (sensors > 1) memcpy(S->init.data(), h_init, sizeof(double) * 16 * (size_t)(sensors - 1));
    if (S->sharded) {
        // The collectives' issue order (kpx_order): frame f's exchange and slab all-gather go behind the master broadcast of frame
        // f + lookahead.  Rounds 3-4 tied the lookahead to the number of slots (depth - 1): with every slot busy, frame f + depth - 1
        // cannot START before frame f - 1 has been collected, so frame f's exchange waited for a frame that waited for its predecessor
        // -- measured on one rank of the 8-GPU partition (bench.py --emulate-world 8): 1.9 of a frame's 3.1 ms idle in front of the
        // exchange, 0.64 ms per frame whatever the rank computes (profiles/r05/emulate_world8_rank1_lookahead3_overlap.txt).  The
        // lookahead is now its own number (KPX_ORDER_LOOKAHEAD, default 2: the exchange of a frame is ready ~1.3 ms after its broadcast,
        // by when two later frames have broadcast theirs), the slots can be more (depth 6-8): no frame waits for a slot to open.
        static const int look_env = [] { const char *e = getenv("KPX_ORDER_LOOKAHEAD"); return e ? atoi(e) : -1; }();
        int look = look_env >= 0 ? look_env : 2;
        if (look > depth - 1) look = depth - 1;
        const int rc = kpx_order_create(look + 1, &S->order);
        if (rc) return rc;
    }
    // KPX_STREAM_ENGINE=1: the device's ICP engine carries the registrations of all frames in flight in one launch per tick (kpx_icp.hip,
    // IcpEngine).  Built and measured in round 5: 1550 (one chain) / 1720 (two chains) against 2740-2800 Mpoints/s with a chain of launches per
    // frame, same box (profiles/r05/exp_icp_engine_*.txt) -- one chain advances every frame at the pace of the slowest launch and leaves
    // the tail of every launch unfilled, where four independent chains fill each other's tails.  Default: off.
    static const bool engine_on = [] { const char *e = getenv("KPX_STREAM_ENGINE"); return e && e[0] == '1'; }();
    S->engine = engine_on ? icp_engine_acquire() : nullptr;
    for (int i = 0; i < S->cap; ++i) S->jobs.emplace_back(new StreamJob());
    const size_t per = need / (size_t)depth;
    for (int i = 0; i < depth; ++i) {
        std::unique_ptr<StreamSlot> s(new StreamSlot());
        s->index = i;
        s->ws = static_cast<char *>(ws) + (size_t)i * per;
        s->ws_bytes = per;
        s->comm = comms ? comms[i] : nullptr;
        if (hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking) != hipSuccess) {
            for (auto &p : S->slots) (void)hipStreamDestroy(p->st);
            if (S->order) kpx_order_destroy(S->order);
            icp_engine_release(S->engine);
            return fail(KPX_ERR_HIP, "kpx_stream_create: hipStreamCreateWithFlags failed");
        }
        S->slots.push_back(std::move(s));
    }
    for (auto &p : S->slots) p->th = std::thread(stream_worker, S.get(), p.get());
    *out = S.release();
    return KPX_OK;
}

KPX_EXPORT int kpx_stream_pending(const kpx_stream *S) { return S ? (int)(S->submitted.load() - S->popped) : 0; }
KPX_EXPORT int kpx_stream_capacity(const kpx_stream *S) { return S ? S->cap : 0; }

KPX_EXPORT int kpx_stream_submit(kpx_stream *S, const void *depth, const void *rgb, int32_t host_input, float *out_pts, float *out_col)
{
    KPX_REQUIRE(S && depth && rgb && out_pts && out_col, "kpx_stream_submit: null pointer");
    const uint64_t n = S->submitted.load(std::memory_order_relaxed);
    KPX_REQUIRE((int)(n - S->popped) < S->cap, "kpx_stream_submit: %d frames queued already (kpx_stream_pop first)", S->cap);
    StreamJob *j = S->jobs[(size_t)(n % (uint64_t)S->cap)].get();
    j->depth = depth; j->rgb = rgb; j->host_input = host_input; j->out_pts = out_pts; j->out_col = out_col;
    if (S->sharded) (void)kpx_order_submit(S->order, &j->frame);
    {
        std::lock_guard<std::mutex> lock(S->mu);
        j->state.store(kJobQueued, std::memory_order_release);
        S->submitted.store(n + 1, std::memory_order_release);
    }
    S->cv_work.notify_all();
    return KPX_OK;
}

KPX_EXPORT int kpx_stream_pop(kpx_stream *S, int32_t *h_count, double *h_T, int32_t *h_info)
{
    KPX_REQUIRE(S && h_count && h_T, "kpx_stream_pop: null pointer");
    KPX_REQUIRE(S->submitted.load() > S->popped, "kpx_stream_pop: no frame in flight");
    StreamJob *j = S->jobs[(size_t)(S->popped % (uint64_t)S->cap)].get();
    for (;;) {
        if (S->sharded) kpx_order_block(S->order, j->frame);      // no frame can be submitted before this one is done: see kpx_order
        spin_then_wait(S->mu, S->cv_done, [&] { return j->state.load(std::memory_order_acquire) == kJobDone; });
        if (S->sharded) kpx_order_block(S->order, -1);
        if (j->rc != KPX_RETRY) break;
        queue_job(S, j);                                           // every rank alike: the frame runs again under a new frame number
    }
    ++S->popped;
    j->state.store(kJobEmpty, std::memory_order_relaxed);
    if (j->rc < 0) return fail(j->rc, "%s", j->err);
    *h_count = j->count;
    memcpy(h_T, j->T, sizeof(double) * 16 * (size_t)S->sensors);
    if (h_info) memcpy(h_info, j->info, sizeof(j->info));
    return KPX_OK;
}

KPX_EXPORT int kpx_stream_destroy(kpx_stream *S)
{
    if (!S) return KPX_OK;
    int32_t c;
    double T[16 * 16];
    while (S->submitted.load() > S->popped) (void)kpx_stream_pop(S, &c, T, nullptr);
    {
        std::lock_guard<std::mutex> lock(S->mu);
        S->quit.store(true, std::memory_order_release);
    }
    S->cv_work.notify_all();
    for (auto &p : S->slots) {
        if (p->th.joinable()) p->th.join();
        (void)hipStreamSynchronize(p->st);
        (void)hipStreamDestroy(p->st);
    }
    if (S->order) kpx_order_destroy(S->order);
    icp_engine_release(S->engine);
    delete S;
    return KPX_OK;
}

KPX_EXPORT int kpx_stream_stats(const kpx_stream *S, uint64_t *h_out4)
{
    KPX_REQUIRE(S && h_out4, "kpx_stream_stats: null pointer");
    unsigned long long launches = 0ull, ticks = 0ull;
    icp_engine_counters(S->engine, &launches, &ticks);
    h_out4[0] = S->frames_done.load();
    h_out4[1] = S->engine ? 1u : 0u;
    h_out4[2] = launches;
    h_out4[3] = ticks;
    return KPX_OK;
}
