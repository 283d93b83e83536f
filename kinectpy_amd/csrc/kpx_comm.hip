// kpx_comm.hip -- the exchange layer of the sensor partition (preprocessing/data.py:44-61, 127-161 across GPUs), native:
//
//   kpx_comm    one communicator over all ranks (one PROCESS per GPU; one communicator per frame slot).  Two transports behind one
//               interface: RCCL called from C++ on the frame's stream (librccl is dlopen'ed at run time, so the library still loads on
//               a box without it; the unique id is created here and carried to the other ranks by the caller, e.g. through
//               torch.distributed), and caller-supplied callbacks (host-staged gloo for rehearsals with several ranks on one GPU,
//               in-process ranks in tests).  It also holds what a slot remembers from frame to frame: the adaptive message
//               capacities (the same on every rank by construction) and the speculated sort-key width.
//   kpx_order   one global issue order for the collectives of the frames in flight on a rank -- the C++ form of
//               parallel.CollectiveOrder: a collective kernel spins on the device until its peers arrive, so every rank has to
//               enqueue the collectives of its frames in flight in the same order whatever the timing of its host threads.
#include <dlfcn.h>

#include <condition_variable>
#include <mutex>
#include <set>
#include <vector>

#include "kpx_internal.h"

namespace kpx {

// ---- RCCL through dlopen ------------------------------------------------------------------------------------------------
// The few prototypes used, declared here so that neither this file nor the build depends on rccl.h being installed
// (/opt/rocm/include/rccl/rccl.h: ncclUniqueId is 128 opaque bytes, ncclInt8 = 0, ncclSuccess = 0).
struct NcclId { char internal[128]; };
typedef int (*nccl_get_unique_id_t)(NcclId *);
typedef int (*nccl_comm_init_rank_t)(void **, int, NcclId, int);
typedef int (*nccl_comm_destroy_t)(void *);
typedef int (*nccl_broadcast_t)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_all_gather_t)(const void *, void *, size_t, int, void *, hipStream_t);
typedef const char *(*nccl_error_string_t)(int);

struct Rccl {
    void *handle = nullptr;
    nccl_get_unique_id_t get_unique_id = nullptr;
    nccl_comm_init_rank_t comm_init_rank = nullptr;
    nccl_comm_destroy_t comm_destroy = nullptr;
    nccl_broadcast_t broadcast = nullptr;
    nccl_all_gather_t all_gather = nullptr;
    nccl_error_string_t error_string = nullptr;
};
static Rccl g_rccl;
static std::mutex g_rccl_mutex;

static int rccl_fail(const char *what, int rc)
{
    return fail(KPX_ERR_HIP, "%s: RCCL error %d (%s)", what, rc, g_rccl.error_string ? g_rccl.error_string(rc) : "?");
}

}  // namespace kpx

using namespace kpx;

struct kpx_comm {
    int32_t rank = 0, world = 1;
    void *nccl = nullptr;                       // ncclComm_t (RCCL transport)
    kpx_bcast_fn bcast = nullptr;               // callback transport
    kpx_allgather_fn allgather = nullptr;
    void *user = nullptr;
    // replay transport (kpx_comm_create_replay): what every rank SENT in every collective of `frames` recorded frames
    std::vector<const void *> rp_ptr;
    std::vector<size_t> rp_bytes;
    int rp_frames = 0, rp_first = 0, rp_stride = 1, rp_per = 3;
    long long rp_calls = 0;
    // slot memory (kpx_frame_step_sharded): rows of the master broadcast / the cloud exchange, 0 = not seen a frame yet
    int64_t cap_master = 0, cap_clouds = 0;
    int spec_bits = 0, fuse_bits = 0;
};

KPX_EXPORT int kpx_rccl_load(const char *path)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return KPX_OK;
    void *h = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(KPX_ERR_INVALID, "kpx_rccl_load: %s", dlerror());
    Rccl r;
    r.handle = h;
    r.get_unique_id = (nccl_get_unique_id_t)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (nccl_comm_init_rank_t)dlsym(h, "ncclCommInitRank");
    r.comm_destroy = (nccl_comm_destroy_t)dlsym(h, "ncclCommDestroy");
    r.broadcast = (nccl_broadcast_t)dlsym(h, "ncclBroadcast");
    r.all_gather = (nccl_all_gather_t)dlsym(h, "ncclAllGather");
    r.error_string = (nccl_error_string_t)dlsym(h, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.broadcast || !r.all_gather)
        return fail(KPX_ERR_INVALID, "kpx_rccl_load: %s lacks the collective entry points", path ? path : "librccl.so");
    g_rccl = r;
    return KPX_OK;
}

KPX_EXPORT int kpx_rccl_unique_id(void *id128)
{
    KPX_REQUIRE(id128, "kpx_rccl_unique_id: null pointer");
    KPX_REQUIRE(g_rccl.handle, "kpx_rccl_unique_id: call kpx_rccl_load first");
    const int rc = g_rccl.get_unique_id(static_cast<NcclId *>(id128));
    return rc ? rccl_fail("ncclGetUniqueId", rc) : KPX_OK;
}

KPX_EXPORT int kpx_comm_create_rccl(const void *id128, int32_t rank, int32_t world, kpx_comm **out)
{
    KPX_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "kpx_comm_create_rccl: bad arguments");
    KPX_REQUIRE(g_rccl.handle, "kpx_comm_create_rccl: call kpx_rccl_load first");
    NcclId id;
    memcpy(&id, id128, sizeof(id));
    void *c = nullptr;
    const int rc = g_rccl.comm_init_rank(&c, world, id, rank);          // collective: every rank calls it, in the same order per slot
    if (rc) return rccl_fail("ncclCommInitRank", rc);
    kpx_comm *k = new kpx_comm();
    k->rank = rank;
    k->world = world;
    k->nccl = c;
    *out = k;
    return KPX_OK;
}

KPX_EXPORT int kpx_comm_create_callbacks(int32_t rank, int32_t world, kpx_bcast_fn bcast, kpx_allgather_fn allgather, void *user, kpx_comm **out)
{
    KPX_REQUIRE(out && world >= 1 && rank >= 0 && rank < world && (world == 1 || (bcast && allgather)), "kpx_comm_create_callbacks: bad arguments");
    kpx_comm *k = new kpx_comm();
    k->rank = rank;
    k->world = world;
    k->bcast = bcast;
    k->allgather = allgather;
    k->user = user;
    *out = k;
    return KPX_OK;
}

// Replay transport -- a MEASUREMENT aid (bench.py --emulate-world): the collectives of ONE rank of a `world`-rank job on one GPU, the
// peers' contributions taken from recordings of a real `world`-rank run (in-process ranks) over `frames` frames: entry
// (per_frame = 3; 2: frames without the slab all-gather, fused_filter 1 / 2)
// [(f * 3 + c) * world + r] = what rank r sent in collective c (0 master broadcast: root's entry only, 1 cloud exchange, 2 slab
// all-gather) of frame f, in device memory.  The communicator counts its calls: call n is collective n % 3 of frame
// (first_frame + stride * (n / 3)) % frames -- the frames a slot of a kpx_stream sees (slot s of depth d: first_frame = s,
// stride = d).  The message sizes must be those of the recording: both runs use fixed worst-case capacities (KPX_SHARD_FIXED_CAP=1).
KPX_EXPORT int kpx_comm_create_replay(int32_t rank, int32_t world, int32_t frames, int32_t first_frame, int32_t stride, int32_t per_frame,
                                      const void *const *d_payloads, const size_t *bytes, kpx_comm **out)
{
    KPX_REQUIRE(out && d_payloads && bytes && world >= 1 && rank >= 0 && rank < world && frames >= 1 && first_frame >= 0 && stride >= 1 &&
                (per_frame == 2 || per_frame == 3), "kpx_comm_create_replay: bad arguments");
    kpx_comm *k = new kpx_comm();
    k->rank = rank;
    k->world = world;
    k->rp_frames = frames; k->rp_first = first_frame; k->rp_stride = stride; k->rp_per = per_frame;
    const size_t n = (size_t)frames * 3 * (size_t)world;
    k->rp_ptr.assign(d_payloads, d_payloads + n);
    k->rp_bytes.assign(bytes, bytes + n);
    *out = k;
    return KPX_OK;
}
static int replay_entry(kpx_comm *c, int want_c, size_t *base)
{
    const long long n = c->rp_calls++;
    const int col = (int)(n % c->rp_per);
    if (col != want_c) return fail(KPX_ERR_INVALID, "replay transport: call %lld is collective %d of its frame, the recording has %d there", n, want_c, col);
    const long long f = ((long long)c->rp_first + (long long)c->rp_stride * (n / c->rp_per)) % c->rp_frames;
    *base = ((size_t)f * 3 + (size_t)col) * (size_t)c->world;
    return KPX_OK;
}

KPX_EXPORT int kpx_comm_destroy(kpx_comm *c)
{
    if (!c) return KPX_OK;
    int rc = 0;
    if (c->nccl && g_rccl.comm_destroy) rc = g_rccl.comm_destroy(c->nccl);
    delete c;
    return rc ? rccl_fail("ncclCommDestroy", rc) : KPX_OK;
}

KPX_EXPORT int kpx_comm_rank(const kpx_comm *c) { return c ? c->rank : 0; }
KPX_EXPORT int kpx_comm_world(const kpx_comm *c) { return c ? c->world : 1; }

// In-place broadcast of `bytes` bytes of device memory from `root`; asynchronous on `stream` (RCCL) or as the callback decides.
KPX_EXPORT int kpx_comm_broadcast(kpx_comm *c, void *d_buf, size_t bytes, int32_t root, void *stream)
{
    KPX_REQUIRE(c && (d_buf || bytes == 0) && root >= 0 && root < c->world, "kpx_comm_broadcast: bad arguments");
    if (bytes == 0) return KPX_OK;
    if (c->nccl) {
        const int rc = g_rccl.broadcast(d_buf, d_buf, bytes, /*ncclInt8*/ 0, root, c->nccl, (hipStream_t)stream);
        return rc ? rccl_fail("ncclBroadcast", rc) : KPX_OK;
    }
    if (c->rp_frames) {
        size_t base;
        const int rc = replay_entry(c, 0, &base);
        if (rc) return rc;
        if (c->rank != root) {
            KPX_REQUIRE(c->rp_bytes[base + (size_t)root] == bytes && c->rp_ptr[base + (size_t)root], "replay transport: the recorded broadcast has %zu bytes, this one %zu",
                        c->rp_bytes[base + (size_t)root], bytes);
            KPX_HIP(hipMemcpyAsync(d_buf, c->rp_ptr[base + (size_t)root], bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        }
        return KPX_OK;
    }
    if (c->bcast) {
        const int rc = c->bcast(c->user, d_buf, bytes, root, stream);
        return rc ? fail(KPX_ERR_HIP, "kpx_comm_broadcast: the transport callback failed (%d)", rc) : KPX_OK;
    }
    KPX_REQUIRE(c->world == 1, "kpx_comm_broadcast: communicator without a transport");
    return KPX_OK;
}

// d_recv [world][bytes_per_rank] <- every rank's d_send [bytes_per_rank] (d_send may be the rank's own slot of d_recv).
KPX_EXPORT int kpx_comm_allgather(kpx_comm *c, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream)
{
    KPX_REQUIRE(c && ((d_send && d_recv) || bytes_per_rank == 0), "kpx_comm_allgather: bad arguments");
    if (bytes_per_rank == 0) return KPX_OK;
    if (c->nccl) {
        const int rc = g_rccl.all_gather(d_send, d_recv, bytes_per_rank, /*ncclInt8*/ 0, c->nccl, (hipStream_t)stream);
        return rc ? rccl_fail("ncclAllGather", rc) : KPX_OK;
    }
    if (c->rp_frames) {
        size_t base;
        const int col = (int)(c->rp_calls % c->rp_per);
        const int rc = replay_entry(c, col == 0 ? 1 : col, &base);
        if (rc) return rc;
        for (int r = 0; r < c->world; ++r) {
            char *dst = static_cast<char *>(d_recv) + (size_t)r * bytes_per_rank;
            if (r == c->rank) {
                if (dst != d_send) KPX_HIP(hipMemcpyAsync(dst, d_send, bytes_per_rank, hipMemcpyDeviceToDevice, (hipStream_t)stream));
                continue;
            }
            KPX_REQUIRE(c->rp_bytes[base + (size_t)r] == bytes_per_rank && c->rp_ptr[base + (size_t)r], "replay transport: rank %d's recorded message has %zu bytes, this one %zu",
                        r, c->rp_bytes[base + (size_t)r], bytes_per_rank);
            KPX_HIP(hipMemcpyAsync(dst, c->rp_ptr[base + (size_t)r], bytes_per_rank, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        }
        return KPX_OK;
    }
    if (c->allgather) {
        const int rc = c->allgather(c->user, d_send, d_recv, bytes_per_rank, stream);
        return rc ? fail(KPX_ERR_HIP, "kpx_comm_allgather: the transport callback failed (%d)", rc) : KPX_OK;
    }
    KPX_REQUIRE(c->world == 1, "kpx_comm_allgather: communicator without a transport");
    char *mine = static_cast<char *>(d_recv) + (size_t)c->rank * bytes_per_rank;
    if (mine != d_send) KPX_HIP(hipMemcpyAsync(mine, d_send, bytes_per_rank, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return KPX_OK;
}

// Utility for callback transports (and tests): copy `bytes` bytes between any two of host / device memory on `stream`
// (hipMemcpyDefault); wait != 0 also waits for the stream.
KPX_EXPORT int kpx_copy_bytes(void *dst, const void *src, size_t bytes, void *stream, int32_t wait)
{
    KPX_REQUIRE((dst && src) || bytes == 0, "kpx_copy_bytes: null pointer");
    if (bytes) KPX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, (hipStream_t)stream));
    if (wait) KPX_HIP(hipStreamSynchronize((hipStream_t)stream));
    return KPX_OK;
}

// slot memory, for kpx_frame.hip
namespace kpx {
int64_t &comm_cap_master(kpx_comm *c) { return c->cap_master; }
int64_t &comm_cap_clouds(kpx_comm *c) { return c->cap_clouds; }
int &comm_spec_bits(kpx_comm *c) { return c->spec_bits; }
int &comm_fuse_bits(kpx_comm *c) { return c->fuse_bits; }
}  // namespace kpx

// ---- the collectives' issue order ---------------------------------------------------------------------------------------
// Stage s (0 master broadcast, 1 cloud exchange, 2 slab all-gather) of frame f has the key 3 f for s = 0 and 3 (f + depth - 1) + s
// otherwise: frame f + 1's broadcast goes before frame f's exchange (the registration lies between them), the software pipeline's
// natural order.  A thread may issue a collective only when every smaller key of the frames submitted so far is done; when the
// next frame has not been submitted yet and its broadcast key is smaller, it waits until the main thread either submits that frame
// or blocks waiting for THIS one (then no frame can come before it on any rank: the main thread's submit / pop sequence is the same
// program everywhere).  Same rules as parallel.CollectiveOrder, whose tests run against both.
struct kpx_order {
    int depth = 1;
    std::mutex m;
    std::condition_variable cv;
    std::set<int64_t> pending;
    int64_t submitted = 0;
    int64_t waiting_for = -1;
    std::vector<int64_t> log;
    int64_t key(int64_t frame, int stage) const { return stage == 0 ? 3 * frame : 3 * (frame + depth - 1) + stage; }
};

KPX_EXPORT int kpx_order_create(int32_t depth, kpx_order **out)
{
    KPX_REQUIRE(out, "kpx_order_create: null pointer");
    kpx_order *o = new kpx_order();
    o->depth = depth < 1 ? 1 : depth;
    *out = o;
    return KPX_OK;
}
KPX_EXPORT int kpx_order_destroy(kpx_order *o)
{
    delete o;
    return KPX_OK;
}

KPX_EXPORT int kpx_order_submit(kpx_order *o, int64_t *frame)
{
    KPX_REQUIRE(o && frame, "kpx_order_submit: null pointer");
    std::lock_guard<std::mutex> lock(o->m);
    const int64_t f = o->submitted++;
    for (int s = 0; s < 3; ++s) o->pending.insert(o->key(f, s));
    o->cv.notify_all();
    *frame = f;
    return KPX_OK;
}

KPX_EXPORT int kpx_order_block(kpx_order *o, int64_t frame)       // frame < 0: the main thread's wait is over
{
    KPX_REQUIRE(o, "kpx_order_block: null pointer");
    std::lock_guard<std::mutex> lock(o->m);
    o->waiting_for = frame;
    o->cv.notify_all();
    return KPX_OK;
}

KPX_EXPORT int kpx_order_turn_begin(kpx_order *o, int64_t frame, int32_t stage)
{
    if (!o) return KPX_OK;
    std::unique_lock<std::mutex> lock(o->m);
    const int64_t k = o->key(frame, stage);
    o->cv.wait(lock, [&] { return !o->pending.empty() && k == *o->pending.begin() && (k < 3 * o->submitted || o->waiting_for == frame); });
    o->log.push_back(k);
    return KPX_OK;
}

KPX_EXPORT int kpx_order_turn_end(kpx_order *o, int64_t frame, int32_t stage)
{
    if (!o) return KPX_OK;
    std::lock_guard<std::mutex> lock(o->m);
    o->pending.erase(o->key(frame, stage));
    o->cv.notify_all();
    return KPX_OK;
}

KPX_EXPORT int kpx_order_skip(kpx_order *o, int64_t frame, int32_t stage) { return kpx_order_turn_end(o, frame, stage); }

KPX_EXPORT int kpx_order_finish(kpx_order *o, int64_t frame)      // the frame is over (also after an error): unreached stages are passed
{
    if (!o) return KPX_OK;
    std::lock_guard<std::mutex> lock(o->m);
    for (int s = 0; s < 3; ++s) o->pending.erase(o->key(frame, s));
    o->cv.notify_all();
    return KPX_OK;
}

KPX_EXPORT int kpx_order_log(kpx_order *o, int64_t *out, int64_t cap, int64_t *count)
{
    KPX_REQUIRE(o && count, "kpx_order_log: null pointer");
    std::lock_guard<std::mutex> lock(o->m);
    const int64_t n = (int64_t)o->log.size();
    for (int64_t i = 0; i < n && i < cap && out; ++i) out[i] = o->log[(size_t)i];
    *count = n;
    return KPX_OK;
}
