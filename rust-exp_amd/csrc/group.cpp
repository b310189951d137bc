// group.cpp -- single-process multi-GPU group (nbx_group_*), RCCL resolved with dlopen.
#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "engine_internal.h"

using namespace nbxi;

// =============================================================================================
// Single-process multi-GPU group: what the unmodified Haskell caller needs to use every GPU of a node.
// G engines, one per device, slab-sharded exactly like the multi-process path (nbody.rs:426-428 split);
// per step every device runs K1+K2 on its slab on its own stream, then ONE RCCL all-gather of the
// (x,y,z,m) array (ncclCommInitAll communicators, one group call).  RCCL is dlopen'ed on first use so that
// single-GPU users carry no dependency on it.
// =============================================================================================

namespace {

struct RcclApi {
    void* so = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;   // optional: used to tear down a collective that only some ranks joined
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi* rccl_api()
{
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (so) {
            api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(dlsym(so, "ncclCommInitAll"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(so, "ncclCommDestroy"));
            api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(so, "ncclCommAbort"));
            api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(so, "ncclAllGather"));
            api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(dlsym(so, "ncclBroadcast"));
            api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(so, "ncclGroupStart"));
            api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(so, "ncclGroupEnd"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(so, "ncclGetErrorString"));
            if (api.CommInitAll && api.CommDestroy && api.AllGather && api.Broadcast && api.GroupStart && api.GroupEnd &&
                api.GetErrorString)
                api.so = so;
        }
    }
    return api.so ? &api : nullptr;
}

#define RCCL_TRY(api, expr)                                                                         \
    do {                                                                                            \
        ncclResult_t _r = (expr);                                                                   \
        if (_r != ncclSuccess) return fail(NBX_ERR_HIP, "%s failed: %s", #expr, (api)->GetErrorString(_r)); \
    } while (0)

}  // namespace


// ---------------------------------------------------------------------------------------------------------------------
// Optional enqueue threads (NBX_GROUP_ENQUEUE=threads or nbx_group_set_enqueue_threads): one persistent host thread per
// engine, so that the order in which ONE thread walks the devices cannot add rank skew: every
// device's K1 + K2 + its share of the exchange is enqueued by its own thread, concurrently.  A step is a short list of
// phases; a barrier separates consecutive phases (the copy exchange needs "every slab's ready-event is recorded" before
// anybody waits on one: hipStreamWaitEvent on a never-recorded event is a no-op).
// ---------------------------------------------------------------------------------------------------------------------
struct GroupWorkers {
    int G = 0;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    const std::vector<std::function<int(int)>>* phases = nullptr;
    uint64_t epoch = 0;
    int pending = 0;
    bool quit = false;
    std::atomic<int> failed{0};
    std::vector<int> rc;
    std::vector<std::string> err;
    std::mutex bmu;
    std::condition_variable bcv;
    int arrived = 0;
    uint64_t bgen = 0;

    void barrier()
    {
        std::unique_lock<std::mutex> lk(bmu);
        const uint64_t gen = bgen;
        if (++arrived == G) {
            arrived = 0;
            bgen++;
            bcv.notify_all();
        } else {
            bcv.wait(lk, [&] { return bgen != gen; });
        }
    }

    void loop(int d)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::vector<std::function<int(int)>>* ph = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return quit || epoch != seen; });
                if (quit) return;
                seen = epoch;
                ph = phases;
            }
            int my = NBX_OK;
            for (size_t k = 0; k < ph->size(); k++) {
                if (my == NBX_OK && !failed.load()) {
                    my = (*ph)[k](d);
                    if (my != NBX_OK) {
                        err[(size_t)d] = g_last_error;   // thread-local text: hand it to the caller's thread
                        failed.store(1);
                    }
                }
                if (k + 1 < ph->size()) barrier();
            }
            rc[(size_t)d] = my;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }

    explicit GroupWorkers(int count) : G(count), rc((size_t)count, NBX_OK), err((size_t)count)
    {
        for (int d = 0; d < G; d++) th.emplace_back([this, d] { loop(d); });
    }

    ~GroupWorkers()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv_go.notify_all();
        for (std::thread& t : th) t.join();
    }

    int run(const std::vector<std::function<int(int)>>& ph)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            phases = &ph;
            failed.store(0);
            pending = G;
            epoch++;
        }
        cv_go.notify_all();
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return pending == 0; });
        }
        for (int d = 0; d < G; d++)
            if (rc[(size_t)d] != NBX_OK) {
                g_last_error = err[(size_t)d];
                return rc[(size_t)d];
            }
        return NBX_OK;
    }
};

namespace {

// every phase for every engine: phase k of all engines completes (is enqueued) before phase k + 1 of any starts
int run_phases(nbx_group* g, const std::vector<std::function<int(int)>>& phases)
{
    if (g->workers) return g->workers->run(phases);
    const int G = (int)g->eng.size();
    for (const auto& ph : phases)
        for (int d = 0; d < G; d++) {
            const int rc = ph(d);
            if (rc != NBX_OK) return rc;
        }
    return NBX_OK;
}

// abort = true: a collective may have been enqueued by only SOME ranks (one rank's enqueue failed after its peers' kernels
// were launched: they spin waiting for it) -- ncclCommAbort tears those kernels down, where a stream synchronisation or
// ncclCommDestroy would wait for them forever
void group_destroy_comms(nbx_group* g, bool abort = false)
{
    if (g->comms.empty()) return;
    if (RcclApi* api = rccl_api())
        for (ncclComm_t c : g->comms)
            if (c) (void)((abort && api->CommAbort) ? api->CommAbort(c) : api->CommDestroy(c));
    g->comms.clear();
    g->rccl_ranks = 0;
}

int group_comms(nbx_group* g)
{
    if (!g->comms.empty()) return NBX_OK;
    if (g->rccl_fail_hook == 1) return fail(NBX_ERR_HIP, "ncclCommInitAll failed: simulated (NBX_GROUP_RCCL_FAIL=init)");
    for (size_t i = 0; i < g->devices.size(); i++)
        for (size_t j = 0; j < i; j++)
            if (g->devices[i] == g->devices[j]) return fail(NBX_ERR_HIP, "RCCL needs one rank per device (device %d listed twice)", g->devices[i]);
    RcclApi* api = rccl_api();
    if (!api) return fail(NBX_ERR_HIP, "librccl.so could not be loaded: %s", dlerror());
    g->comms.assign(g->eng.size(), nullptr);
    const ncclResult_t r = api->CommInitAll(g->comms.data(), (int)g->eng.size(), g->devices.data());
    if (r != ncclSuccess) {
        g->comms.clear();
        return fail(NBX_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", (int)g->eng.size(), api->GetErrorString(r));
    }
    g->rccl_ranks = (int)g->eng.size();
    return NBX_OK;
}

// peer access between every pair of distinct devices of the group (once): hipMemcpyPeerAsync then goes GPU to GPU over
// xGMI instead of staging through the host.  Failures are not fatal -- the copies still work, slower.
void group_enable_peer_access(nbx_group* g)
{
    for (size_t a = 0; a < g->devices.size(); a++)
        for (size_t b = 0; b < g->devices.size(); b++) {
            if (g->devices[a] == g->devices[b]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, g->devices[a], g->devices[b]) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
            if (hipSetDevice(g->devices[a]) != hipSuccess) { (void)hipGetLastError(); continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(g->devices[b], 0);
            if (e != hipSuccess) (void)hipGetLastError();   // hipErrorPeerAccessAlreadyEnabled included
        }
}

// RCCL is unusable (library missing, communicator creation or a collective failed): keep the run alive on the peer-copy
// exchange and say so (VERDICT r02 next #1b) -- the group reports exchange kind 2 and the reason from then on.
void group_fall_back_to_copy(nbx_group* g, const char* why)
{
    group_destroy_comms(g);
    g->copy_exchange = true;
    g->exchange_kind = 2;
    g->exchange_note = why ? why : "";
    std::fprintf(stderr, "[nbx] group of %d: RCCL exchange unavailable (%s); falling back to event-ordered hipMemcpyPeerAsync pulls\n",
                 (int)g->eng.size(), g->exchange_note.c_str());
}

int group_copy_events(nbx_group* g)
{
    const int G = (int)g->eng.size();
    if (!g->ev_ready.empty()) return NBX_OK;
    group_enable_peer_access(g);
    g->ev_ready.assign((size_t)G, nullptr);
    g->ev_copied.assign((size_t)G, nullptr);
    for (int d = 0; d < G; d++) {
        HIP_TRY(hipSetDevice(g->eng[d]->device));
        HIP_TRY(hipEventCreateWithFlags(&g->ev_ready[d], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g->ev_copied[d], hipEventDisableTiming));
    }
    return NBX_OK;
}

// The exchange without RCCL: every engine pulls every other engine's slab with hipMemcpyPeerAsync on its own stream,
// ordered by events -- a slab is read only after its owner's kick-drift (ev_ready), and nobody starts the next step
// before everyone has pulled from it (ev_copied).  Works with or without peer access, needs no communicator, and lets
// several engines share one device (how the group logic is tested on a single-GPU box).  Three phases per engine.
int group_exchange_copy(nbx_group* g, bool half)
{
    const int G = (int)g->eng.size();
    const size_t rec = half ? 8 : sizeof(float4);   // half4 source copy or float4 (x,y,z,m)
    int rc = group_copy_events(g);
    if (rc != NBX_OK) return rc;
    std::vector<std::unique_ptr<ProfScope>> scopes((size_t)G);   // per engine: start after its own kernels, stop when it may go on
    const std::vector<std::function<int(int)>> phases = {
        [&](int d) -> int {
            HIP_TRY(hipSetDevice(g->eng[d]->device));
            HIP_TRY(hipEventRecord(g->ev_ready[d], g->eng[d]->stream));
            return NBX_OK;
        },
        [&](int d) -> int {
            nbx_engine* dst = g->eng[d];
            HIP_TRY(hipSetDevice(dst->device));
            scopes[(size_t)d].reset(new ProfScope(dst, NBX_K_EXCHANGE));
            for (int k = 1; k < G; k++) {
                const int s = (d + k) % G;   // staggered: at any moment the G pulls target G different owners
                nbx_engine* src = g->eng[s];
                if (src->slab() == 0) continue;
                HIP_TRY(hipStreamWaitEvent(dst->stream, g->ev_ready[s], 0));
                char* to = (half ? static_cast<char*>(dst->d_posh) : reinterpret_cast<char*>(dst->d_posm)) + rec * (size_t)src->lo;
                const char* from = (half ? static_cast<const char*>(src->d_posh) : reinterpret_cast<const char*>(src->d_posm)) + rec * (size_t)src->lo;
                HIP_TRY(hipMemcpyPeerAsync(to, dst->device, from, src->device, rec * (size_t)src->slab(), dst->stream));
            }
            HIP_TRY(hipEventRecord(g->ev_copied[d], dst->stream));
            return NBX_OK;
        },
        [&](int d) -> int {
            HIP_TRY(hipSetDevice(g->eng[d]->device));
            for (int s = 0; s < G; s++)
                if (s != d) HIP_TRY(hipStreamWaitEvent(g->eng[d]->stream, g->ev_copied[s], 0));
            scopes[(size_t)d].reset();   // stop event on d's stream, with d's device current
            return NBX_OK;
        }};
    rc = run_phases(g, phases);
    for (int d = 0; d < G; d++)
        if (scopes[(size_t)d]) { (void)hipSetDevice(g->eng[d]->device); scopes[(size_t)d].reset(); }
    return rc;
}

// one in-place all-gather (sendbuff = recvbuff + lo) of ONE array through RCCL, on the same streams as the kernels
int group_exchange_rccl(nbx_group* g, bool half)
{
    const int G = (int)g->eng.size();
    const int n = g->eng[0]->n;
    if (g->rccl_fail_hook == 2) {
        g->rccl_fail_hook = 0;
        return fail(NBX_ERR_HIP, "ncclAllGather failed: simulated (NBX_GROUP_RCCL_FAIL=gather)");
    }
    RcclApi* api = rccl_api();
    const ncclDataType_t ty = half ? ncclFloat16 : ncclFloat32;
    const size_t rec = half ? 8 : sizeof(float4);
    auto base = [&](nbx_engine* e) { return half ? static_cast<char*>(e->d_posh) : reinterpret_cast<char*>(e->d_posm); };
    // rank d's share of the collective: one all-gather when the slabs are equal, else (ragged last slab of the reference
    // split) one broadcast per owner, the same sequence on every rank
    auto enqueue = [&](int d) -> int {
        nbx_engine* e = g->eng[d];
        if (n % G == 0) {
            RCCL_TRY(api, api->AllGather(base(e) + rec * (size_t)e->lo, base(e), (size_t)e->slab() * 4, ty, g->comms[d], e->stream));
            return NBX_OK;
        }
        for (int r = 0; r < G; r++) {
            const int lo = g->eng[r]->lo, cnt = g->eng[r]->slab();
            if (cnt == 0) continue;
            RCCL_TRY(api, api->Broadcast(base(e) + rec * (size_t)lo, base(e) + rec * (size_t)lo, (size_t)cnt * 4, ty, r, g->comms[d], e->stream));
        }
        return NBX_OK;
    };
    if (g->workers) {   // one thread per rank: each enqueues on its own communicator, no group call needed
        const std::vector<std::function<int(int)>> phases = {[&](int d) -> int {
            HIP_TRY(hipSetDevice(g->eng[d]->device));
            ProfScope ps(g->eng[d], NBX_K_EXCHANGE);
            return enqueue(d);
        }};
        return g->workers->run(phases);
    }
    std::vector<std::unique_ptr<ProfScope>> scopes;
    for (nbx_engine* e : g->eng) {
        HIP_TRY(hipSetDevice(e->device));
        scopes.emplace_back(new ProfScope(e, NBX_K_EXCHANGE));
    }
    int rc = NBX_OK;
    const ncclResult_t gs = api->GroupStart();
    if (gs != ncclSuccess) rc = fail(NBX_ERR_HIP, "ncclGroupStart failed: %s", api->GetErrorString(gs));
    for (int d = 0; d < G && rc == NBX_OK; d++) rc = enqueue(d);
    if (gs == ncclSuccess) {
        const ncclResult_t ge = api->GroupEnd();   // always closed, also after a failed enqueue
        if (ge != ncclSuccess && rc == NBX_OK) rc = fail(NBX_ERR_HIP, "ncclGroupEnd failed: %s", api->GetErrorString(ge));
    }
    for (size_t d = 0; d < g->eng.size(); d++) {
        (void)hipSetDevice(g->eng[d]->device);
        scopes[d].reset();
    }
    return rc;
}

// one exchange of the slabs of ONE array: half = false: the float4 (x,y,z,m) array; half = true: the half4 source copy
// (ncclFloat16, half the bytes on the wire -- SURVEY.md 8(e), BASELINE config #5).  RCCL unless peer copies were asked
// for; any RCCL failure (library, communicator, collective) switches the group to peer copies for good and the
// exchange is redone that way -- the in-place gather is idempotent on the data.
int group_exchange_array(nbx_group* g, bool half)
{
    const int G = (int)g->eng.size();
    if (!g->copy_exchange) {
        int rc = group_comms(g);
        if (rc == NBX_OK) rc = group_exchange_rccl(g, half);
        if (rc != NBX_OK) {
            const std::string why = g_last_error;
            // nothing of the failed attempt may still be in flight: the communicators are ABORTED first (ranks that did enqueue
            // their share wait for the one that could not; no stream they run on would ever drain), then the streams drained
            group_destroy_comms(g, /*abort=*/true);
            for (nbx_engine* e : g->eng) {
                (void)hipSetDevice(e->device);
                (void)hipStreamSynchronize(e->stream);
            }
            (void)hipGetLastError();
            group_fall_back_to_copy(g, why.c_str());
        }
    }
    if (g->copy_exchange && G > 1) {
        const int rc = group_exchange_copy(g, half);
        if (rc != NBX_OK) return rc;
    }
    // the device arrays of the other slabs just changed: no engine's host mirror of the positions may be trusted
    // (a mirror cached while the fp32 array was stale survived the re-gather)
    if (G > 1)
        for (nbx_engine* e : g->eng) e->host_pos_valid = false;
    g->exchanges++;
    return NBX_OK;
}

// every engine runs the packed fp16-source sweep: the half4 copy is then the ONLY array the next all-pairs step reads
// from other slabs, so it is the array that travels
bool group_wants_half_exchange(const nbx_group* g)
{
    for (const nbx_engine* e : g->eng)
        if (!(e->source_half && e->force_mode == 0)) return false;
    return true;
}

// The per-step exchange. fp32 sources: one all-gather of (x,y,z,m). fp16 sources on every engine: one all-gather of the
// half4 copy instead (each engine refreshed its own slab's slot in step_brute); the fp32 positions of the other slabs go
// stale and are re-gathered only when somebody needs them (get, draw, Barnes-Hut, a bit-exact step).  Mixed settings
// (some engines fp16): fp32 travels and those engines re-pack the slabs they received.
int group_exchange(nbx_group* g, bool allow_half)
{
    const int n = g->eng[0]->n;
    if (n == 0) return NBX_OK;
    int rc;
    if (allow_half && group_wants_half_exchange(g)) {
        rc = group_exchange_array(g, true);
        if (rc != NBX_OK) return rc;
        g->fp32_stale = g->eng.size() > 1;
    } else {
        g->fp32_stale = false;   // everything is about to be current
        rc = group_exchange_array(g, false);
        if (rc != NBX_OK) return rc;
        for (nbx_engine* e : g->eng)
            if (e->source_half && g->eng.size() > 1) {   // stream-ordered after the gather on e's stream
                HIP_TRY(hipSetDevice(e->device));
                rc = refresh_half_sources(e, 0, e->n_pad);
                if (rc != NBX_OK) return rc;
            }
    }
    for (nbx_engine* e : g->eng) e->host_pos_valid = false;
    return NBX_OK;
}

}  // namespace

namespace nbxi {

// bring the fp32 (x,y,z,m) array of every engine up to date after steps that exchanged only the fp16 copy
int group_replicate_fp32(nbx_group* g)
{
    if (!g->fp32_stale) return NBX_OK;
    const int rc = group_exchange_array(g, false);
    if (rc != NBX_OK) return rc;
    g->fp32_stale = false;
    return NBX_OK;
}

}  // namespace nbxi

extern "C" {

int32_t nbx_group_create(nbx_group** out, const int32_t* devices, int32_t count)
{
    if (!out || count < 1) return fail(NBX_ERR_INVALID, "bad group arguments");
    const int present = nbx_device_count();
    nbx_group* g = new (std::nothrow) nbx_group();
    if (!g) return fail(NBX_ERR_ALLOC, "out of memory");
    if (const char* x = std::getenv("NBX_GROUP_EXCHANGE")) g->copy_exchange = std::strcmp(x, "copy") == 0;
    g->exchange_kind = g->copy_exchange ? 1 : 0;
    if (const char* x = std::getenv("NBX_GROUP_RCCL_FAIL"))   // tests: make the RCCL path "fail" to exercise the fallback
        g->rccl_fail_hook = std::strcmp(x, "init") == 0 ? 1 : (std::strcmp(x, "gather") == 0 ? 2 : 0);
    for (int i = 0; i < count; i++) {
        const int dev = devices ? devices[i] : i;
        for (int j = 0; j < i; j++)   // RCCL wants one rank per device; the copy exchange does not care
            if (g->devices[j] == dev && !g->copy_exchange && g->rccl_fail_hook != 1) { nbx_group_destroy(g); return fail(NBX_ERR_INVALID, "device %d listed twice", dev); }
        if (present > 0 && (dev < 0 || dev >= present)) { nbx_group_destroy(g); return fail(NBX_ERR_NO_DEVICE, "no device %d (%d present)", dev, present); }
        nbx_engine* e = nullptr;
        if (nbx_create(&e, dev) != NBX_OK) { nbx_group_destroy(g); return NBX_ERR_ALLOC; }
        e->rank = i;
        e->world = count;
        g->eng.push_back(e);
        g->devices.push_back(dev);
    }
    if (const char* x = std::getenv("NBX_GROUP_ENQUEUE"))
        if (std::strcmp(x, "threads") == 0 && count > 1) g->workers = new (std::nothrow) GroupWorkers(count);
    *out = g;
    return NBX_OK;
}

void nbx_group_destroy(nbx_group* g)
{
    if (!g) return;
    delete g->workers;
    g->workers = nullptr;
    for (nbx_engine* e : g->eng)
        if (e && e->dev_ready) { (void)hipSetDevice(e->device); (void)hipStreamSynchronize(e->stream); }
    group_destroy_comms(g);
    for (hipEvent_t ev : g->ev_ready) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : g->ev_copied) (void)hipEventDestroy(ev);
    for (nbx_engine* e : g->eng) nbx_destroy(e);
    delete g;
}

int32_t nbx_group_size(const nbx_group* g) { return g ? (int32_t)g->eng.size() : NBX_ERR_INVALID; }
nbx_engine* nbx_group_engine(nbx_group* g, int32_t i) { return (g && i >= 0 && i < (int)g->eng.size()) ? g->eng[i] : nullptr; }

int32_t nbx_group_set_option(nbx_group* g, int32_t option, int64_t value)
{
    if (!g) return fail(NBX_ERR_INVALID, "null group");
    for (nbx_engine* e : g->eng) {
        const int rc = nbx_set_option(e, option, value);
        if (rc != NBX_OK) return rc;
    }
    return NBX_OK;
}

int32_t nbx_group_num_particles(const nbx_group* g) { return g ? g->eng[0]->n : NBX_ERR_INVALID; }

int32_t nbx_group_set_particles3(nbx_group* g, int32_t n, const float* px, const float* py, const float* pz, const float* vx,
                                 const float* vy, const float* vz, const float* m)
{
    if (!g) return fail(NBX_ERR_INVALID, "null group");
    for (nbx_engine* e : g->eng) {
        const int rc = nbx_set_particles3(e, n, px, py, pz, vx, vy, vz, m);
        if (rc != NBX_OK) return rc;
    }
    g->fp32_stale = false;
    return NBX_OK;
}

int32_t nbx_group_get_particles3(nbx_group* g, int32_t cap, float* px, float* py, float* pz, float* vx, float* vy, float* vz,
                                 float* m)
{
    if (!g) return fail(NBX_ERR_INVALID, "null group");
    nbx_engine* e0 = g->eng[0];
    if (cap < e0->n) return fail(NBX_ERR_INVALID, "capacity %d < particle count %d", cap, e0->n);
    int rc = group_replicate_fp32(g);
    if (rc != NBX_OK) return rc;
    rc = nbx_get_particles3(e0, cap, px, py, pz, vx, vy, vz, m);   // positions are replicated after the all-gather
    if (rc < 0) return rc;
    for (size_t d = 1; d < g->eng.size(); d++) {                        // velocities live on their owner
        nbx_engine* e = g->eng[d];
        rc = download_velocities(e);
        if (rc != NBX_OK) return rc;
        const size_t bytes = sizeof(float) * (size_t)e->slab();
        if (vx) std::memcpy(vx + e->lo, e->host.vx.data() + e->lo, bytes);
        if (vy) std::memcpy(vy + e->lo, e->host.vy.data() + e->lo, bytes);
        if (vz) std::memcpy(vz + e->lo, e->host.vz.data() + e->lo, bytes);
    }
    return e0->n;
}

int32_t nbx_group_step_brute_force(nbx_group* g, float dt)
{
    if (!g) return fail(NBX_ERR_INVALID, "null group");
    if (g->fp32_stale && !group_wants_half_exchange(g)) {   // e.g. a bit-exact step after fp16-source steps: it reads fp32 sources
        const int rc = group_replicate_fp32(g);
        if (rc != NBX_OK) return rc;
    }
    // asynchronous: every device works on its slab concurrently (enqueued by one thread in turn, or by one thread per device)
    const std::vector<std::function<int(int)>> phases = {[&](int d) -> int { return step_brute(g->eng[d], dt); }};
    const int rc = run_phases(g, phases);
    if (rc != NBX_OK) return rc;
    return group_exchange(g, true);
}

int32_t nbx_group_step_barnes_hut(nbx_group* g, float theta, float dt, int32_t nthreads)
{
    if (!g) return fail(NBX_ERR_INVALID, "null group");
    if (theta == 0.0f) return nbx_group_step_brute_force(g, dt);   // nbody.rs:197-200
    if (nthreads <= 0) return fail(NBX_ERR_INVALID, "nthreads must be >= 1");
    // tree replica per device, built once per step and shared; each device evaluates its slab
    int rc = group_replicate_fp32(g);   // the tree is built from, and the walk reads, fp32 positions
    if (rc != NBX_OK) return rc;
    rc = step_bh_group(g->eng.data(), (int)g->eng.size(), theta, dt);
    if (rc != NBX_OK) return rc;
    return group_exchange(g, false);
}

int32_t nbx_group_synchronize(nbx_group* g)
{
    if (!g) return fail(NBX_ERR_INVALID, "null group");
    for (nbx_engine* e : g->eng) {
        const int rc = nbx_synchronize(e);
        if (rc != NBX_OK) return rc;
    }
    return NBX_OK;
}

int32_t nbx_group_draw(nbx_group* g, int32_t w, int32_t h, uint32_t* fb)
{
    if (!g || !fb || w <= 0 || h <= 0) return fail(NBX_ERR_INVALID, "bad draw arguments");
    const int n = g->eng[0]->n;
    std::vector<float> px(n), py(n), vx(n), vy(n);
    const int rc = nbx_group_get_particles3(g, n, px.data(), py.data(), nullptr, vx.data(), vy.data(), nullptr, nullptr);
    if (rc < 0) return rc;
    nbx::draw_particles(px.data(), py.data(), vx.data(), vy.data(), n, w, h, fb);
    return NBX_OK;
}

int32_t nbx_group_exchanges(const nbx_group* g) { return g ? g->exchanges : NBX_ERR_INVALID; }

int64_t nbx_group_info(const nbx_group* g, int32_t what)
{
    if (!g) return NBX_ERR_INVALID;
    switch (what) {
        case NBX_GROUP_INFO_EXCHANGE: return g->exchange_kind;
        case NBX_GROUP_INFO_RCCL_RANKS: return g->rccl_ranks;
        case NBX_GROUP_INFO_ENQUEUE_THREADS: return g->workers ? (int64_t)g->eng.size() : 0;
        case NBX_GROUP_INFO_FP32_STALE: return g->fp32_stale ? 1 : 0;
        default: return NBX_ERR_INVALID;
    }
}

const char* nbx_group_exchange_note(const nbx_group* g) { return g ? g->exchange_note.c_str() : ""; }

int32_t nbx_group_set_enqueue_threads(nbx_group* g, int32_t on)
{
    if (!g) return fail(NBX_ERR_INVALID, "null group");
    const int rc = nbx_group_synchronize(g);
    if (rc != NBX_OK) return rc;
    if (on && !g->workers && g->eng.size() > 1) {
        g->workers = new (std::nothrow) GroupWorkers((int)g->eng.size());
        if (!g->workers) return fail(NBX_ERR_ALLOC, "out of memory");
    } else if (!on && g->workers) {
        delete g->workers;
        g->workers = nullptr;
    }
    return NBX_OK;
}

}  // extern "C"
