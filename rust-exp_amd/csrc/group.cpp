// group.cpp -- single-process multi-GPU group (nbx_group_*), RCCL resolved with dlopen.
#include <dlfcn.h>

#include <memory>

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


namespace {

int group_comms(nbx_group* g)
{
    if (!g->comms.empty()) return NBX_OK;
    RcclApi* api = rccl_api();
    if (!api) return fail(NBX_ERR_HIP, "librccl.so could not be loaded: %s", dlerror());
    g->comms.resize(g->eng.size());
    RCCL_TRY(api, api->CommInitAll(g->comms.data(), (int)g->eng.size(), g->devices.data()));
    return NBX_OK;
}

// The same exchange without RCCL (NBX_GROUP_EXCHANGE=copy): every engine pulls every other engine's slab with
// hipMemcpyPeerAsync on its own stream, ordered by events -- a slab is read only after its owner's kick-drift
// (ev_ready), and nobody starts the next step before everyone has pulled from it (ev_copied).  Works with or
// without peer access, needs no communicator, and lets several engines share one device (how the group logic is
// tested on a single-GPU box).
int group_exchange_copy(nbx_group* g, bool half)
{
    const int G = (int)g->eng.size();
    const size_t rec = half ? 8 : sizeof(float4);   // half4 source copy or float4 (x,y,z,m)
    if (g->ev_ready.empty()) {
        g->ev_ready.resize(G);
        g->ev_copied.resize(G);
        for (int d = 0; d < G; d++) {
            HIP_TRY(hipSetDevice(g->eng[d]->device));
            HIP_TRY(hipEventCreateWithFlags(&g->ev_ready[d], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&g->ev_copied[d], hipEventDisableTiming));
        }
    }
    for (int d = 0; d < G; d++) {
        HIP_TRY(hipSetDevice(g->eng[d]->device));
        HIP_TRY(hipEventRecord(g->ev_ready[d], g->eng[d]->stream));
    }
    std::vector<std::unique_ptr<ProfScope>> scopes;   // per engine: start after its own kernels, stop when it may go on
    for (int d = 0; d < G; d++) {
        nbx_engine* dst = g->eng[d];
        HIP_TRY(hipSetDevice(dst->device));
        scopes.emplace_back(new ProfScope(dst, NBX_K_EXCHANGE));
        for (int s = 0; s < G; s++) {
            if (s == d) continue;
            nbx_engine* src = g->eng[s];
            if (src->slab() == 0) continue;
            HIP_TRY(hipStreamWaitEvent(dst->stream, g->ev_ready[s], 0));
            char* to = (half ? static_cast<char*>(dst->d_posh) : reinterpret_cast<char*>(dst->d_posm)) + rec * (size_t)src->lo;
            const char* from = (half ? static_cast<const char*>(src->d_posh) : reinterpret_cast<const char*>(src->d_posm)) + rec * (size_t)src->lo;
            HIP_TRY(hipMemcpyPeerAsync(to, dst->device, from, src->device, rec * (size_t)src->slab(), dst->stream));
        }
        HIP_TRY(hipEventRecord(g->ev_copied[d], dst->stream));
    }
    for (int d = 0; d < G; d++) {
        HIP_TRY(hipSetDevice(g->eng[d]->device));
        for (int s = 0; s < G; s++)
            if (s != d) HIP_TRY(hipStreamWaitEvent(g->eng[d]->stream, g->ev_copied[s], 0));
        scopes[(size_t)d].reset();   // stop event on d's stream, with d's device current
    }
    return NBX_OK;
}

// one all-gather of the slabs of ONE array, in place (sendbuff = recvbuff + lo), on the same streams as the kernels:
// half = false: the float4 (x,y,z,m) array; half = true: the half4 source copy (ncclFloat16, half the bytes on the wire --
// SURVEY.md 8(e), BASELINE config #5)
int group_exchange_array(nbx_group* g, bool half)
{
    const int G = (int)g->eng.size();
    const int n = g->eng[0]->n;
    if (g->copy_exchange) {
        if (G > 1) {
            const int rc = group_exchange_copy(g, half);
            if (rc != NBX_OK) return rc;
        }
        g->exchanges++;
        return NBX_OK;
    }
    int rc = group_comms(g);
    if (rc != NBX_OK) return rc;
    RcclApi* api = rccl_api();
    const ncclDataType_t ty = half ? ncclFloat16 : ncclFloat32;
    const size_t rec = half ? 8 : sizeof(float4);
    auto base = [&](nbx_engine* e) { return half ? static_cast<char*>(e->d_posh) : reinterpret_cast<char*>(e->d_posm); };
    std::vector<std::unique_ptr<ProfScope>> scopes;
    for (nbx_engine* e : g->eng) {
        HIP_TRY(hipSetDevice(e->device));
        scopes.emplace_back(new ProfScope(e, NBX_K_EXCHANGE));
    }
    RCCL_TRY(api, api->GroupStart());
    if (n % G == 0) {
        for (int d = 0; d < G; d++) {
            nbx_engine* e = g->eng[d];
            RCCL_TRY(api, api->AllGather(base(e) + rec * (size_t)e->lo, base(e), (size_t)e->slab() * 4, ty, g->comms[d], e->stream));
        }
    } else {   // ragged last slab (reference split): one broadcast per owner
        for (int r = 0; r < G; r++) {
            const int lo = g->eng[r]->lo, cnt = g->eng[r]->slab();
            if (cnt == 0) continue;
            for (int d = 0; d < G; d++) {
                nbx_engine* e = g->eng[d];
                RCCL_TRY(api, api->Broadcast(base(e) + rec * (size_t)lo, base(e) + rec * (size_t)lo, (size_t)cnt * 4, ty, r, g->comms[d], e->stream));
            }
        }
    }
    RCCL_TRY(api, api->GroupEnd());
    for (size_t d = 0; d < g->eng.size(); d++) {
        HIP_TRY(hipSetDevice(g->eng[d]->device));
        scopes[d].reset();
    }
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

// bring the fp32 (x,y,z,m) array of every engine up to date after steps that exchanged only the fp16 copy
int group_replicate_fp32(nbx_group* g)
{
    if (!g->fp32_stale) return NBX_OK;
    const int rc = group_exchange_array(g, false);
    if (rc != NBX_OK) return rc;
    g->fp32_stale = false;
    return NBX_OK;
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

extern "C" {

int32_t nbx_group_create(nbx_group** out, const int32_t* devices, int32_t count)
{
    if (!out || count < 1) return fail(NBX_ERR_INVALID, "bad group arguments");
    const int present = nbx_device_count();
    nbx_group* g = new (std::nothrow) nbx_group();
    if (!g) return fail(NBX_ERR_ALLOC, "out of memory");
    if (const char* x = std::getenv("NBX_GROUP_EXCHANGE")) g->copy_exchange = std::strcmp(x, "copy") == 0;
    for (int i = 0; i < count; i++) {
        const int dev = devices ? devices[i] : i;
        for (int j = 0; j < i; j++)   // RCCL wants one rank per device; the copy exchange does not care
            if (g->devices[j] == dev && !g->copy_exchange) { nbx_group_destroy(g); return fail(NBX_ERR_INVALID, "device %d listed twice", dev); }
        if (present > 0 && (dev < 0 || dev >= present)) { nbx_group_destroy(g); return fail(NBX_ERR_NO_DEVICE, "no device %d (%d present)", dev, present); }
        nbx_engine* e = nullptr;
        if (nbx_create(&e, dev) != NBX_OK) { nbx_group_destroy(g); return NBX_ERR_ALLOC; }
        e->rank = i;
        e->world = count;
        g->eng.push_back(e);
        g->devices.push_back(dev);
    }
    *out = g;
    return NBX_OK;
}

void nbx_group_destroy(nbx_group* g)
{
    if (!g) return;
    for (nbx_engine* e : g->eng)
        if (e && e->dev_ready) { (void)hipSetDevice(e->device); (void)hipStreamSynchronize(e->stream); }
    if (!g->comms.empty())
        if (RcclApi* api = rccl_api())
            for (ncclComm_t c : g->comms) (void)api->CommDestroy(c);
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
    for (nbx_engine* e : g->eng) {   // asynchronous: every device works on its slab concurrently
        const int rc = step_brute(e, dt);
        if (rc != NBX_OK) return rc;
    }
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

}  // extern "C"
