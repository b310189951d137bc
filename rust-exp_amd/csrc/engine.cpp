// engine.cpp -- state owner and C ABI of libnbody_mi355x.so (see include/nbody_mi355x.h).
//
// Plays the role the north_star gives to "Rust host code": it owns the particle arrays (a host
// SoA mirror + the device-resident float4 arrays), implements the reference's six extern "C"
// entry points (rs-src/nbody.rs:34-35,:39-40,:73-74,:106-107,:186-187,:482-483) on top of a
// handle API, and drives the gfx950 kernels.  No CPU fallback: steps need a device.
//
// HBM layout (all 16-B records, coalesced 16 B/lane):
//   posm  float4[n_pad]        (x, y, z, m) for ALL bodies; n_pad = n rounded up to 256, padding
//                              entries are (0,0,0,0): zero mass => exact zero contribution.
//   vel   float4[slab]         (vx, vy, vz, 0) for this engine's slab of targets only.
//   acc   float4[S][stride]    per-source-split partial accelerations of the fast kernel.
//   f2    float2[slab]         forces (strict / Barnes-Hut paths).
//   nodes BhNode[n_nodes]      flattened quadtree, rebuilt on the host every Barnes-Hut step.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <vector>

#include "../../include/nbody_mi355x.h"
#include "host_ops.h"
#include "kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(NBX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct ProfRec {
    int kernel;
    hipEvent_t start, stop;
};

}  // namespace

struct nbx_engine {
    int device = 0;
    bool dev_ready = false;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cu_count = 256;

    nbx::HostState host;
    bool host_pos_valid = true, host_vel_valid = true;  // host mirror current?
    bool dev_valid = false;                             // device arrays current?
    int n = 0, n_pad = 0;
    int rank = 0, world = 1, lo = 0, hi = 0;

    float4* d_posm = nullptr;
    bool posm_external = false;
    size_t posm_cap = 0;  // records
    float4* d_vel = nullptr;
    size_t vel_cap = 0;
    float4* d_acc = nullptr;
    size_t acc_cap = 0;
    float2* d_f2 = nullptr;
    size_t f2_cap = 0;
    float4* d_out4 = nullptr;
    size_t out4_cap = 0;
    nbx::BhNode* d_nodes = nullptr;
    size_t nodes_cap = 0;
    unsigned* d_guard = nullptr;   // max|coord| word for the batched-reciprocal kernel
    size_t guard_cap = 0;
    void* d_tree_ws = nullptr;     // device tree build workspace (NBX_OPT_BH_TREE = 1)
    size_t tree_ws_bytes = 0;
    int* h_counters = nullptr;     // pinned: per-level node counters of the device build
    const unsigned* d_perm = nullptr;   // spatial body order produced by the device build
    int bh_tree_device = 0;
    int bh_wave = 1;               // wave-uniform traversal when a spatial body order is available
    int bh_fallbacks = 0;          // device builds that fell back to the host (node pool exhausted)
    void* d_counts = nullptr;      // device draw: uint2 hit counters per pixel
    size_t counts_cap = 0;         // pixels
    unsigned* d_fb = nullptr;
    size_t fb_cap = 0;
    int draw_device = 0;
    void* d_posh = nullptr;        // half4 (x,y,z,m) source copy (NBX_OPT_SOURCE_PRECISION = 16)
    bool posh_external = false;
    size_t posh_cap = 0;           // records
    int source_half = 0;

    // options
    int force_mode = 0, jsplit = 0, bpt = 0, dim_opt = 0, profile = 0, variant = -1;
    bool any_z = false;

    nbx::Rng rng{0};
    bool seeded = false;

    nbx::QuadTree tree;
    nbx::QuadTree::FlatPlan plan;
    std::vector<nbx::BhNode> flat_small;
    nbx::BhNode* h_nodes = nullptr;   // pinned host staging of the flattened tree
    size_t h_nodes_cap = 0;
    size_t n_flat = 0;
    float4* h_stage = nullptr;        // pinned host staging for position downloads
    size_t h_stage_cap = 0;

    std::vector<ProfRec> prof;
    nbx::ForceLaunch last{0, 0, 0, 0, 0, 0};
    double host_ms[4] = {0, 0, 0, 0};  // Barnes-Hut host phases: download, build, flatten, upload (cumulative)
    int host_steps = 0;

    int slab() const { return hi - lo; }
};

namespace {

using nbx::kTile;

void compute_slab(nbx_engine* e)
{
    // the reference's static split: range = N / T, the last worker takes the remainder (nbody.rs:426-428)
    const int range = e->n / e->world;
    e->lo = range * e->rank;
    e->hi = (e->rank == e->world - 1) ? e->n : range * (e->rank + 1);
}

int ensure_device(nbx_engine* e)
{
    if (e->dev_ready) {
        HIP_TRY(hipSetDevice(e->device));
        return NBX_OK;
    }
    int count = 0;
    hipError_t err = hipGetDeviceCount(&count);
    if (err != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(NBX_ERR_NO_DEVICE,
                    "no HIP device available (%s); the MI355X engine has no CPU fallback by design",
                    err == hipSuccess ? "device count 0" : hipGetErrorString(err));
    }
    if (e->device < 0 || e->device >= count) return fail(NBX_ERR_NO_DEVICE, "device %d out of range (%d present)", e->device, count);
    HIP_TRY(hipSetDevice(e->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, e->device));
    e->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (!e->stream) {
        HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        e->own_stream = true;
    }
    e->dev_ready = true;
    return NBX_OK;
}

template <typename T>
int grow(T** ptr, size_t* cap, size_t need)
{
    if (need <= *cap && *ptr) return NBX_OK;
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    const size_t want = std::max<size_t>(need, 256);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(ptr), want * sizeof(T)));
    *cap = want;
    return NBX_OK;
}

// (re)build the fp16 source copy for records [first, first+count) from the fp32 array
int refresh_half_sources(nbx_engine* e, int first, int count)
{
    if (!e->posh_external && (size_t)e->n_pad > e->posh_cap) {
        if (e->d_posh) HIP_TRY(hipFree(e->d_posh));
        e->d_posh = nullptr;
        e->posh_cap = 0;
        HIP_TRY(hipMalloc(&e->d_posh, (size_t)e->n_pad * 8));
        e->posh_cap = (size_t)e->n_pad;
        first = 0;
        count = e->n_pad;
    }
    if (e->posh_external && (size_t)e->n_pad > e->posh_cap) return fail(NBX_ERR_STATE, "bound half-source buffer too small");
    HIP_TRY(nbx::launch_pack_half(e->d_posm, e->d_posh, first, count, e->stream));
    return NBX_OK;
}

int upload(nbx_engine* e)
{
    int rc = ensure_device(e);
    if (rc != NBX_OK) return rc;
    if (e->dev_valid) return NBX_OK;
    if (!(e->host_pos_valid && e->host_vel_valid)) return fail(NBX_ERR_STATE, "no valid state to upload");
    const int n = e->n;
    e->n_pad = ((n + kTile - 1) / kTile) * kTile;
    if (e->n_pad == 0) e->n_pad = kTile;
    if (e->posm_external) {
        if ((size_t)e->n_pad > e->posm_cap) return fail(NBX_ERR_STATE, "bound positions buffer too small");
    } else {
        rc = grow(&e->d_posm, &e->posm_cap, (size_t)e->n_pad);
        if (rc != NBX_OK) return rc;
    }
    const int slab = e->slab();
    rc = grow(&e->d_vel, &e->vel_cap, (size_t)std::max(slab, 1));
    if (rc != NBX_OK) return rc;
    std::vector<float4> tmp((size_t)e->n_pad, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int i = 0; i < n; i++) tmp[i] = make_float4(e->host.px[i], e->host.py[i], e->host.pz[i], e->host.m[i]);
    HIP_TRY(hipMemcpyAsync(e->d_posm, tmp.data(), sizeof(float4) * (size_t)e->n_pad, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (slab > 0) {
        std::vector<float4> tv((size_t)slab);
        for (int i = 0; i < slab; i++)
            tv[i] = make_float4(e->host.vx[e->lo + i], e->host.vy[e->lo + i], e->host.vz[e->lo + i], 0.f);
        HIP_TRY(hipMemcpyAsync(e->d_vel, tv.data(), sizeof(float4) * (size_t)slab, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    e->dev_valid = true;
    if (e->source_half) {
        rc = refresh_half_sources(e, 0, e->n_pad);
        if (rc != NBX_OK) return rc;
    }
    return NBX_OK;
}

int download_positions(nbx_engine* e)
{
    if (e->host_pos_valid) return NBX_OK;
    int rc = ensure_device(e);
    if (rc != NBX_OK) return rc;
    if ((size_t)e->n > e->h_stage_cap) {
        if (e->h_stage) HIP_TRY(hipHostFree(e->h_stage));
        e->h_stage = nullptr;
        e->h_stage_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_stage), sizeof(float4) * (size_t)std::max(e->n, 256), hipHostMallocDefault));
        e->h_stage_cap = (size_t)std::max(e->n, 256);
    }
    float4* tmp = e->h_stage;
    HIP_TRY(hipMemcpyAsync(tmp, e->d_posm, sizeof(float4) * (size_t)e->n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int i = 0; i < e->n; i++) {
        e->host.px[i] = tmp[i].x; e->host.py[i] = tmp[i].y; e->host.pz[i] = tmp[i].z;
    }
    e->host_pos_valid = true;
    return NBX_OK;
}

int download_velocities(nbx_engine* e)
{
    if (e->host_vel_valid) return NBX_OK;
    int rc = ensure_device(e);
    if (rc != NBX_OK) return rc;
    const int slab = e->slab();
    if (slab > 0) {
        std::vector<float4> tmp((size_t)slab);
        HIP_TRY(hipMemcpyAsync(tmp.data(), e->d_vel, sizeof(float4) * (size_t)slab, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        for (int i = 0; i < slab; i++) {
            e->host.vx[e->lo + i] = tmp[i].x; e->host.vy[e->lo + i] = tmp[i].y; e->host.vz[e->lo + i] = tmp[i].z;
        }
    }
    e->host_vel_valid = true;
    return NBX_OK;
}

struct ProfScope {
    nbx_engine* e;
    int idx = -1;
    ProfScope(nbx_engine* eng, int kernel) : e(eng)
    {
        if (!e->profile) return;
        ProfRec r{kernel, nullptr, nullptr};
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
        (void)hipEventRecord(r.start, e->stream);
        e->prof.push_back(r);
        idx = (int)e->prof.size() - 1;
    }
    ~ProfScope()
    {
        if (idx >= 0) (void)hipEventRecord(e->prof[idx].stop, e->stream);
    }
};

void choose_launch(const nbx_engine* e, int n_targets, int tiles_total, int* variant, int* bpt, int* jsplit, int* dim)
{
    // Defaults from the measured launch-shape sweeps (profiles/r01_shapes_sweep*.txt):
    //  * kernel: scalar-cache sources + packed math (variant 5) once the source array has >= 32768 bodies
    //    (no LDS traffic -> +2.4 % clock under the power cap, +3.5 % throughput); LDS tiles (variant 1) below.
    //  * register blocking 4 (two packed pairs) when a GPU owns >= 32768 targets, else 2.
    //  * source split S = smallest power of two giving >= 32 workgroups per CU (64 for variant 5 with < 131072
    //    targets per GPU), capped at 64 and at half the tile count.
    *dim = e->dim_opt ? e->dim_opt : (e->any_z ? 3 : 2);
    int v = e->variant;
    if (v < 0) v = (tiles_total * kTile >= 32768) ? 5 : 1;
    *variant = v;
    int b = e->bpt ? e->bpt : (n_targets >= 32768 ? 4 : 2);
    if (b != 1 && b != 2 && b != 4) b = 2;
    *bpt = b;
    int s = e->jsplit;
    if (s <= 0) {
        const int iblocks = (n_targets + kTile * b - 1) / (kTile * b);
        // 64 workgroups per CU only where targets are scarce (sharded shapes: tail effect); 32 otherwise --
        // same speed at N = 262144 on one GPU and half the partial-slab traffic
        const int want = e->cu_count * ((v == 5 && n_targets < 131072) ? 64 : 32);
        s = 1;
        while (iblocks * s < want && s < 64) s *= 2;
        s = std::min(s, std::max(1, tiles_total / 2));
    }
    s = std::max(1, std::min(s, tiles_total));
    *jsplit = s;
}

int launch_forces_fast(nbx_engine* e)
{
    const int slab = e->slab();
    const int tiles_total = e->n_pad / kTile;
    int variant, bpt, jsplit, dim;
    choose_launch(e, slab, tiles_total, &variant, &bpt, &jsplit, &dim);
    const int stride = ((slab + kTile - 1) / kTile) * kTile;
    int rc = grow(&e->d_acc, &e->acc_cap, (size_t)jsplit * (size_t)std::max(stride, kTile));
    if (rc != NBX_OK) return rc;
    if (variant == 4) {
        rc = grow(&e->d_guard, &e->guard_cap, 1);
        if (rc != NBX_OK) return rc;
    }
    if (e->source_half) {
        ProfScope ps(e, NBX_K_FORCE);
        HIP_TRY(nbx::launch_force_tile_half(e->d_posm, e->d_posh, e->lo, slab, tiles_total, jsplit, bpt, dim, e->d_acc,
                                            stride, e->stream, &e->last));
        return NBX_OK;
    }
    {
        ProfScope ps(e, NBX_K_FORCE);
        HIP_TRY(nbx::launch_force_tile(e->d_posm, e->lo, slab, tiles_total, jsplit, bpt, dim, variant, e->d_acc, stride,
                                       variant == 4 ? e->d_guard : nullptr, e->stream, &e->last));
    }
    return NBX_OK;
}

int step_brute(nbx_engine* e, float dt)
{
    int rc = upload(e);
    if (rc != NBX_OK) return rc;
    const int slab = e->slab();
    if (e->n == 0 || slab == 0) return NBX_OK;
    if (e->force_mode == 1) {
        rc = grow(&e->d_f2, &e->f2_cap, (size_t)slab);
        if (rc != NBX_OK) return rc;
        {
            ProfScope ps(e, NBX_K_FORCE);
            HIP_TRY(nbx::launch_force_strict(e->d_posm, e->n, e->lo, slab, e->d_f2, e->stream));
            e->last = nbx::ForceLaunch{(slab + kTile - 1) / kTile, kTile, 1, 1, 2, -1};
        }
        {
            ProfScope ps(e, NBX_K_INTEGRATE);
            HIP_TRY(nbx::launch_integrate_f2(e->d_posm, e->lo, slab, e->d_vel, e->d_f2, dt, 0, 0, e->stream));
        }
        if (e->source_half) {   // keep the fp16 source copy coherent even when a bit-exact step moved the bodies
            rc = refresh_half_sources(e, e->lo, slab);
            if (rc != NBX_OK) return rc;
        }
    } else {
        rc = launch_forces_fast(e);
        if (rc != NBX_OK) return rc;
        const int stride = ((slab + kTile - 1) / kTile) * kTile;
        ProfScope ps(e, NBX_K_INTEGRATE);
        HIP_TRY(nbx::launch_integrate(e->d_posm, e->lo, slab, e->d_vel, e->d_acc, e->last.jsplit, stride, dt,
                                      e->stream));
        if (e->source_half) {   // refresh this slab's slot of the fp16 source copy (the all-gather send slot)
            rc = refresh_half_sources(e, e->lo, slab);
            if (rc != NBX_OK) return rc;
        }
    }
    e->host_pos_valid = false;
    e->host_vel_valid = false;
    return NBX_OK;
}

// host tree (reference-faithful) -> flatten -> device
int build_and_upload_tree(nbx_engine* e)
{
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clk::now();
    int rc = download_positions(e);
    if (rc != NBX_OK) return rc;
    const auto t1 = clk::now();
    rc = e->tree.build(e->host.px.data(), e->host.py.data(), e->host.m.data(), e->n);
    if (rc == NBX_ERR_TREE_DEPTH) return fail(rc, "quadtree depth > 50 (the reference panics here, nbody.rs:230-232)");
    if (rc != NBX_OK) return fail(rc, "quadtree build hit a reference assert (nbody.rs:267/:293/:304)");
    const auto t2 = clk::now();
    const bool big = e->tree.forest;
    size_t count;
    if (big) {
        count = e->tree.flatten_prepare(e->plan);
    } else {
        e->tree.flatten(e->flat_small);
        count = e->flat_small.size();
    }
    if (count > e->h_nodes_cap) {
        if (e->h_nodes) HIP_TRY(hipHostFree(e->h_nodes));
        e->h_nodes = nullptr;
        e->h_nodes_cap = 0;
        const size_t want = std::max<size_t>(count + count / 4, 1024);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_nodes), sizeof(nbx::BhNode) * want, hipHostMallocDefault));
        e->h_nodes_cap = want;
    }
    if (big)
        e->tree.flatten_write(e->plan, e->h_nodes);
    else if (count)
        std::memcpy(e->h_nodes, e->flat_small.data(), sizeof(nbx::BhNode) * count);
    e->n_flat = count;
    const auto t3 = clk::now();
    rc = grow(&e->d_nodes, &e->nodes_cap, std::max<size_t>(count, 1));
    if (rc != NBX_OK) return rc;
    if (count) {
        HIP_TRY(hipMemcpyAsync(e->d_nodes, e->h_nodes, sizeof(nbx::BhNode) * count, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));  // the staging buffer is rewritten next step
    }
    const auto t4 = clk::now();
    e->host_ms[0] += ms(t0, t1); e->host_ms[1] += ms(t1, t2); e->host_ms[2] += ms(t2, t3); e->host_ms[3] += ms(t3, t4);
    e->host_steps++;
    return NBX_OK;
}

// quadtree on the device (bh_build.hip); falls back to the host build when the node pool overflows
int build_tree_on_device(nbx_engine* e, bool* done)
{
    using clk = std::chrono::steady_clock;
    *done = false;
    const auto t0 = clk::now();
    const int node_cap = 4 * e->n + 1024;
    size_t sort_tmp = 0;
    const size_t need = nbx::device_tree_workspace_bytes(e->n, node_cap, &sort_tmp);
    if (need > e->tree_ws_bytes) {
        if (e->d_tree_ws) HIP_TRY(hipFree(e->d_tree_ws));
        e->d_tree_ws = nullptr;
        e->tree_ws_bytes = 0;
        HIP_TRY(hipMalloc(&e->d_tree_ws, need));
        e->tree_ws_bytes = need;
    }
    if (!e->h_counters) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_counters), 64, hipHostMallocDefault));
    int rc = grow(&e->d_nodes, &e->nodes_cap, (size_t)node_cap);
    if (rc != NBX_OK) return rc;
    int n_nodes = 0, status = 0;
    HIP_TRY(nbx::device_tree_build(e->d_posm, e->n, e->d_tree_ws, e->tree_ws_bytes, node_cap, e->d_nodes, e->h_counters,
                                   &n_nodes, &e->d_perm, &status, e->stream));
    if (status != 0) {
        e->bh_fallbacks++;
        e->d_perm = nullptr;
        return NBX_OK;   // caller takes the host path
    }
    e->n_flat = (size_t)n_nodes;
    e->host_ms[1] += std::chrono::duration<double, std::milli>(clk::now() - t0).count();
    e->host_steps++;
    *done = true;
    return NBX_OK;
}

int step_bh(nbx_engine* e, float theta, float dt)
{
    int rc = upload(e);
    if (rc != NBX_OK) return rc;
    const int slab = e->slab();
    if (e->n == 0) return NBX_OK;
    bool on_device = false;
    if (e->bh_tree_device && e->force_mode == 0) {
        rc = build_tree_on_device(e, &on_device);
        if (rc != NBX_OK) return rc;
    }
    if (!on_device) {
        rc = build_and_upload_tree(e);
        if (rc != NBX_OK) return rc;
    }
    if (slab == 0) return NBX_OK;
    rc = grow(&e->d_f2, &e->f2_cap, (size_t)slab);
    if (rc != NBX_OK) return rc;
    {
        ProfScope ps(e, NBX_K_BH_EVAL);
        const unsigned* perm = (on_device && e->world == 1) ? e->d_perm : nullptr;
        HIP_TRY(nbx::launch_bh_eval(e->d_posm, e->lo, slab, e->d_nodes, (int)e->n_flat, theta,
                                    (e->force_mode == 0 && perm && e->bh_wave) ? 2 : e->force_mode, e->d_f2, e->stream, perm));
    }
    {
        ProfScope ps(e, NBX_K_INTEGRATE);
        HIP_TRY(nbx::launch_integrate_f2(e->d_posm, e->lo, slab, e->d_vel, e->d_f2, dt, e->force_mode == 0 ? 1 : 0, 1,
                                         e->stream));
    }
    if (e->source_half) {
        rc = refresh_half_sources(e, e->lo, slab);
        if (rc != NBX_OK) return rc;
    }
    e->host_pos_valid = false;
    e->host_vel_valid = false;
    return NBX_OK;
}

void free_device(nbx_engine* e)
{
    if (!e->dev_ready) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto& r : e->prof) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    e->prof.clear();
    if (e->d_posm && !e->posm_external) (void)hipFree(e->d_posm);
    if (e->d_vel) (void)hipFree(e->d_vel);
    if (e->d_acc) (void)hipFree(e->d_acc);
    if (e->d_f2) (void)hipFree(e->d_f2);
    if (e->d_out4) (void)hipFree(e->d_out4);
    if (e->d_nodes) (void)hipFree(e->d_nodes);
    if (e->d_guard) (void)hipFree(e->d_guard);
    if (e->d_tree_ws) (void)hipFree(e->d_tree_ws);
    if (e->h_counters) (void)hipHostFree(e->h_counters);
    if (e->d_counts) (void)hipFree(e->d_counts);
    if (e->d_fb) (void)hipFree(e->d_fb);
    if (e->d_posh && !e->posh_external) (void)hipFree(e->d_posh);
    if (e->h_nodes) (void)hipHostFree(e->h_nodes);
    if (e->h_stage) (void)hipHostFree(e->h_stage);
    if (e->stream && e->own_stream) (void)hipStreamDestroy(e->stream);
}

uint64_t entropy_seed()
{
    std::random_device rd;
    return ((uint64_t)rd() << 32) ^ (uint64_t)rd();
}

void after_host_state_change(nbx_engine* e)
{
    e->n = e->host.n();
    compute_slab(e);
    e->host_pos_valid = e->host_vel_valid = true;
    e->dev_valid = false;
    e->any_z = false;
    for (int i = 0; i < e->n && !e->any_z; i++)
        if (e->host.pz[i] != 0.0f || e->host.vz[i] != 0.0f) e->any_z = true;
}

}  // namespace

// =============================================================================================
// Level 2
// =============================================================================================
extern "C" {

const char* nbx_last_error(void) { return g_last_error.c_str(); }
const char* nbx_version(void) { return "nbody_mi355x 0.1 (gfx950)"; }

int32_t nbx_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

int32_t nbx_device_info_get(int32_t device, nbx_device_info* out)
{
    if (!out) return fail(NBX_ERR_INVALID, "null out");
    if (device < 0 || device >= nbx_device_count()) return fail(NBX_ERR_NO_DEVICE, "no such device %d", device);
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    std::memset(out, 0, sizeof *out);
    // some ROCm stacks leave the marketing name empty; fall back to the architecture string
    std::snprintf(out->name, sizeof out->name, "%s", p.name[0] ? p.name : "AMD Instinct (gfx950)");
    std::snprintf(out->arch, sizeof out->arch, "%s", p.gcnArchName);
    out->compute_units = p.multiProcessorCount;
    out->clock_khz = p.clockRate;
    out->wavefront_size = p.warpSize;
    out->lds_bytes_per_cu = (int32_t)p.maxSharedMemoryPerMultiProcessor;
    out->peak_fp32_flops = (double)p.multiProcessorCount * (double)p.clockRate * 1e3 * 256.0;
    out->hbm_bytes = (uint64_t)p.totalGlobalMem;
    return NBX_OK;
}

int32_t nbx_create(nbx_engine** out, int32_t device)
{
    if (!out) return fail(NBX_ERR_INVALID, "null out");
    nbx_engine* e = new (std::nothrow) nbx_engine();
    if (!e) return fail(NBX_ERR_ALLOC, "out of memory");
    e->device = device;
    *out = e;
    return NBX_OK;
}

void nbx_destroy(nbx_engine* e)
{
    if (!e) return;
    free_device(e);
    delete e;
}

int32_t nbx_set_option(nbx_engine* e, int32_t option, int64_t value)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    switch (option) {
        case NBX_OPT_FORCE_MODE:
            if (value != 0 && value != 1) return fail(NBX_ERR_INVALID, "force mode must be 0 (fast) or 1 (strict)");
            e->force_mode = (int)value;
            return NBX_OK;
        case NBX_OPT_JSPLIT:
            if (value < 0 || value > 4096) return fail(NBX_ERR_INVALID, "jsplit out of range");
            e->jsplit = (int)value;
            return NBX_OK;
        case NBX_OPT_BODIES_PER_THREAD:
            if (value != 0 && value != 1 && value != 2 && value != 4) return fail(NBX_ERR_INVALID, "bodies/thread must be 0,1,2,4");
            e->bpt = (int)value;
            return NBX_OK;
        case NBX_OPT_DIM:
            if (value != 0 && value != 2 && value != 3) return fail(NBX_ERR_INVALID, "dim must be 0,2,3");
            e->dim_opt = (int)value;
            return NBX_OK;
        case NBX_OPT_PROFILE:
            e->profile = value ? 1 : 0;
            return NBX_OK;
        case NBX_OPT_KERNEL_VARIANT:
            e->variant = (int)value;
            return NBX_OK;
        case NBX_OPT_DRAW_DEVICE:
            e->draw_device = value ? 1 : 0;
            return NBX_OK;
        case NBX_OPT_BH_WAVE:
            e->bh_wave = value ? 1 : 0;
            return NBX_OK;
        case NBX_OPT_BH_TREE:
            if (value != 0 && value != 1) return fail(NBX_ERR_INVALID, "bh tree must be 0 (host) or 1 (device)");
            e->bh_tree_device = (int)value;
            return NBX_OK;
        case NBX_OPT_SOURCE_PRECISION:
            if (value != 16 && value != 32) return fail(NBX_ERR_INVALID, "source precision must be 16 or 32");
            e->source_half = value == 16;
            if (e->source_half && e->dev_valid) {   // device state is live: build the fp16 copy from it now
                int rc = ensure_device(e);
                if (rc != NBX_OK) return rc;
                return refresh_half_sources(e, 0, e->n_pad);
            }
            return NBX_OK;
        default:
            return fail(NBX_ERR_INVALID, "unknown option %d", option);
    }
}

int64_t nbx_get_option(const nbx_engine* e, int32_t option)
{
    if (!e) return NBX_ERR_INVALID;
    switch (option) {
        case NBX_OPT_FORCE_MODE: return e->force_mode;
        case NBX_OPT_JSPLIT: return e->jsplit;
        case NBX_OPT_BODIES_PER_THREAD: return e->bpt;
        case NBX_OPT_DIM: return e->dim_opt;
        case NBX_OPT_PROFILE: return e->profile;
        case NBX_OPT_KERNEL_VARIANT: return e->variant;
        case NBX_OPT_SOURCE_PRECISION: return e->source_half ? 16 : 32;
        case NBX_OPT_DRAW_DEVICE: return e->draw_device;
        case NBX_OPT_BH_TREE: return e->bh_tree_device;
        case NBX_OPT_BH_WAVE: return e->bh_wave;
        default: return NBX_ERR_INVALID;
    }
}

int32_t nbx_seed(nbx_engine* e, uint64_t seed)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    e->rng.s = seed;
    e->seeded = true;
    return NBX_OK;
}

static void ensure_seed(nbx_engine* e)
{
    if (e->seeded) return;
    const char* env = std::getenv("NB_SEED");
    e->rng.s = env ? std::strtoull(env, nullptr, 0) : entropy_seed();
    e->seeded = true;
}

int32_t nbx_random_disk(nbx_engine* e, int32_t n)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    ensure_seed(e);
    nbx::preset_random_disk(e->host, n, e->rng);
    after_host_state_change(e);
    return NBX_OK;
}

int32_t nbx_stable_orbits(nbx_engine* e, int32_t n, float rmin, float rmax)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    ensure_seed(e);
    nbx::preset_stable_orbits(e->host, n, rmin, rmax, e->rng);
    after_host_state_change(e);
    return NBX_OK;
}

int32_t nbx_num_particles(const nbx_engine* e) { return e ? e->n : NBX_ERR_INVALID; }

int32_t nbx_set_particles3(nbx_engine* e, int32_t n, const float* px, const float* py, const float* pz, const float* vx,
                           const float* vy, const float* vz, const float* m)
{
    if (!e || n < 0) return fail(NBX_ERR_INVALID, "bad engine or n");
    if (n > 0 && (!px || !py || !vx || !vy || !m)) return fail(NBX_ERR_INVALID, "null input array");
    e->host.resize(n);
    for (int i = 0; i < n; i++) {
        e->host.px[i] = px[i]; e->host.py[i] = py[i]; e->host.pz[i] = pz ? pz[i] : 0.0f;
        e->host.vx[i] = vx[i]; e->host.vy[i] = vy[i]; e->host.vz[i] = vz ? vz[i] : 0.0f;
        e->host.m[i] = m[i];
    }
    after_host_state_change(e);
    return NBX_OK;
}

int32_t nbx_set_particles(nbx_engine* e, int32_t n, const float* px, const float* py, const float* vx, const float* vy,
                          const float* m)
{
    return nbx_set_particles3(e, n, px, py, nullptr, vx, vy, nullptr, m);
}

int32_t nbx_get_particles3(nbx_engine* e, int32_t cap, float* px, float* py, float* pz, float* vx, float* vy, float* vz,
                           float* m)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    if (cap < e->n) return fail(NBX_ERR_INVALID, "capacity %d < particle count %d", cap, e->n);
    int rc = download_positions(e);
    if (rc != NBX_OK) return rc;
    rc = download_velocities(e);
    if (rc != NBX_OK) return rc;
    const size_t bytes = sizeof(float) * (size_t)e->n;
    if (px) std::memcpy(px, e->host.px.data(), bytes);
    if (py) std::memcpy(py, e->host.py.data(), bytes);
    if (pz) std::memcpy(pz, e->host.pz.data(), bytes);
    if (vx) std::memcpy(vx, e->host.vx.data(), bytes);
    if (vy) std::memcpy(vy, e->host.vy.data(), bytes);
    if (vz) std::memcpy(vz, e->host.vz.data(), bytes);
    if (m) std::memcpy(m, e->host.m.data(), bytes);
    return e->n;
}

int32_t nbx_get_particles(nbx_engine* e, int32_t cap, float* px, float* py, float* vx, float* vy, float* m)
{
    return nbx_get_particles3(e, cap, px, py, nullptr, vx, vy, nullptr, m);
}

int32_t nbx_step_brute_force(nbx_engine* e, float dt)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    return step_brute(e, dt);
}

int32_t nbx_step_barnes_hut(nbx_engine* e, float theta, float dt, int32_t nthreads)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    if (theta == 0.0f) return step_brute(e, dt);  // nbody.rs:197-200 (exact compare, before anything else)
    if (nthreads <= 0) return fail(NBX_ERR_INVALID, "nthreads must be >= 1 (the reference divides by it, nbody.rs:426)");
    return step_bh(e, theta, dt);
}

int32_t nbx_step_local(nbx_engine* e, float dt)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    return step_brute(e, dt);
}

int32_t nbx_synchronize(nbx_engine* e)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    if (!e->dev_ready) return NBX_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return NBX_OK;
}

int32_t nbx_forces(nbx_engine* e, float theta, int32_t cap, float* fx, float* fy, float* fz)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    int rc = upload(e);
    if (rc != NBX_OK) return rc;
    const int slab = e->slab();
    if (cap < slab) return fail(NBX_ERR_INVALID, "capacity %d < slab %d", cap, slab);
    if (slab == 0) return 0;
    if (theta == 0.0f && e->force_mode == 0) {
        rc = launch_forces_fast(e);
        if (rc != NBX_OK) return rc;
        rc = grow(&e->d_out4, &e->out4_cap, (size_t)slab);
        if (rc != NBX_OK) return rc;
        const int stride = ((slab + kTile - 1) / kTile) * kTile;
        HIP_TRY(nbx::launch_reduce_forces(e->d_posm, e->lo, slab, e->d_acc, e->last.jsplit, stride, e->d_out4, e->stream));
        std::vector<float4> tmp((size_t)slab);
        HIP_TRY(hipMemcpyAsync(tmp.data(), e->d_out4, sizeof(float4) * (size_t)slab, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        for (int i = 0; i < slab; i++) {
            if (fx) fx[i] = tmp[i].x;
            if (fy) fy[i] = tmp[i].y;
            if (fz) fz[i] = tmp[i].z;
        }
        return slab;
    }
    rc = grow(&e->d_f2, &e->f2_cap, (size_t)slab);
    if (rc != NBX_OK) return rc;
    bool is_accel = false;
    if (theta == 0.0f) {
        ProfScope ps(e, NBX_K_FORCE);
        HIP_TRY(nbx::launch_force_strict(e->d_posm, e->n, e->lo, slab, e->d_f2, e->stream));
    } else {
        bool on_device = false;
        if (e->bh_tree_device && e->force_mode == 0) {
            rc = build_tree_on_device(e, &on_device);
            if (rc != NBX_OK) return rc;
        }
        if (!on_device) {
            rc = build_and_upload_tree(e);
            if (rc != NBX_OK) return rc;
        }
        ProfScope ps(e, NBX_K_BH_EVAL);
        const unsigned* perm = (on_device && e->world == 1) ? e->d_perm : nullptr;
        HIP_TRY(nbx::launch_bh_eval(e->d_posm, e->lo, slab, e->d_nodes, (int)e->n_flat, theta,
                                    (e->force_mode == 0 && perm && e->bh_wave) ? 2 : e->force_mode, e->d_f2, e->stream, perm));
        is_accel = e->force_mode == 0;
    }
    std::vector<float2> tmp((size_t)slab);
    HIP_TRY(hipMemcpyAsync(tmp.data(), e->d_f2, sizeof(float2) * (size_t)slab, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int i = 0; i < slab; i++) {
        const float mi = is_accel ? e->host.m[e->lo + i] : 1.0f;
        if (fx) fx[i] = is_accel ? mi * tmp[i].x : tmp[i].x;
        if (fy) fy[i] = is_accel ? mi * tmp[i].y : tmp[i].y;
        if (fz) fz[i] = 0.0f;
    }
    return slab;
}

// device splat (draw.hip): needs the whole state on this GPU (unsharded) and a live device
static int draw_on_device(nbx_engine* e, int32_t w, int32_t h, uint32_t* fb)
{
    int rc = upload(e);
    if (rc != NBX_OK) return rc;
    const size_t px = (size_t)w * (size_t)h;
    if (px > e->counts_cap) {
        if (e->d_counts) HIP_TRY(hipFree(e->d_counts));
        if (e->d_fb) HIP_TRY(hipFree(e->d_fb));
        e->d_counts = nullptr; e->d_fb = nullptr; e->counts_cap = e->fb_cap = 0;
        HIP_TRY(hipMalloc(&e->d_counts, px * 8));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_fb), px * 4));
        e->counts_cap = e->fb_cap = px;
    }
    // viewport transform evaluated on the host exactly as nbody.rs:494-506 does (f32, same order)
    const float aspect = (float)h / (float)w;
    const float x1 = 0.0f - 100.0f / 2.0f, y1 = (0.0f - 100.0f / 2.0f) * aspect;
    const float x2 = 0.0f + 100.0f / 2.0f, y2 = (0.0f + 100.0f / 2.0f) * aspect;
    const float scalex = (1.0f / (x2 - x1)) * (float)w, scaley = (1.0f / (y2 - y1)) * (float)h;
    HIP_TRY(nbx::launch_draw(e->d_posm, e->d_vel, e->n, w, h, x1, y1, scalex, scaley, e->d_counts, e->d_fb, e->stream));
    HIP_TRY(hipMemcpyAsync(fb, e->d_fb, px * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return NBX_OK;
}

int32_t nbx_draw(nbx_engine* e, int32_t w, int32_t h, uint32_t* fb)
{
    if (!e || !fb || w <= 0 || h <= 0) return fail(NBX_ERR_INVALID, "bad draw arguments");
    if (e->draw_device) {
        if (e->world != 1) return fail(NBX_ERR_STATE, "device draw needs the whole state on one GPU");
        return draw_on_device(e, w, h, fb);
    }
    int rc = download_positions(e);
    if (rc != NBX_OK) return rc;
    rc = download_velocities(e);
    if (rc != NBX_OK) return rc;
    nbx::draw_particles(e->host.px.data(), e->host.py.data(), e->host.vx.data(), e->host.vy.data(), e->n, w, h, fb);
    return NBX_OK;
}

int32_t nbx_bh_tree_dump(nbx_engine* e, float* rows, int32_t cap)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    int rc = download_positions(e);
    if (rc != NBX_OK) return rc;
    rc = e->tree.build(e->host.px.data(), e->host.py.data(), e->host.m.data(), e->n);
    if (rc != NBX_OK) return fail(rc, "quadtree build failed (%d)", rc);
    return e->tree.dump_preorder(rows, cap);
}

int32_t nbx_bh_flat_dump(nbx_engine* e, void* rows, int32_t cap, int32_t threaded)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    if (threaded == 2) {   // the DEVICE-built tree (needs a GPU)
        int rc0 = upload(e);
        if (rc0 != NBX_OK) return rc0;
        bool done = false;
        rc0 = build_tree_on_device(e, &done);
        if (rc0 != NBX_OK) return rc0;
        if (!done) return fail(NBX_ERR_STATE, "device tree build fell back (node pool exhausted)");
        if ((size_t)cap >= e->n_flat && rows && e->n_flat) {
            HIP_TRY(hipMemcpyAsync(rows, e->d_nodes, sizeof(nbx::BhNode) * e->n_flat, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
        }
        return (int32_t)e->n_flat;
    }
    int rc = download_positions(e);
    if (rc != NBX_OK) return rc;
    rc = e->tree.build(e->host.px.data(), e->host.py.data(), e->host.m.data(), e->n);
    if (rc != NBX_OK) return fail(rc, "quadtree build failed (%d)", rc);
    if (threaded && e->tree.forest) {
        const size_t count = e->tree.flatten_prepare(e->plan);
        if ((size_t)cap >= count && rows) e->tree.flatten_write(e->plan, static_cast<nbx::BhNode*>(rows));
        return (int32_t)count;
    }
    e->tree.flatten(e->flat_small);
    if ((size_t)cap >= e->flat_small.size() && rows && !e->flat_small.empty())
        std::memcpy(rows, e->flat_small.data(), sizeof(nbx::BhNode) * e->flat_small.size());
    return (int32_t)e->flat_small.size();
}

// ---- checkpoint: the reference has none (state is lost on every experiment switch) ----
// File = "NBXCKPT1" | int32 n | int32 reserved | 7 arrays of n little-endian f32: px py pz vx vy vz m
int32_t nbx_save(nbx_engine* e, const char* path)
{
    if (!e || !path) return fail(NBX_ERR_INVALID, "null argument");
    if (e->world != 1) return fail(NBX_ERR_STATE, "save the gathered state from rank 0 of a sharded run via get/set");
    int rc = download_positions(e);
    if (rc != NBX_OK) return rc;
    rc = download_velocities(e);
    if (rc != NBX_OK) return rc;
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(NBX_ERR_INVALID, "cannot open %s for writing", path);
    const int32_t hdr[2] = {e->n, 0};
    bool ok = std::fwrite("NBXCKPT1", 1, 8, f) == 8 && std::fwrite(hdr, sizeof hdr, 1, f) == 1;
    const std::vector<float>* arrs[7] = {&e->host.px, &e->host.py, &e->host.pz, &e->host.vx, &e->host.vy, &e->host.vz, &e->host.m};
    for (auto* a : arrs) ok = ok && (e->n == 0 || std::fwrite(a->data(), sizeof(float), (size_t)e->n, f) == (size_t)e->n);
    ok = (std::fclose(f) == 0) && ok;
    return ok ? NBX_OK : fail(NBX_ERR_INVALID, "short write to %s", path);
}

int32_t nbx_load(nbx_engine* e, const char* path)
{
    if (!e || !path) return fail(NBX_ERR_INVALID, "null argument");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(NBX_ERR_INVALID, "cannot open %s", path);
    char magic[8];
    int32_t hdr[2] = {0, 0};
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "NBXCKPT1", 8) == 0 && std::fread(hdr, sizeof hdr, 1, f) == 1 &&
              hdr[0] >= 0;
    nbx::HostState st;
    if (ok) {
        st.resize(hdr[0]);
        std::vector<float>* arrs[7] = {&st.px, &st.py, &st.pz, &st.vx, &st.vy, &st.vz, &st.m};
        for (auto* a : arrs) ok = ok && (hdr[0] == 0 || std::fread(a->data(), sizeof(float), (size_t)hdr[0], f) == (size_t)hdr[0]);
    }
    std::fclose(f);
    if (!ok) return fail(NBX_ERR_INVALID, "%s is not a valid NBXCKPT1 checkpoint", path);
    e->host = std::move(st);
    after_host_state_change(e);
    return e->n;
}

int32_t nbx_set_shard(nbx_engine* e, int32_t rank, int32_t world)
{
    if (!e || world < 1 || rank < 0 || rank >= world) return fail(NBX_ERR_INVALID, "bad shard %d/%d", rank, world);
    if (e->dev_valid) return fail(NBX_ERR_STATE, "set the shard before the state is uploaded");
    e->rank = rank;
    e->world = world;
    compute_slab(e);
    return NBX_OK;
}

int32_t nbx_get_slab(const nbx_engine* e, int32_t* lo, int32_t* hi)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    if (lo) *lo = e->lo;
    if (hi) *hi = e->hi;
    return NBX_OK;
}

size_t nbx_positions_bytes(const nbx_engine* e)
{
    if (!e) return 0;
    int n_pad = ((e->n + kTile - 1) / kTile) * kTile;
    if (n_pad == 0) n_pad = kTile;
    return sizeof(float4) * (size_t)n_pad;
}

int32_t nbx_bind_positions(nbx_engine* e, void* device_ptr, size_t bytes)
{
    if (!e || !device_ptr) return fail(NBX_ERR_INVALID, "null argument");
    if (bytes < nbx_positions_bytes(e)) return fail(NBX_ERR_INVALID, "buffer too small: %zu < %zu", bytes, nbx_positions_bytes(e));
    int rc = ensure_device(e);
    if (rc != NBX_OK) return rc;
    rc = download_positions(e);   // keep whatever the device currently holds
    if (rc != NBX_OK) return rc;
    rc = download_velocities(e);
    if (rc != NBX_OK) return rc;
    if (e->d_posm && !e->posm_external) HIP_TRY(hipFree(e->d_posm));
    e->d_posm = static_cast<float4*>(device_ptr);
    e->posm_external = true;
    e->posm_cap = bytes / sizeof(float4);
    e->dev_valid = false;
    return upload(e);
}

size_t nbx_half_sources_bytes(const nbx_engine* e) { return nbx_positions_bytes(e) / 2; }

int32_t nbx_bind_half_sources(nbx_engine* e, void* device_ptr, size_t bytes)
{
    if (!e || !device_ptr) return fail(NBX_ERR_INVALID, "null argument");
    if (!e->source_half) return fail(NBX_ERR_STATE, "set NBX_OPT_SOURCE_PRECISION to 16 first");
    if (bytes < nbx_half_sources_bytes(e)) return fail(NBX_ERR_INVALID, "buffer too small: %zu < %zu", bytes, nbx_half_sources_bytes(e));
    int rc = upload(e);
    if (rc != NBX_OK) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->d_posh && !e->posh_external) HIP_TRY(hipFree(e->d_posh));
    e->d_posh = device_ptr;
    e->posh_external = true;
    e->posh_cap = bytes / 8;
    return refresh_half_sources(e, 0, e->n_pad);
}

void* nbx_positions_device(nbx_engine* e)
{
    if (!e) return nullptr;
    if (upload(e) != NBX_OK) return nullptr;
    return e->d_posm;
}

int32_t nbx_set_stream(nbx_engine* e, void* hip_stream)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    int rc = ensure_device(e);
    if (rc != NBX_OK) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->own_stream && e->stream) HIP_TRY(hipStreamDestroy(e->stream));
    e->stream = static_cast<hipStream_t>(hip_stream);
    e->own_stream = false;
    return NBX_OK;
}

int32_t nbx_profile_reset(nbx_engine* e)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    if (e->dev_ready) {
        HIP_TRY(hipSetDevice(e->device));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    for (auto& r : e->prof) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    e->prof.clear();
    return NBX_OK;
}

int32_t nbx_profile_read(nbx_engine* e, int32_t kernel_id, double* total_ms, int32_t* launches)
{
    if (!e || kernel_id < 0 || kernel_id >= NBX_K_COUNT) return fail(NBX_ERR_INVALID, "bad kernel id");
    double total = 0.0;
    int count = 0;
    if (e->dev_ready) {
        HIP_TRY(hipSetDevice(e->device));
        HIP_TRY(hipStreamSynchronize(e->stream));
        for (auto& r : e->prof) {
            if (r.kernel != kernel_id) continue;
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, r.start, r.stop));
            total += ms;
            count++;
        }
    }
    if (total_ms) *total_ms = total;
    if (launches) *launches = count;
    return NBX_OK;
}

int32_t nbx_bh_work(nbx_engine* e, float theta, uint64_t* node_visits, uint64_t* pair_evals)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    int rc = upload(e);
    if (rc != NBX_OK) return rc;
    bool on_device = false;
    if (e->bh_tree_device && e->force_mode == 0) {
        rc = build_tree_on_device(e, &on_device);
        if (rc != NBX_OK) return rc;
    }
    if (!on_device) {
        rc = build_and_upload_tree(e);
        if (rc != NBX_OK) return rc;
    }
    unsigned long long* d_tot = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_tot), 16));
    HIP_TRY(hipMemsetAsync(d_tot, 0, 16, e->stream));
    HIP_TRY(nbx::launch_bh_count(e->d_posm, e->lo, e->slab(), e->d_nodes, (int)e->n_flat, theta, d_tot, e->stream));
    unsigned long long h[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(h, d_tot, 16, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipFree(d_tot));
    if (node_visits) *node_visits = h[0];
    if (pair_evals) *pair_evals = h[1];
    return NBX_OK;
}

int32_t nbx_bh_host_timing(nbx_engine* e, double* ms4, int32_t* steps, int32_t* nodes)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    if (ms4) for (int i = 0; i < 4; i++) ms4[i] = e->host_ms[i];
    if (steps) *steps = e->host_steps;
    if (nodes) *nodes = (int32_t)e->n_flat;
    for (int i = 0; i < 4; i++) e->host_ms[i] = 0;
    e->host_steps = 0;
    return NBX_OK;
}

int32_t nbx_last_launch(const nbx_engine* e, int32_t* grid, int32_t* block, int32_t* jsplit, int32_t* bodies_per_thread,
                        int32_t* dim, int32_t* variant)
{
    if (e && variant) *variant = e->last.variant;
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    if (grid) *grid = e->last.grid;
    if (block) *block = e->last.block;
    if (jsplit) *jsplit = e->last.jsplit;
    if (bodies_per_thread) *bodies_per_thread = e->last.bpt;
    if (dim) *dim = e->last.dim;
    return NBX_OK;
}


// =============================================================================================
// Single-process multi-GPU group: what the unmodified Haskell caller needs to use every GPU of a node.
// G engines, one per device, slab-sharded exactly like the multi-process path (nbody.rs:426-428 split);
// per step every device runs K1+K2 on its slab on its own stream, then ONE RCCL all-gather of the
// (x,y,z,m) array (ncclCommInitAll communicators, one group call).  RCCL is dlopen'ed on first use so that
// single-GPU users carry no dependency on it.
// =============================================================================================
}  // extern "C"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only; the functions are resolved at run time

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

struct nbx_group {
    std::vector<nbx_engine*> eng;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    int exchanges = 0;
};

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

// one all-gather of the (x,y,z,m) slabs: in place, sendbuff = recvbuff + lo (per device), same stream as the kernels
int group_exchange(nbx_group* g)
{
    const int G = (int)g->eng.size();
    int rc = group_comms(g);
    if (rc != NBX_OK) return rc;
    RcclApi* api = rccl_api();
    const int n = g->eng[0]->n;
    if (n == 0) return NBX_OK;
    RCCL_TRY(api, api->GroupStart());
    if (n % G == 0) {
        for (int d = 0; d < G; d++) {
            nbx_engine* e = g->eng[d];
            RCCL_TRY(api, api->AllGather(e->d_posm + e->lo, e->d_posm, (size_t)e->slab() * 4, ncclFloat32, g->comms[d], e->stream));
        }
    } else {   // ragged last slab (reference split): one broadcast per owner
        for (int r = 0; r < G; r++) {
            const int lo = g->eng[r]->lo, cnt = g->eng[r]->slab();
            if (cnt == 0) continue;
            for (int d = 0; d < G; d++) {
                nbx_engine* e = g->eng[d];
                RCCL_TRY(api, api->Broadcast(e->d_posm + lo, e->d_posm + lo, (size_t)cnt * 4, ncclFloat32, r, g->comms[d], e->stream));
            }
        }
    }
    RCCL_TRY(api, api->GroupEnd());
    for (nbx_engine* e : g->eng) e->host_pos_valid = false;
    g->exchanges++;
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
    for (int i = 0; i < count; i++) {
        const int dev = devices ? devices[i] : i;
        for (int j = 0; j < i; j++)
            if (g->devices[j] == dev) { nbx_group_destroy(g); return fail(NBX_ERR_INVALID, "device %d listed twice", dev); }
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
    return NBX_OK;
}

int32_t nbx_group_get_particles3(nbx_group* g, int32_t cap, float* px, float* py, float* pz, float* vx, float* vy, float* vz,
                                 float* m)
{
    if (!g) return fail(NBX_ERR_INVALID, "null group");
    nbx_engine* e0 = g->eng[0];
    if (cap < e0->n) return fail(NBX_ERR_INVALID, "capacity %d < particle count %d", cap, e0->n);
    int rc = nbx_get_particles3(e0, cap, px, py, pz, vx, vy, vz, m);   // positions are replicated after the all-gather
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
    for (nbx_engine* e : g->eng) {   // asynchronous: every device works on its slab concurrently
        const int rc = step_brute(e, dt);
        if (rc != NBX_OK) return rc;
    }
    return group_exchange(g);
}

int32_t nbx_group_step_barnes_hut(nbx_group* g, float theta, float dt, int32_t nthreads)
{
    if (!g) return fail(NBX_ERR_INVALID, "null group");
    if (theta == 0.0f) return nbx_group_step_brute_force(g, dt);   // nbody.rs:197-200
    if (nthreads <= 0) return fail(NBX_ERR_INVALID, "nthreads must be >= 1");
    for (nbx_engine* e : g->eng) {   // tree replica per device; each evaluates its slab
        const int rc = step_bh(e, theta, dt);
        if (rc != NBX_OK) return rc;
    }
    return group_exchange(g);
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

// =============================================================================================
// Level 1: the reference's six symbols on a process-global engine
// =============================================================================================

static std::mutex g_mutex;           // PARTICLES: Mutex<..> (nbody.rs:28-32)
static nbx_engine* g_engine = nullptr;
static nbx_group* g_group = nullptr;   // NB_GPUS > 1: every call below is served by the multi-GPU group

[[noreturn]] static void die(const char* where)
{
    // the reference panics (and poisons its mutex) on failure; across the C ABI that is an abort
    std::fprintf(stderr, "nbody_mi355x: fatal in %s: %s\n", where, nbx_last_error());
    std::abort();
}

static void apply_env(nbx_engine* e)
{
    const char* mode = std::getenv("NB_FORCE_MODE");
    if (mode && std::strcmp(mode, "strict") == 0) e->force_mode = 1;
    const char* tree = std::getenv("NB_BH_TREE");
    if (tree && std::strcmp(tree, "device") == 0) e->bh_tree_device = 1;
    const char* draw = std::getenv("NB_DRAW");
    if (draw && std::strcmp(draw, "device") == 0) e->draw_device = 1;
}

static nbx_engine* global_engine()   // engine 0 of the group when NB_GPUS > 1
{
    if (!g_engine) {
        const char* gpus = std::getenv("NB_GPUS");
        int want = gpus ? (std::strcmp(gpus, "all") == 0 ? nbx_device_count() : std::atoi(gpus)) : 1;
        if (want > 1) {
            if (nbx_group_create(&g_group, nullptr, want) != NBX_OK) die("NB_GPUS group creation");
            for (nbx_engine* e : g_group->eng) apply_env(e);
            g_engine = g_group->eng[0];
            return g_engine;
        }
        const char* dev = std::getenv("NB_DEVICE");
        if (nbx_create(&g_engine, dev ? std::atoi(dev) : 0) != NBX_OK) die("engine creation");
        apply_env(g_engine);
    }
    return g_engine;
}

// after a preset ran on engine 0 (host side), replicate its state to the other engines of the group
static int replicate_preset()
{
    if (!g_group) return NBX_OK;
    nbx_engine* e0 = g_group->eng[0];
    for (size_t d = 1; d < g_group->eng.size(); d++) {
        const int rc = nbx_set_particles3(g_group->eng[d], e0->n, e0->host.px.data(), e0->host.py.data(), e0->host.pz.data(),
                                          e0->host.vx.data(), e0->host.vy.data(), e0->host.vz.data(), e0->host.m.data());
        if (rc != NBX_OK) return rc;
    }
    return NBX_OK;
}

int32_t nb_num_particles(void)
{
    std::lock_guard<std::mutex> lk(g_mutex);
    return nbx_num_particles(global_engine());
}

void nb_random_disk(int32_t num_particles)
{
    std::lock_guard<std::mutex> lk(g_mutex);
    if (nbx_random_disk(global_engine(), num_particles) != NBX_OK || replicate_preset() != NBX_OK) die("nb_random_disk");
}

void nb_stable_orbits(int32_t num_particles, float rmin, float rmax)
{
    std::lock_guard<std::mutex> lk(g_mutex);
    if (nbx_stable_orbits(global_engine(), num_particles, rmin, rmax) != NBX_OK || replicate_preset() != NBX_OK)
        die("nb_stable_orbits");
}

void nb_step_brute_force(float dt)
{
    std::lock_guard<std::mutex> lk(g_mutex);
    nbx_engine* e = global_engine();
    if (g_group) {
        if (nbx_group_step_brute_force(g_group, dt) != NBX_OK || nbx_group_synchronize(g_group) != NBX_OK) die("nb_step_brute_force");
        return;
    }
    if (nbx_step_brute_force(e, dt) != NBX_OK || nbx_synchronize(e) != NBX_OK) die("nb_step_brute_force");
}

void nb_step_barnes_hut(float theta, float dt, int32_t nthreads)
{
    std::lock_guard<std::mutex> lk(g_mutex);
    nbx_engine* e = global_engine();
    if (theta != 0.0f && nthreads <= 0) return;  // reference: integer division by zero panic; here a no-op
    if (g_group) {
        if (nbx_group_step_barnes_hut(g_group, theta, dt, nthreads) != NBX_OK || nbx_group_synchronize(g_group) != NBX_OK)
            die("nb_step_barnes_hut");
        return;
    }
    if (nbx_step_barnes_hut(e, theta, dt, nthreads) != NBX_OK || nbx_synchronize(e) != NBX_OK) die("nb_step_barnes_hut");
}

void nb_draw(int32_t w, int32_t h, uint32_t* fb)
{
    std::lock_guard<std::mutex> lk(g_mutex);
    if (w <= 0 || h <= 0 || !fb) return;
    nbx_engine* e = global_engine();
    if (g_group) {
        if (nbx_group_draw(g_group, w, h, fb) != NBX_OK) die("nb_draw");
        return;
    }
    if (nbx_draw(e, w, h, fb) != NBX_OK) die("nb_draw");
}

}  // extern "C"
