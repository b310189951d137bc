// engine.cpp -- state owner and step drivers of libnbody_mi355x.so (ABI: include/nbody_mi355x.h; the entry
// points live in c_api.cpp / group.cpp / level1.cpp).
//
// Plays the role the north_star gives to "Rust host code": it owns the particle arrays (a host
// SoA mirror + the device-resident float4 arrays) and drives the gfx950 kernels.  No CPU fallback:
// steps need a device.
//
// HBM layout (all 16-B records, coalesced 16 B/lane):
//   posm  float4[n_pad]        (x, y, z, m) for ALL bodies; n_pad = n rounded up to 256, padding
//                              entries are (0,0,0,0): zero mass => exact zero contribution.
//   vel   float4[slab]         (vx, vy, vz, 0) for this engine's slab of targets only.
//   acc   float4[S][stride]    per-source-split partial accelerations of the fast kernel.
//   f2    float2[slab]         forces (strict / Barnes-Hut paths).
//   nodes BhNode[n_nodes]      flattened quadtree, rebuilt on the host every Barnes-Hut step.
#include <cmath>
#include <limits>
#include <random>

#include "engine_internal.h"

namespace nbxi {

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
        return e->any_pending() ? resolve_pending(e) : NBX_OK;
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

// pinned staging buffer of at least `records` float4 (shared by the position and velocity downloads)
static int ensure_stage(nbx_engine* e, size_t records)
{
    if (records <= e->h_stage_cap) return NBX_OK;
    if (e->h_stage) HIP_TRY(hipHostFree(e->h_stage));
    e->h_stage = nullptr;
    e->h_stage_cap = 0;
    const size_t want = std::max<size_t>(records, 256);
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_stage), sizeof(float4) * want, hipHostMallocDefault));
    e->h_stage_cap = want;
    return NBX_OK;
}

// per-record conversion between the AoS staging buffer and the SoA host mirror, on a few pool threads for big systems
template <typename F>
static void unpack_records(int count, F&& one)
{
    if (count >= 262144) {
        const int parts = 8;
        nbx::parallel_for(parts, [&](int p) {
            const int a = (int)((long long)count * p / parts), b = (int)((long long)count * (p + 1) / parts);
            for (int i = a; i < b; i++) one(i);
        });
    } else {
        for (int i = 0; i < count; i++) one(i);
    }
}

int upload(nbx_engine* e)
{
    int rc = ensure_device(e);
    if (rc != NBX_OK) return rc;
    if (e->dev_valid) return NBX_OK;
    if (!(e->host_pos_valid && e->host_vel_valid)) return fail(NBX_ERR_STATE, "no valid state to upload");
    const int n = e->n;
    e->sort_warm_n = 0;   // new positions from the host: last step's order says nothing about them
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
    // pack SoA -> float4 records in the pinned staging buffer (threads for big systems), one copy per array
    rc = ensure_stage(e, (size_t)e->n_pad);
    if (rc != NBX_OK) return rc;
    float4* tmp = e->h_stage;
    unpack_records(e->n_pad, [&](int i) {
        tmp[i] = i < n ? make_float4(e->host.px[i], e->host.py[i], e->host.pz[i], e->host.m[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    });
    HIP_TRY(hipMemcpyAsync(e->d_posm, tmp, sizeof(float4) * (size_t)e->n_pad, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (slab > 0) {
        const int lo = e->lo;
        unpack_records(slab, [&](int i) { tmp[i] = make_float4(e->host.vx[lo + i], e->host.vy[lo + i], e->host.vz[lo + i], 0.f); });
        HIP_TRY(hipMemcpyAsync(e->d_vel, tmp, sizeof(float4) * (size_t)slab, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    if (!e->exc_idx.empty()) {   // the exceptional sources of the unit-mass sweep (their weights m_j - mass_common follow on the device)
        const size_t k = e->exc_idx.size();
        if (k > e->exc_cap_dev) {
            if (e->d_exc_idx) HIP_TRY(hipFree(e->d_exc_idx));
            if (e->d_exc_rec) HIP_TRY(hipFree(e->d_exc_rec));
            e->d_exc_idx = nullptr; e->d_exc_rec = nullptr; e->exc_cap_dev = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_exc_idx), sizeof(int) * k));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_exc_rec), sizeof(float4) * k));
            e->exc_cap_dev = k;
        }
        HIP_TRY(hipMemcpyAsync(e->d_exc_idx, e->exc_idx.data(), sizeof(int) * k, hipMemcpyHostToDevice, e->stream));
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
    rc = ensure_stage(e, (size_t)e->n);
    if (rc != NBX_OK) return rc;
    float4* tmp = e->h_stage;
    HIP_TRY(hipMemcpyAsync(tmp, e->d_posm, sizeof(float4) * (size_t)e->n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    unpack_records(e->n, [&](int i) { e->host.px[i] = tmp[i].x; e->host.py[i] = tmp[i].y; e->host.pz[i] = tmp[i].z; });
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
        rc = ensure_stage(e, (size_t)slab);
        if (rc != NBX_OK) return rc;
        float4* tmp = e->h_stage;
        HIP_TRY(hipMemcpyAsync(tmp, e->d_vel, sizeof(float4) * (size_t)slab, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        const int lo = e->lo;
        unpack_records(slab, [&](int i) { e->host.vx[lo + i] = tmp[i].x; e->host.vy[lo + i] = tmp[i].y; e->host.vz[lo + i] = tmp[i].z; });
    }
    e->host_vel_valid = true;
    return NBX_OK;
}


void choose_launch(const nbx_engine* e, int n_targets, int tiles_total, int* variant, int* bpt, int* jsplit, int* dim)
{
    // Defaults from the measured launch-shape sweeps (profiles/r01_shapes_sweep*.txt, profiles/r02_k1_wave_split_sweep.txt):
    //  * kernel, >= 16384 sources: the wave-split scalar-cache sweep -- variant 7 (unit-mass: 9 packed ops + 2 rcp per two
    //    interactions) when every body has the same mass, else variant 6 (10 + 2); LDS tiles (variant 1) below that size.
    //  * variants 6 / 7: 256 targets per workgroup, S = smallest power of two giving >= 32 workgroups per CU (64 when a GPU
    //    owns < 131072 targets: tail effect), at most 64 and at least 4 source tiles per workgroup (one per wave).
    //    N = 262144: S = 8 (34 MB of partial slabs per launch; S = 8..32 are within 1 % of each
    //    other), N = 65536: S = 64, 32768 targets x 262144 sources (8-way shard): S = 64.
    //  * variant 1: register blocking 4 (two packed pairs) when a GPU owns >= 32768 targets, else 2; S = smallest power
    //    of two giving >= 32 workgroups per CU, capped at 64 and half the tiles.
    *dim = e->dim_opt ? e->dim_opt : (e->any_z ? 3 : 2);
    int v = e->variant;
    if (v < 0) v = (tiles_total * kTile >= 16384) ? 7 : 1;   // crossover measured in profiles/r02_small_n_variants.txt
    if (v != 1 && v != 6 && v != 7) v = 1;      // (nbx_set_option admits no other)
    if (v == 7 && !e->unit_sweep_ok()) v = 6;   // unit-mass sweep needs one common mass (+ at most a handful of exceptions)
    *variant = v;
    // 256 targets per workgroup, 4 source quarters per workgroup; the fp16-source kernel (K4) keeps the 1024-target workgroups
    const bool wave_split = (v == 6 || v == 7);
    int b = wave_split ? 4 : (e->bpt ? e->bpt : (n_targets >= 32768 ? 4 : 2));
    if (b != 2 && b != 4) b = 2;                // packed pairs: two or four targets per thread
    *bpt = b;
    int s = e->jsplit;
    if (s <= 0) {
        const int per_wg = wave_split ? kTile : kTile * b;
        const int iblocks = (n_targets + per_wg - 1) / per_wg;
        // 64 workgroups per CU only where targets are scarce (sharded shapes: tail effect); 32 otherwise --
        // same speed at N = 262144 on one GPU and half the partial-slab traffic
        const int want = e->cu_count * ((wave_split && n_targets < 131072) ? 64 : 32);
        s = 1;
        while (iblocks * s < want && s < 64) s *= 2;
        // every workgroup keeps >= 2 tiles of sources; a wave-split workgroup >= 4 (one per wave).  Tiny systems (at most one
        // workgroup per CU even at one tile each) are latency-bound -- a wave alone on its SIMD needs ~8 us per tile -- and
        // take one tile per workgroup: N <= 4096, 20.5 -> 12.5 us per K1 launch (profiles/r02_small_n_variants.txt)
        int cap = std::max(1, tiles_total / (wave_split ? 4 : 2));
        if (!wave_split && iblocks * tiles_total <= e->cu_count) cap = tiles_total;
        s = std::min(s, cap);
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
    if (e->source_half && !(variant == 6 || variant == 7)) {   // small systems: the LDS-tile sweep on the half4 copy
        ProfScope ps(e, NBX_K_FORCE);
        HIP_TRY(nbx::launch_force_tile_half(e->d_posm, e->d_posh, e->lo, slab, tiles_total, jsplit, bpt, dim, e->d_acc,
                                            stride, e->stream, &e->last));
        return NBX_OK;
    }
    if (variant == 6 || variant == 7) {
        const float4* widened = nullptr;
        float mass = e->mass_common;
        if (e->source_half) {
            // K4 on the wave-split kernels: the half4 copy widened to float4 once per step (exact), then the same sweep with those
            // records as sources; the common mass is the fp16 image of the bodies' mass, like every source's
            rc = grow(&e->d_src4, &e->src4_cap, (size_t)e->n_pad);
            if (rc != NBX_OK) return rc;
            widened = e->d_src4;
            mass = nbx::half_image(e->mass_common);
        }
        ProfScope ps(e, NBX_K_FORCE);
        if (widened) HIP_TRY(nbx::launch_widen_half(e->d_posh, e->d_src4, e->n_pad, e->stream));
        const int n_exc = variant == 7 ? (int)e->exc_idx.size() : 0;
        HIP_TRY(nbx::launch_force_wave_split(e->d_posm, e->lo, slab, tiles_total, e->n, jsplit, dim, variant == 7, mass,
                                             e->d_acc, stride, e->stream, &e->last, e->d_exc_idx, e->d_exc_rec, n_exc, widened));
        return NBX_OK;
    }
    {
        ProfScope ps(e, NBX_K_FORCE);
        HIP_TRY(nbx::launch_force_tile(e->d_posm, e->lo, slab, tiles_total, jsplit, bpt, dim, e->d_acc, stride, e->stream, &e->last));
    }
    return NBX_OK;
}

// the exceptional sources K2 / the force readout must add after a unit-mass sweep (variant 7) of a system with exceptions
nbx::MassExceptions exceptions_of(const nbx_engine* e)
{
    if ((e->last.variant != 7 && e->last.variant != 18) || e->exc_idx.empty()) return nbx::MassExceptions{nullptr, nullptr, 0, e->last.dim};
    return nbx::MassExceptions{e->d_exc_rec, e->d_exc_idx, (int)e->exc_idx.size(), e->last.dim};
}

// K4 on the wave-split kernels (variants 17 / 18): what K2 / the force readout must take out of every target's sum
nbx::SelfImage self_image_of(const nbx_engine* e)
{
    if (e->last.variant != 17 && e->last.variant != 18) return nbx::SelfImage{nullptr, 0.0f, e->last.dim};
    return nbx::SelfImage{e->d_src4, e->last.variant == 18 ? nbx::half_image(e->mass_common) : 0.0f, e->last.dim};
}

// NBX_LOG=1: one stderr line per step (the reference has no logging on this path; its Haskell shell has Trace.hs)
static bool log_enabled()
{
    static const bool on = std::getenv("NBX_LOG") != nullptr;
    return on;
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
            unsigned* guard = nullptr;   // the short exact division needs the mass range here and max|coordinate| on the device
            if (nbx::strict_fastdiv_ok(e->mass_min, e->mass_max)) {
                rc = grow(&e->d_guard, &e->guard_cap, 1);
                if (rc != NBX_OK) return rc;
                guard = e->d_guard;
            }
            HIP_TRY(nbx::launch_force_strict(e->d_posm, e->n, e->lo, slab, e->d_f2, e->stream, &e->last, guard, e->strict_kernel));
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
                                      e->stream, exceptions_of(e), self_image_of(e)));
        if (e->source_half) {   // refresh this slab's slot of the fp16 source copy (the all-gather send slot)
            rc = refresh_half_sources(e, e->lo, slab);
            if (rc != NBX_OK) return rc;
        }
    }
    e->host_pos_valid = false;
    e->host_vel_valid = false;
    if (log_enabled())
        std::fprintf(stderr, "[nbx] step_brute_force dev=%d n=%d slab=[%d,%d) dt=%g mode=%s variant=%d grid=%d S=%d B=%d dim=%d%s\n", e->device,
                     e->n, e->lo, e->hi, (double)dt, e->force_mode ? "strict" : "fast", e->last.variant, e->last.grid, e->last.jsplit,
                     e->last.bpt, e->last.dim, e->source_half ? " fp16-sources" : "");
    return NBX_OK;
}

// host tree (reference-faithful) -> flatten -> device.  `also` (single-process multi-GPU group): further engines that
// hold the same bodies on other devices and receive the same node array, so the tree is built once per step, not once
// per device.
int build_and_upload_tree(nbx_engine* e, nbx_engine* const* also, int n_also, bool order_bodies)
{
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clk::now();
    int rc = NBX_OK;
    HIP_TRY(hipSetDevice(e->device));
    const float *bx = e->host.px.data(), *by = e->host.py.data();
    if (!e->host_pos_valid) {
        // the build needs (x, y) only (masses never change): the device writes them as planar arrays into pinned host
        // memory; the full host mirror is refreshed lazily by whoever asks for it (get_particles, host draw, ...)
        if ((size_t)e->n > e->h_xy_cap) {
            if (e->h_xy) HIP_TRY(hipHostFree(e->h_xy));
            e->h_xy = nullptr;
            e->h_xy_cap = 0;
            const size_t want = std::max<size_t>((size_t)e->n + (size_t)e->n / 8, 1024);
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_xy), sizeof(float) * 2 * want, hipHostMallocDefault));
            e->h_xy_cap = want;
        }
        HIP_TRY(nbx::launch_split_xy(e->d_posm, e->n, e->h_xy, e->h_xy + e->h_xy_cap, e->stream));
        if (!e->ev_xy) HIP_TRY(hipEventCreateWithFlags(&e->ev_xy, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(e->ev_xy, e->stream));
        if (order_bodies) {   // GPU work that overlaps the host build
            rc = spatial_order(e);
            if (rc != NBX_OK) return rc;
            order_bodies = false;
        }
        HIP_TRY(hipEventSynchronize(e->ev_xy));
        // into the (cacheable) host mirror: the build makes several scattered passes over the positions, which is
        // slow straight out of the pinned, device-visible allocation
        const float* sx = e->h_xy;
        const float* sy = e->h_xy + e->h_xy_cap;
        float* dx = e->host.px.data();
        float* dy = e->host.py.data();
        const size_t n = (size_t)e->n;
        if (n >= 262144) {
            nbx::parallel_for(4, [&](int q) {
                const size_t a = (q & 1) ? n / 2 : 0, b = (q & 1) ? n : n / 2;
                std::memcpy((q < 2 ? dx : dy) + a, (q < 2 ? sx : sy) + a, sizeof(float) * (b - a));
            });
        } else {
            std::memcpy(dx, sx, sizeof(float) * n);
            std::memcpy(dy, sy, sizeof(float) * n);
        }
    }
    if (order_bodies) {
        rc = spatial_order(e);
        if (rc != NBX_OK) return rc;
    }
    const auto t1 = clk::now();
    // bigger systems: the routing of the bodies to the top tree's buckets and their stable scatter run on the device
    // (bh_build.hip) while the host threads fold; any device-side problem just leaves both to the host
    nbx::QuadTree::RouteFn route = [e](const nbx::QuadTree::TopView& v, int warm, int rest, int* pbucket,
                                       nbx::QuadTree::Event* sorted, size_t* offset) -> bool {
        struct TopRec { float x1, y1, x2, y2; int32_t first_child, bucket; };
        static_assert(sizeof(TopRec) == 24 && sizeof(nbx::QuadTree::Event) == 16, "layout shared with bh_build.hip");
        std::vector<TopRec> top((size_t)v.ntop);
        for (int k = 0; k < v.ntop; k++)
            top[(size_t)k] = TopRec{v.top[k].x1, v.top[k].y1, v.top[k].x2, v.top[k].y2, v.top[k].first_child, v.bucket_of[k]};
        const size_t need_ws = nbx::device_route_workspace_bytes(rest, v.ntop, v.nb);
        if (need_ws > e->route_ws_bytes) {
            if (e->d_route_ws && hipFree(e->d_route_ws) != hipSuccess) return false;
            e->d_route_ws = nullptr;
            e->route_ws_bytes = 0;
            if (hipMalloc(&e->d_route_ws, need_ws + need_ws / 8) != hipSuccess) return false;
            e->route_ws_bytes = need_ws + need_ws / 8;
        }
        const size_t ev_bytes = ((size_t)rest * 16 + 255) & ~(size_t)255, pb_bytes = ((size_t)rest * 4 + 255) & ~(size_t)255;
        const size_t need_host = ev_bytes + pb_bytes + ((size_t)v.nb + 1) * 8;
        if (need_host > e->h_route_bytes) {
            if (e->h_route && hipHostFree(e->h_route) != hipSuccess) return false;
            e->h_route = nullptr;
            e->h_route_bytes = 0;
            if (hipHostMalloc(reinterpret_cast<void**>(&e->h_route), need_host + need_host / 8, hipHostMallocDefault) != hipSuccess) return false;
            e->h_route_bytes = need_host + need_host / 8;
        }
        char* ev_host = e->h_route;
        int* pb_host = reinterpret_cast<int*>(e->h_route + ev_bytes);
        unsigned long long* off_host = reinterpret_cast<unsigned long long*>(e->h_route + ev_bytes + pb_bytes);
        if (nbx::device_route_and_scatter(e->d_posm, warm, rest, top.data(), v.ntop, v.bucket_depth, v.nb, e->d_route_ws,
                                          e->route_ws_bytes, pb_host, ev_host, off_host, e->stream) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
        // out of the pinned allocation into the build's own (cacheable) arrays
        nbx::parallel_for(8, [&](int t) {
            const size_t a = (size_t)rest * t / 8, b = (size_t)rest * (t + 1) / 8;
            std::memcpy(sorted + a, ev_host + a * 16, (b - a) * 16);
            std::memcpy(pbucket + a, pb_host + a, (b - a) * 4);
        });
        for (int b = 0; b <= v.nb; b++) offset[b] = (size_t)off_host[b];
        return true;
    };
    // (from 16 384 bodies on: the host's own routing + scatter is 0.9 of the build's 2.0 ms at 65 536 bodies -- host-tree step
    //  3.1 -> 2.2 ms there, 1.27 -> 1.10 at 20 000, 0.84 -> 0.89 at 10 000; round 2 used it from 262 144 bodies only)
    const bool device_routes = e->dev_valid && e->n >= 16384;
    rc = e->tree.build(bx, by, e->host.m.data(), e->n, /*preflatten=*/true, device_routes ? &route : nullptr);
    if (rc == NBX_ERR_TREE_DEPTH) return fail(rc, "quadtree depth > 50 (the reference panics here, nbody.rs:230-232)");
    if (rc != NBX_OK) return fail(rc, "quadtree build hit a reference assert (nbody.rs:267/:293/:304)");
    const auto t2 = clk::now();
    const bool big = e->tree.forest;
    // upper bound of the flattened size before it is known exactly (the sequential tree is written in one pass)
    size_t count = big ? e->tree.flatten_prepare(e->plan) : e->tree.nodes.size();
    if (count > e->h_nodes_cap) {
        if (e->h_nodes) HIP_TRY(hipHostFree(e->h_nodes));
        e->h_nodes = nullptr;
        e->h_nodes_cap = 0;
        const size_t want = std::max<size_t>(count + count / 4, 1024);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_nodes), sizeof(nbx::BhNode) * want, hipHostMallocPortable));   // read by every device of a group
        e->h_nodes_cap = want;
    }
    if (!big) count = e->tree.flatten_into(e->h_nodes);
    std::vector<nbx_engine*> dst{e};
    for (int i = 0; i < n_also; i++) dst.push_back(also[i]);
    for (nbx_engine* d : dst) {
        HIP_TRY(hipSetDevice(d->device));
        rc = grow(&d->d_nodes, &d->nodes_cap, std::max<size_t>(count, 1));
        if (rc != NBX_OK) return rc;
    }
    hipError_t copy_err = hipSuccess;
    auto send = [&](size_t a, size_t b) {
        for (nbx_engine* d : dst) {
            hipError_t ce = dst.size() > 1 ? hipSetDevice(d->device) : hipSuccess;
            if (ce == hipSuccess)
                ce = hipMemcpyAsync(d->d_nodes + a, e->h_nodes + a, sizeof(nbx::BhNode) * (b - a), hipMemcpyHostToDevice, d->stream);
            if (ce != hipSuccess && copy_err == hipSuccess) copy_err = ce;
        }
    };
    if (big) {
        // the host-to-device copy of every finished prefix of the array starts while the rest is still being written
        e->tree.flatten_write(e->plan, e->h_nodes, send);
    } else if (count) {
        send(0, count);
    }
    for (nbx_engine* d : dst) d->n_flat = count;
    const auto t3 = clk::now();
    HIP_TRY(copy_err);
    if (count)
        for (nbx_engine* d : dst) {   // the staging buffer is rewritten next step
            HIP_TRY(hipSetDevice(d->device));
            HIP_TRY(hipStreamSynchronize(d->stream));
        }
    HIP_TRY(hipSetDevice(e->device));
    const auto t4 = clk::now();
    e->host_ms[0] += ms(t0, t1); e->host_ms[1] += ms(t1, t2); e->host_ms[2] += ms(t2, t3); e->host_ms[3] += ms(t3, t4);
    e->host_steps++;
    return NBX_OK;
}

// node slots are 32-bit and a body can own up to 32 nodes: beyond this size the host build is used
static constexpr int kDeviceTreeMaxBodies = 1 << 25;
static constexpr int kBackoffMaxSteps = 32;   // see nbx_engine::note_refusal

// quadtree on the device (bh_build.hip), in two halves so that a group can start every device's build before it waits
// for any: begin enqueues the build, end waits for it. *done = false when the node pool overflowed (the caller falls
// back to the host build).
int build_tree_on_device_begin(nbx_engine* e, int* host_counters, bool publish_by_kernel)
{
    HIP_TRY(hipSetDevice(e->device));
    e->tree_t0 = std::chrono::steady_clock::now();
    const int node_cap = 4 * e->n + 1024;
    size_t sort_tmp = 0;
    const size_t need = nbx::device_tree_workspace_bytes(e->n, node_cap, &sort_tmp);
    if (need > e->tree_ws_bytes) {
        if (e->d_tree_ws) HIP_TRY(hipFree(e->d_tree_ws));
        e->d_tree_ws = nullptr;
        e->tree_ws_bytes = 0;
        HIP_TRY(hipMalloc(&e->d_tree_ws, need));
        e->tree_ws_bytes = need;
        e->sort_warm_n = 0;
        HIP_TRY(nbx::device_tree_workspace_init(e->d_tree_ws, e->stream));
    }
    if (!e->h_counters) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_counters), 64, hipHostMallocDefault));
    const int rc = grow(&e->d_nodes, &e->nodes_cap, (size_t)node_cap);
    if (rc != NBX_OK) return rc;
    const int fold = e->effective_fold();
    if (fold == 1 && !e->side_stream && !std::getenv("NBX_NO_SIDE_STREAM")) {
        HIP_TRY(hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_side_go, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_side_done, hipEventDisableTiming));
    }
    ProfScope ps(e, NBX_K_TREE_BUILD);
    HIP_TRY(nbx::device_tree_build_begin(e->d_posm, e->n, e->d_tree_ws, e->tree_ws_bytes, node_cap, e->d_nodes,
                                         publish_by_kernel ? nullptr : (host_counters ? host_counters : e->h_counters), &e->d_perm,
                                         e->stream, fold, e->side_stream, e->ev_side_go, e->ev_side_done,
                                         /*depth_panic_guard=*/e->force_mode != 0, /*warm=*/e->sort_warm_n == e->n));
    e->sort_warm_n = e->n;   // (a refusal -- of this build, or of one whose verdict is still in flight -- takes it back)
    return NBX_OK;
}

// longest run of steps a back-off sends straight to the host build (engine_internal.h); NBX_BH_BACKOFF_MAX=0 turns it off
static int backoff_max_steps()
{
    static const int v = [] {
        const char* s = std::getenv("NBX_BH_BACKOFF_MAX");
        const int x = s ? std::atoi(s) : kBackoffMaxSteps;
        return x < 0 ? 0 : x;
    }();
    return v;
}

int build_tree_on_device_end(nbx_engine* e, bool* done)
{
    *done = false;
    HIP_TRY(hipSetDevice(e->device));
    const int node_cap = 4 * e->n + 1024;
    int n_nodes = 0, status = 0;
    HIP_TRY(nbx::device_tree_build_end(e->n, node_cap, e->h_counters, &n_nodes, &status, e->stream, e->effective_fold()));
    if (status != 0) {
        e->bh_fallbacks++;
        e->note_refusal(backoff_max_steps());
        e->note_why(status, e->h_counters[5]);
        e->d_perm = nullptr;
        e->sort_warm_n = 0;
        if (std::getenv("NBX_LOG"))
            std::fprintf(stderr, "[nbx] device tree build of %d bodies handed over to the host build: status %d (1 = pool / queue overflow, 2 = EPS "
                                 "clusters), nodes %d of %d, left-behind bodies %d (why 0x%x), queued folds %d\n", e->n, status, e->h_counters[0], node_cap,
                         e->h_counters[1], (unsigned)e->h_counters[5], e->h_counters[2]);
        return NBX_OK;   // caller takes the host path
    }
    e->note_accepted();
    e->n_flat = (size_t)n_nodes;
    e->host_ms[1] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - e->tree_t0).count();
    e->host_steps++;
    *done = true;
    return NBX_OK;
}

int build_tree_on_device(nbx_engine* e, bool* done)
{
    *done = false;
    if (e->n > kDeviceTreeMaxBodies) return NBX_OK;   // caller takes the host path
    const int rc = build_tree_on_device_begin(e);
    if (rc != NBX_OK) return rc;
    return build_tree_on_device_end(e, done);
}

// Morton permutation of the bodies on the device (for the traversal of a host-built tree)
int spatial_order(nbx_engine* e)
{
    size_t sort_tmp = 0;
    const size_t need = nbx::device_tree_workspace_bytes(e->n, 1, &sort_tmp);
    if (need > e->tree_ws_bytes) {
        if (e->d_tree_ws) HIP_TRY(hipFree(e->d_tree_ws));
        e->d_tree_ws = nullptr;
        e->tree_ws_bytes = 0;
        HIP_TRY(hipMalloc(&e->d_tree_ws, need));
        e->tree_ws_bytes = need;
        e->sort_warm_n = 0;
        HIP_TRY(nbx::device_tree_workspace_init(e->d_tree_ws, e->stream));
    }
    HIP_TRY(nbx::device_spatial_order(e->d_posm, e->n, e->d_tree_ws, e->tree_ws_bytes, &e->d_perm, e->stream, e->sort_warm_n == e->n));
    e->sort_warm_n = e->n;
    return NBX_OK;
}

// world > 1: the Morton order of the bodies (e->d_perm, all n of them) restricted to this engine's slab
int slab_order(nbx_engine* e)
{
    e->d_slab_perm = nullptr;
    if (!e->d_perm || e->slab() == 0) return NBX_OK;
    const size_t need = nbx::device_slab_order_workspace_bytes(e->n);
    if (need > e->slab_ws_bytes) {
        if (e->d_slab_ws) HIP_TRY(hipFree(e->d_slab_ws));
        e->d_slab_ws = nullptr;
        e->slab_ws_bytes = 0;
        HIP_TRY(hipMalloc(&e->d_slab_ws, need));
        e->slab_ws_bytes = need;
    }
    HIP_TRY(nbx::device_slab_order(e->d_perm, e->n, e->lo, e->hi, e->d_slab_ws, e->slab_ws_bytes, &e->d_slab_perm, e->stream));
    return NBX_OK;
}

// May the child-group walk of this step apply the kick-drift itself (kernels.h BhKick)?  Only the wave form has it, on one GPU
// (a group's exchange reads the kick-drift's output slab by slab).  One dependent kernel less per step: 0.0932 -> 0.0899 ms at
// 10 000 bodies, 0.8375 -> 0.8252 at 1 M (the walk itself +0.002 ms there, the 0.018 ms kick-drift kernel gone).
bool walk_takes_kick(const nbx_engine* e, const unsigned* perm, bool wave, int nodes_or_cap)
{
    if (!e->bh_fuse_kick || e->force_mode != 0 || e->world != 1 || e->source_half || e->bh_walk == 0 || !(wave && perm)) return false;
    return nbx::bh_groups_addressable(nodes_or_cap);
}

// Fast-mode traversal of the node array this engine holds (e->d_nodes, e->n_flat -- or, gated, the count the device build left in
// its counters) for the slab's bodies, accelerations into e->d_f2 -- or, with kick, straight into the bodies' velocities and
// positions.  NBX_OPT_BH_WALK = 1 (default) / 2: the tree is first re-laid as child groups with the step's theta (k_bh_groups),
// then walked group by group (bh_walk.hip: hand-scheduled / compiled loop); 0 (or a tree too large for 31-bit record offsets:
// beyond ~4 M bodies): the node walk of rounds 1-3.
int launch_fast_walk(nbx_engine* e, float theta, const unsigned* perm, bool wave, bool on_device, int* gate, int gate_node_cap,
                     int gate_crowd_limit, int gate_queue_limit, const nbx::BhKick* kick)
{
    const int slab = e->slab();
    ProfScope ps(e, NBX_K_BH_EVAL);
    const int nodes_or_cap = gate ? gate_node_cap : (int)e->n_flat;
    if (e->bh_walk != 0 && nbx::bh_groups_addressable(nodes_or_cap)) {
        const int rc = grow(&e->d_groups, &e->groups_cap, nbx::bh_groups_count(nodes_or_cap));
        if (rc != NBX_OK) return rc;
        HIP_TRY(nbx::launch_bh_groups(e->d_nodes, nodes_or_cap, theta, e->d_groups, /*compact=*/on_device, e->stream, gate, gate_node_cap,
                                      gate_crowd_limit, gate_queue_limit));
        // EXPERIMENT (NBX_WALK_SPLIT_PCT = p > 0; off by default): two rounds of walks or more (> 8192 of them on the chip's 8192 wave
        // slots) -- the p percent that loaded the most groups in the previous step run as two halves of 32 bodies, in Morton order
        // (bh_walk.hip k_walk_split_list; a stale or missing list costs time, never a result)
        const int* list = nullptr;
        int* cost = nullptr;
        int bpw = 0;
        int walks = (wave && perm && e->bh_walk == 1 && e->walk_split_pct > 0) ? nbx::bh_walk_count(slab, &bpw) : 0;
        if (bpw != 64) walks = 0;          // (halves are halves of 64-body walks)
        const int budget = walks > 8192 ? (int)((long long)walks * e->walk_split_pct / 100) : 0;
        if (walks > 8192) {
            const size_t had = e->walk_cost_cap;
            int rc2 = grow(&e->d_walk_cost, &e->walk_cost_cap, 2 * (size_t)walks);
            if (rc2 == NBX_OK) rc2 = grow(&e->d_walk_list, &e->walk_list_cap, (size_t)walks + (size_t)budget + 8);
            if (rc2 != NBX_OK) return rc2;
            if (e->walk_cost_cap != had || e->walk_list_walks != walks || e->walk_list_slab != slab) {
                // fresh memory or another shape: walks that write no cost (no body of theirs in the slab; a refused, gated step) read as zero
                HIP_TRY(hipMemsetAsync(e->d_walk_cost, 0, sizeof(int) * e->walk_cost_cap, e->stream));
                e->walk_list_walks = 0;
                e->walk_cost_flip = 0;
            }
            cost = e->d_walk_cost + (e->walk_cost_flip ? walks : 0);
            if (e->walk_list_walks == walks && e->walk_list_slab == slab) list = e->d_walk_list;
        }
        if (e->d_walk_trace && wave && perm) e->walk_traced = true;
        HIP_TRY(nbx::launch_bh_walk_groups(e->d_posm, e->lo, slab, e->d_groups, e->d_f2, e->stream, perm, wave, e->bh_walk == 1, gate,
                                           gate_node_cap, gate_crowd_limit, gate_queue_limit, list, cost, e->d_walk_trace, kick, budget));
        if (cost) {
            int* next = e->d_walk_cost + (e->walk_cost_flip ? 0 : walks);
            HIP_TRY(nbx::launch_walk_split_list(cost, next, e->d_walk_list, walks, budget, e->stream));
            e->walk_cost_flip ^= 1;
            e->walk_list_walks = walks;
            e->walk_list_slab = slab;
        }
        return NBX_OK;
    }
    if (kick) return fail(NBX_ERR_STATE, "kick-drift handed to a walk that cannot apply it");
    HIP_TRY(nbx::launch_bh_eval(e->d_posm, e->lo, slab, e->d_nodes, (int)e->n_flat, theta, wave ? 2 : 0, e->d_f2, e->stream, perm, gate,
                                gate_node_cap, gate_crowd_limit, gate_queue_limit));
    return NBX_OK;
}

// traversal + kick-drift of this engine's slab against the node array it holds (e->d_nodes, e->n_flat)
int bh_eval_and_integrate(nbx_engine* e, float theta, float dt, bool on_device, bool have_perm, bool gated, int* gate_host_out)
{
    HIP_TRY(hipSetDevice(e->device));
    int* gate = nullptr;
    int node_cap = 0, crowd_limit = 0, queue_limit = 0;
    if (gated) {   // the device build's verdict is still on the device: the kernels check it themselves
        gate = nbx::device_tree_counters(e->d_tree_ws);
        node_cap = 4 * e->n + 1024;
        nbx::device_tree_limits(e->n, e->effective_fold(), &crowd_limit, &queue_limit);
    }
    const int slab = e->slab();
    if (slab == 0) return NBX_OK;
    int rc = grow(&e->d_f2, &e->f2_cap, (size_t)slab);
    if (rc != NBX_OK) return rc;
    const unsigned* perm = nullptr;
    if (have_perm) {   // a Morton order helps the per-lane walks too (NBX_OPT_BH_WAVE = 0 only turns the shared walk off)
        if (e->world == 1) {
            perm = e->d_perm;
        } else {   // several GPUs share the bodies: this engine's part of the Morton order
            rc = slab_order(e);
            if (rc != NBX_OK) return rc;
            perm = e->d_slab_perm;
        }
    }
    e->bh_last_tree_device = on_device ? 1 : 0;
    const bool wave = perm != nullptr && e->bh_wave;   // shared walk per wave, in both modes (same results as the per-lane walks)
    bool kicked = false;
    if (e->force_mode == 0) {
        const nbx::BhKick kick{e->d_vel, e->d_posm, dt, gated ? gate_host_out : nullptr};
        kicked = walk_takes_kick(e, perm, wave, gate ? node_cap : (int)e->n_flat);
        rc = launch_fast_walk(e, theta, perm, wave, on_device, gate, node_cap, crowd_limit, queue_limit, kicked ? &kick : nullptr);
        if (rc != NBX_OK) return rc;
    } else {
        ProfScope ps(e, NBX_K_BH_EVAL);
        HIP_TRY(nbx::launch_bh_eval(e->d_posm, e->lo, slab, e->d_nodes, (int)e->n_flat, theta, wave ? 3 : e->force_mode, e->d_f2,
                                    e->stream, perm));
    }
    if (!kicked) {
        ProfScope ps(e, NBX_K_INTEGRATE);
        HIP_TRY(nbx::launch_integrate_f2(e->d_posm, e->lo, slab, e->d_vel, e->d_f2, dt, e->force_mode == 0 ? 1 : 0, 1,
                                         e->stream, gate, node_cap, crowd_limit, queue_limit, gated ? gate_host_out : nullptr));
    }
    if (e->source_half && !gated) {
        rc = refresh_half_sources(e, e->lo, slab);
        if (rc != NBX_OK) return rc;
    }
    e->host_pos_valid = false;
    e->host_vel_valid = false;
    if (log_enabled())
        std::fprintf(stderr, "[nbx] step_barnes_hut dev=%d n=%d slab=[%d,%d) theta=%g dt=%g mode=%s tree=%s nodes=%zu walk=%s\n", e->device, e->n,
                     e->lo, e->hi, (double)theta, (double)dt, e->force_mode ? "strict" : "fast", on_device ? "device" : "host", e->n_flat,
                     wave ? "wave" : "lane");
    return NBX_OK;
}

// ---- Barnes-Hut steps without a host wait in the middle (NBX_OPT_BH_ASYNC) --------------------------------------------------
// A step on the device tree = build + walk + kick-drift, all enqueued at once: walk and kick-drift check the build's verdict
// (node count, EPS clusters) on the device (bh_eval.hip BuildGate) and leave the state untouched when the build had to refuse;
// the kick-drift then raises a device flag ("poison") that makes every later gated kernel do nothing as well.  The host reads a
// step's verdict only AFTER it has enqueued the next step (two slots), so neither a wait in the middle of a step nor one
// between steps leaves the GPU idle.  A refused step (rare: EPS clusters, exhausted node pool) is redone on the host tree once
// its verdict is read, and the step enqueued behind it -- which the flag turned into a no-op -- is enqueued again.
static int verdict_of(const nbx_engine* e, int slot)
{
    const int* c = e->h_verdict[slot];
    int crowd = 0, queue = 0;
    nbx::device_tree_limits(e->n, e->pending[slot].fold, &crowd, &queue);
    if (c[0] > e->pending[slot].node_cap) return 1;
    if (c[1] > crowd) return 2;
    if (c[2] > queue) return 1;
    return 0;
}

static int resolve_slot(nbx_engine* e, int slot)
{
    if (!e->pending[slot].active) return NBX_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(wait_event(e->ev_step[slot]));
    const nbx_engine::PendingStep p = e->pending[slot];
    e->pending[slot].active = false;
    const int status = verdict_of(e, slot);
    if (status == 0) {
        e->note_accepted();
        e->n_flat = (size_t)e->h_verdict[slot][0];
        e->bh_last_tree_device = 1;
        e->host_steps++;
        return NBX_OK;
    }
    // refused: this step's gated kernels did nothing and poisoned the step behind it (if one is in flight)
    e->bh_fallbacks++;
    e->note_refusal(backoff_max_steps());
    e->note_why(status, e->h_verdict[slot][5]);
    e->d_perm = nullptr;
    e->sort_warm_n = 0;
    const int other = slot ^ 1;
    const bool redo_later = e->pending[other].active;
    const nbx_engine::PendingStep later = e->pending[other];
    e->pending[other].active = false;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (std::getenv("NBX_LOG"))
        std::fprintf(stderr, "[nbx] device tree build of %d bodies refused (status %d: nodes %d of %d, left-behind bodies %d (why 0x%x), queued folds %d): "
                             "step redone on the host tree%s\n", e->n, status, e->h_verdict[slot][0], p.node_cap, e->h_verdict[slot][1],
                     (unsigned)e->h_verdict[slot][5], e->h_verdict[slot][2], redo_later ? ", the step behind it enqueued again" : "");
    HIP_TRY(hipMemsetAsync(nbx::device_tree_counters(e->d_tree_ws) + nbx::kTreePoisonWord, 0, sizeof(int), e->stream));
    const bool want_order = e->bh_wave && e->n >= 65536;
    int rc = build_and_upload_tree(e, nullptr, 0, want_order);
    if (rc != NBX_OK) return rc;
    rc = bh_eval_and_integrate(e, p.theta, p.dt, false, want_order && e->d_perm != nullptr);
    if (rc != NBX_OK) return rc;
    return redo_later ? step_bh(e, later.theta, later.dt) : NBX_OK;
}

int resolve_pending(nbx_engine* e)
{
    for (int k = 0; k < 2; k++) {   // oldest first: pend_next is the slot the next step would take, i.e. the older one
        const int rc = resolve_slot(e, e->pend_next ^ (k & 1));
        if (rc != NBX_OK) return rc;
    }
    return NBX_OK;
}

static int step_bh_async(nbx_engine* e, float theta, float dt)
{
    HIP_TRY(hipSetDevice(e->device));
    for (int s = 0; s < 2; s++)
        if (!e->h_verdict[s]) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_verdict[s]), 64, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&e->ev_step[s], hipEventDisableTiming));
        }
    const int slot = e->pend_next;
    int rc = resolve_slot(e, slot);                 // the slot must be free (only the case when two steps are already in flight)
    if (rc != NBX_OK) return rc;
    rc = build_tree_on_device_begin(e, e->h_verdict[slot], /*publish_by_kernel=*/true);   // the gated kick-drift hands the counters over
    if (rc != NBX_OK) return rc;
    e->pending[slot].theta = theta; e->pending[slot].dt = dt;
    e->pending[slot].node_cap = 4 * e->n + 1024;
    e->pending[slot].fold = e->effective_fold();
    rc = bh_eval_and_integrate(e, theta, dt, true, true, /*gated=*/true, e->h_verdict[slot]);
    if (rc != NBX_OK) return rc;
    HIP_TRY(hipEventRecord(e->ev_step[slot], e->stream));
    e->pending[slot].active = true;
    e->pend_next = slot ^ 1;
    return resolve_slot(e, slot ^ 1);               // the step BEFORE this one: its verdict is (nearly) there by now
}

int step_bh(nbx_engine* e, float theta, float dt)
{
    int rc = NBX_OK;
    if (e->any_pending() && e->bh_refusal_streak > 0) {   // a verdict that may start a back-off: read it before choosing the path
        rc = resolve_pending(e);
        if (rc != NBX_OK) return rc;
    }
    bool device_tree = e->use_device_tree() && e->n <= kDeviceTreeMaxBodies;
    if (device_tree && e->bh_host_steps_left > 0) {       // back-off after refusals in a row (engine_internal.h)
        e->bh_host_steps_left--;
        e->bh_fallbacks++;
        device_tree = false;
    }
    const bool async_ok = e->bh_async && e->world == 1 && !e->source_half && device_tree && e->force_mode == 0;
    if (!(async_ok && e->dev_ready && e->dev_valid && e->n > 0)) {   // (a live device state needs no upload, and no verdict read)
        rc = upload(e);
        if (rc != NBX_OK) return rc;
    }
    if (e->n == 0) return NBX_OK;
    if (async_ok) return step_bh_async(e, theta, dt);
    bool on_device = false;
    if (device_tree) {
        rc = build_tree_on_device(e, &on_device);
        if (rc != NBX_OK) return rc;
    }
    bool have_perm = on_device;
    if (!on_device) {
        // host tree, big system: a Morton order of the bodies (0.4 ms at 1 M) makes the walk wave-coherent and lets the
        // fast mode take the wave-uniform form (4.4 -> 0.64 ms). Results are unaffected. The sort only reads the
        // positions: it is enqueued right after the (x, y) download, so the GPU does it while the host builds the tree.
        const bool want_order = e->bh_wave && e->n >= 65536;
        rc = build_and_upload_tree(e, nullptr, 0, want_order);
        if (rc != NBX_OK) return rc;
        have_perm = want_order && e->d_perm != nullptr;
    }
    return bh_eval_and_integrate(e, theta, dt, on_device, have_perm);
}

// One Barnes-Hut step of a single-process group: every engine holds the same bodies (positions replicated by the
// per-step all-gather), so the quadtree is built ONCE -- on the host from engine 0's copy and sent to every device, or
// on every device concurrently (all builds are enqueued before any is waited for) -- and each engine evaluates its slab.
int step_bh_group(nbx_engine* const* eng, int count, float theta, float dt)
{
    if (count == 1) return step_bh(eng[0], theta, dt);
    for (int d = 0; d < count; d++) {
        const int rc = upload(eng[d]);
        if (rc != NBX_OK) return rc;
    }
    nbx_engine* e0 = eng[0];
    if (e0->n == 0) return NBX_OK;
    bool on_device = e0->use_device_tree() && e0->n <= kDeviceTreeMaxBodies;
    if (on_device) {
        for (int d = 0; d < count; d++) {
            const int rc = build_tree_on_device_begin(eng[d]);
            if (rc != NBX_OK) return rc;
        }
        for (int d = 0; d < count; d++) {
            bool done = false;
            const int rc = build_tree_on_device_end(eng[d], &done);
            if (rc != NBX_OK) return rc;
            if (!done) on_device = false;   // same bodies, same tree: if one pool overflows, all do
        }
    }
    if (!on_device) {
        const int rc = build_and_upload_tree(e0, eng + 1, count - 1);
        if (rc != NBX_OK) return rc;
    }
    for (int d = 0; d < count; d++) {
        nbx_engine* e = eng[d];
        bool have_perm = on_device;
        if (!on_device && e->bh_wave && e->n >= 65536) {   // as in step_bh: Morton order for the walk
            HIP_TRY(hipSetDevice(e->device));
            const int rc = spatial_order(e);
            if (rc != NBX_OK) return rc;
            have_perm = e->d_perm != nullptr;
        }
        const int rc = bh_eval_and_integrate(e, theta, dt, on_device, have_perm);
        if (rc != NBX_OK) return rc;
    }
    return NBX_OK;
}

void free_device(nbx_engine* e)
{
    if (!e->dev_ready) return;
    e->pending[0].active = e->pending[1].active = false;
    // everything in flight first -- a gated kick-drift of an asynchronous Barnes-Hut step may still be about to write its
    // verdict into h_verdict[], the root's fold may still run on the side stream -- THEN the buffers and events they use
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->side_stream) (void)hipStreamSynchronize(e->side_stream);
    for (int k = 0; k < 2; k++) {
        if (e->h_verdict[k]) (void)hipHostFree(e->h_verdict[k]);
        if (e->ev_step[k]) (void)hipEventDestroy(e->ev_step[k]);
        e->h_verdict[k] = nullptr;
        e->ev_step[k] = nullptr;
    }
    for (auto& r : e->prof) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    e->prof.clear();
    for (hipEvent_t ev : e->ev_free) (void)hipEventDestroy(ev);
    e->ev_free.clear();
    if (e->d_posm && !e->posm_external) (void)hipFree(e->d_posm);
    if (e->d_vel) (void)hipFree(e->d_vel);
    if (e->d_acc) (void)hipFree(e->d_acc);
    if (e->d_f2) (void)hipFree(e->d_f2);
    if (e->d_out4) (void)hipFree(e->d_out4);
    if (e->d_nodes) (void)hipFree(e->d_nodes);
    if (e->d_groups) (void)hipFree(e->d_groups);
    if (e->d_walk_cost) (void)hipFree(e->d_walk_cost);
    if (e->d_walk_list) (void)hipFree(e->d_walk_list);
    if (e->d_guard) (void)hipFree(e->d_guard);
    if (e->d_exc_idx) (void)hipFree(e->d_exc_idx);
    if (e->d_src4) (void)hipFree(e->d_src4);
    if (e->d_exc_rec) (void)hipFree(e->d_exc_rec);
    if (e->d_tree_ws) (void)hipFree(e->d_tree_ws);
    if (e->d_slab_ws) (void)hipFree(e->d_slab_ws);
    if (e->h_counters) (void)hipHostFree(e->h_counters);
    if (e->side_stream) (void)hipStreamDestroy(e->side_stream);
    if (e->ev_side_go) (void)hipEventDestroy(e->ev_side_go);
    if (e->ev_side_done) (void)hipEventDestroy(e->ev_side_done);
    if (e->d_counts) (void)hipFree(e->d_counts);
    if (e->d_fb) (void)hipFree(e->d_fb);
    if (e->h_fb) (void)hipHostFree(e->h_fb);
    if (e->d_amb) (void)hipFree(e->d_amb);
    if (e->d_posh && !e->posh_external) (void)hipFree(e->d_posh);
    if (e->h_nodes) (void)hipHostFree(e->h_nodes);
    if (e->h_stage) (void)hipHostFree(e->h_stage);
    if (e->h_xy) (void)hipHostFree(e->h_xy);
    if (e->ev_xy) (void)hipEventDestroy(e->ev_xy);
    if (e->d_route_ws) (void)hipFree(e->d_route_ws);
    if (e->h_route) (void)hipHostFree(e->h_route);
    if (e->stream && e->own_stream) (void)hipStreamDestroy(e->stream);
}

uint64_t entropy_seed()
{
    std::random_device rd;
    return ((uint64_t)rd() << 32) ^ (uint64_t)rd();
}

void after_host_state_change(nbx_engine* e)
{
    if (e->any_pending()) {   // the state is being replaced: whatever the pending steps' verdicts, their results are discarded
        if (e->dev_ready) {
            (void)hipSetDevice(e->device);
            (void)hipStreamSynchronize(e->stream);
            if (e->d_tree_ws) (void)hipMemsetAsync(nbx::device_tree_counters(e->d_tree_ws) + nbx::kTreePoisonWord, 0, sizeof(int), e->stream);
        }
        e->pending[0].active = e->pending[1].active = false;
    }
    e->n = e->host.n();
    e->bh_refusal_streak = e->bh_host_steps_left = 0;
    compute_slab(e);
    e->host_pos_valid = e->host_vel_valid = true;
    e->dev_valid = false;
    e->any_z = false;
    for (int i = 0; i < e->n && !e->any_z; i++)
        if (e->host.pz[i] != 0.0f || e->host.vz[i] != 0.0f) e->any_z = true;
    // mass range of the current bodies; ONE NaN or infinite mass anywhere poisons the range for good (sticky: a running
    // min/max would overwrite it with the next finite mass) so that every range test -- strict_fastdiv_ok -- fails
    bool bad = false;
    float lo = e->n ? e->host.m[0] : 0.0f, hi = lo;
    for (int i = 0; i < e->n; i++) {
        const float m = e->host.m[i];
        if (!(m == m) || std::isinf(m)) bad = true;
        if (m < lo) lo = m;
        if (m > hi) hi = m;
    }
    const float nan = std::numeric_limits<float>::quiet_NaN();
    e->mass_min = bad ? nan : lo;
    e->mass_max = bad ? nan : hi;
    // common mass = the majority value (Boyer-Moore vote, one pass), exceptions = everybody else, if they are few
    e->mass_common = 0.0f;
    e->exc_idx.clear();
    if (!bad && e->n > 0 && lo > 0.0f) {
        float cand = e->host.m[0];
        int votes = 0;
        for (int i = 0; i < e->n; i++) {
            const float m = e->host.m[i];
            if (votes == 0) { cand = m; votes = 1; }
            else if (m == cand) votes++;
            else votes--;
        }
        const int cap = nbx_engine::exc_cap(e->n);
        bool few = true;
        for (int i = 0; i < e->n && few; i++)
            if (e->host.m[i] != cand) {
                if ((int)e->exc_idx.size() >= cap) { few = false; break; }
                e->exc_idx.push_back(i);
            }
        if (few) e->mass_common = cand;
        else e->exc_idx.clear();
    }
}


}  // namespace nbxi
