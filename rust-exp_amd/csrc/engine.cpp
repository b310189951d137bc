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
    e->positions_moved();
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
bool log_enabled()
{
    static const bool on = std::getenv("NBX_LOG") != nullptr;
    return on;
}

int step_brute(nbx_engine* e, float dt)
{
    e->positions_moved();   // (an all-pairs step moves every body: the order-sorted copy of the positions is stale from here)
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
    if (e->d_sorted_pos) (void)hipFree(e->d_sorted_pos);
    if (e->d_nodes) (void)hipFree(e->d_nodes);
    if (e->d_groups) (void)hipFree(e->d_groups);
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
    e->reset_backoff();
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
