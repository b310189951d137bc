// c_api.cpp -- level 2 of the C ABI (include/nbody_mi355x.h): the handle-based nbx_* entry points.
#include <cstdint>
#include <cstring>
#include <new>

#include "engine_internal.h"
#include "bh_threshold.h"

using namespace nbxi;

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
    switch (option) {   // a step that is still in flight was issued under the CURRENT options: settle it (and redo it, if its device
                        // build was refused) before any option that shapes a step changes
        case NBX_OPT_FORCE_MODE: case NBX_OPT_BH_TREE: case NBX_OPT_BH_FOLD: case NBX_OPT_BH_WAVE: case NBX_OPT_BH_ASYNC:
        case NBX_OPT_SOURCE_PRECISION: case NBX_OPT_BH_WALK: case NBX_OPT_BH_FUSE_KICK: {
            const int rc = resolve_pending(e);
            if (rc != NBX_OK) return rc;
            break;
        }
        default: break;
    }
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
            if (value != 0 && value != 2 && value != 4) return fail(NBX_ERR_INVALID, "bodies/thread must be 0 (auto), 2 or 4 (packed pairs)");
            e->bpt = (int)value;
            return NBX_OK;
        case NBX_OPT_DIM:
            if (value != 0 && value != 2 && value != 3) return fail(NBX_ERR_INVALID, "dim must be 0,2,3");
            e->dim_opt = (int)value;
            return NBX_OK;
        case NBX_OPT_PROFILE:
            e->profile = value ? 1 : 0;
            if (e->profile && e->dev_ready && e->ev_free.size() < 64) {   // events exist before anything is timed
                HIP_TRY(hipSetDevice(e->device));
                while (e->ev_free.size() < 64) {
                    hipEvent_t ev = nullptr;
                    HIP_TRY(hipEventCreate(&ev));
                    e->ev_free.push_back(ev);
                }
            }
            return NBX_OK;
        case NBX_OPT_KERNEL_VARIANT:
            // (17 / 18 are what the engine REPORTS for the fp16-source sweeps; K2 and the force readout key off them, so a caller
            //  must not be able to set them)
            if (value != -1 && value != 1 && value != 6 && value != 7)
                return fail(NBX_ERR_INVALID, "kernel variant must be -1 (auto), 1 (LDS tiles), 6 or 7 (scalar-cache sweep, wave split)");
            e->variant = (int)value;
            return NBX_OK;
        case NBX_OPT_STRICT_KERNEL:
            if (value != 0 && value != 1 && value != 8 && value != 16) return fail(NBX_ERR_INVALID, "strict kernel must be 0 (auto), 1, 8 or 16");
            e->strict_kernel = (int)value;
            return NBX_OK;
        case NBX_OPT_DRAW_DEVICE:
            e->draw_device = value < 0 ? -1 : (value ? 1 : 0);   // -1 = by size (default)
            return NBX_OK;
        case NBX_OPT_BH_WAVE:
            e->bh_wave = value ? 1 : 0;
            return NBX_OK;
        case NBX_OPT_BH_TREE:
            if (value != 0 && value != 1 && value != -1) return fail(NBX_ERR_INVALID, "bh tree must be 0 (host), 1 (device) or -1 (by mode and size)");
            e->bh_tree_device = value < 0 ? -1 : (value ? 1 : 0);   // -1 = by mode and size (default)
            return NBX_OK;
        case NBX_OPT_BH_ASYNC:
            e->bh_async = value ? 1 : 0;
            return NBX_OK;
        case NBX_OPT_BH_WALK:
            if (value != 0 && value != 1 && value != 2)
                return fail(NBX_ERR_INVALID, "bh walk must be 1 (child groups), 2 (child groups, compiled loop) or 0 (node walk)");
            e->bh_walk = (int)value;
            return NBX_OK;
        case NBX_OPT_BH_FUSE_KICK:
            if (value != 0 && value != 1) return fail(NBX_ERR_INVALID, "bh fuse kick must be 0 or 1");
            e->bh_fuse_kick = (int)value;
            return NBX_OK;
        case NBX_OPT_BH_FOLD:
            if (value != 0 && value != 1 && value != -1) return fail(NBX_ERR_INVALID, "bh fold must be 0 (exact sums), 1 (reference fold) or -1 (by size)");
            e->bh_fold = (int)value;
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
        case NBX_OPT_STRICT_KERNEL: return e->strict_kernel;
        case NBX_OPT_SOURCE_PRECISION: return e->source_half ? 16 : 32;
        case NBX_OPT_DRAW_DEVICE: return e->draw_device;
        case NBX_OPT_BH_TREE: return e->bh_tree_device;
        case NBX_OPT_BH_WAVE: return e->bh_wave;
        case NBX_OPT_BH_FOLD: return e->bh_fold;
        case NBX_OPT_BH_ASYNC: return e->bh_async;
        case NBX_OPT_BH_WALK: return e->bh_walk;
        case NBX_OPT_BH_FUSE_KICK: return e->bh_fuse_kick;
        default: return NBX_ERR_INVALID;
    }
}

// What the engine has done so far (enum nbx_stat; rounds 1-4 had these among the options).  INT64_MIN for an unknown one:
// -1 is a legitimate value of NBX_STAT_DRAW_AMBIGUOUS ("the last draw ran on the host").
int64_t nbx_get_stat(const nbx_engine* e, int32_t stat)
{
    if (!e) return INT64_MIN;
    if (e->any_pending() && (stat == NBX_STAT_BH_FALLBACKS || stat == NBX_STAT_BH_LAST_TREE || stat == NBX_STAT_BH_REFUSAL ||
                             stat == NBX_STAT_BH_CLASS_SWITCHES || stat == NBX_STAT_BH_COLD_RESORTS || stat == NBX_STAT_BH_CHAIN_MERGED ||
                             stat == NBX_STAT_BH_CHAIN_APPROX)) {
        // what the last step ran on is known once its build's verdict is read; a redo that fails leaves the counters stale
        if (resolve_pending(const_cast<nbx_engine*>(e)) != NBX_OK) return INT64_MIN;
    }
    switch (stat) {
        case NBX_STAT_BH_CLASS_SWITCHES: return e->bh_class_switches;
        case NBX_STAT_BH_COLD_RESORTS: return e->bh_cold_resorts;
        case NBX_STAT_BH_CHAIN_MERGED: return e->bh_chain_merged;
        case NBX_STAT_BH_CHAIN_APPROX: return e->bh_chain_approx;
        case NBX_STAT_BH_FALLBACKS: return e->bh_fallbacks;
        case NBX_STAT_BH_LAST_TREE: return e->bh_last_tree_device;
        case NBX_STAT_BH_REFUSAL: return e->bh_last_refusal;
        case NBX_STAT_DRAW_AMBIGUOUS: return e->draw_ambiguous;
        default: return INT64_MIN;
    }
}

// nbx_get_option cannot tell the legitimate value -1 ("auto" of NBX_OPT_DRAW_DEVICE / NBX_OPT_BH_TREE / NBX_OPT_BH_FOLD) from
// NBX_ERR_INVALID: this form returns the status and the value separately.
int32_t nbx_query_option(const nbx_engine* e, int32_t option, int64_t* value)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    switch (option) {
        case NBX_OPT_FORCE_MODE: case NBX_OPT_JSPLIT: case NBX_OPT_BODIES_PER_THREAD: case NBX_OPT_DIM: case NBX_OPT_PROFILE:
        case NBX_OPT_KERNEL_VARIANT: case NBX_OPT_SOURCE_PRECISION: case NBX_OPT_DRAW_DEVICE: case NBX_OPT_BH_TREE: case NBX_OPT_BH_WAVE:
        case NBX_OPT_STRICT_KERNEL: case NBX_OPT_BH_FOLD: case NBX_OPT_BH_ASYNC: case NBX_OPT_BH_WALK: case NBX_OPT_BH_FUSE_KICK:
            if (value) *value = nbx_get_option(e, option);
            return NBX_OK;
        default:
            return fail(NBX_ERR_INVALID, "unknown option %d", option);
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
    try {   // nothing may unwind across the C ABI
        nbx::preset_random_disk(e->host, n, e->rng);
    } catch (const std::bad_alloc&) {
        return fail(NBX_ERR_ALLOC, "out of memory for %d particles", (int)n);
    }
    after_host_state_change(e);
    return NBX_OK;
}

int32_t nbx_stable_orbits(nbx_engine* e, int32_t n, float rmin, float rmax)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    ensure_seed(e);
    try {
        nbx::preset_stable_orbits(e->host, n, rmin, rmax, e->rng);
    } catch (const std::bad_alloc&) {
        return fail(NBX_ERR_ALLOC, "out of memory for %d particles", (int)n);
    }
    after_host_state_change(e);
    return NBX_OK;
}

int32_t nbx_plummer_sphere(nbx_engine* e, int32_t n, uint64_t seed, int32_t dim)
{
    if (!e || n < 0 || (dim != 2 && dim != 3)) return fail(NBX_ERR_INVALID, "bad engine, n or dim (2 | 3)");
    try {
        nbx::workload_plummer_sphere(e->host, n, seed, dim);
    } catch (const std::bad_alloc&) {
        return fail(NBX_ERR_ALLOC, "out of memory for %d particles", (int)n);
    }
    after_host_state_change(e);
    return NBX_OK;
}

int32_t nbx_two_galaxies(nbx_engine* e, int32_t n, uint64_t seed)
{
    if (!e || n < 0) return fail(NBX_ERR_INVALID, "bad engine or n");
    try {
        nbx::workload_two_galaxies(e->host, n, seed);
    } catch (const std::bad_alloc&) {
        return fail(NBX_ERR_ALLOC, "out of memory for %d particles", (int)n);
    }
    after_host_state_change(e);
    return NBX_OK;
}

int32_t nbx_num_particles(const nbx_engine* e) { return e ? e->n : NBX_ERR_INVALID; }

int32_t nbx_set_particles3(nbx_engine* e, int32_t n, const float* px, const float* py, const float* pz, const float* vx,
                           const float* vy, const float* vz, const float* m)
{
    if (!e || n < 0) return fail(NBX_ERR_INVALID, "bad engine or n");
    if (n > 0 && (!px || !py || !vx || !vy || !m)) return fail(NBX_ERR_INVALID, "null input array");
    try {
        e->host.resize(n);
    } catch (const std::bad_alloc&) {
        return fail(NBX_ERR_ALLOC, "out of memory for %d particles", (int)n);
    }
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
    // the reference spawns no worker and updates nobody for nthreads <= 0 (the division by it sits inside the closure of an
    // empty iterator, nbody.rs:424-428); level 2 reports the argument instead of silently doing nothing
    if (nthreads <= 0) return fail(NBX_ERR_INVALID, "nthreads must be >= 1 (the reference would update no particle)");
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
    const int prc = resolve_pending(e);
    if (prc != NBX_OK) return prc;
    HIP_TRY(wait_stream(e->stream, e->n <= kSpinMaxBodies));
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
        HIP_TRY(nbx::launch_reduce_forces(e->d_posm, e->lo, slab, e->d_acc, e->last.jsplit, stride, e->d_out4, e->stream,
                                          exceptions_of(e), self_image_of(e)));
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
        unsigned* guard = nullptr;
        if (nbx::strict_fastdiv_ok(e->mass_min, e->mass_max)) {
            rc = grow(&e->d_guard, &e->guard_cap, 1);
            if (rc != NBX_OK) return rc;
            guard = e->d_guard;
        }
        HIP_TRY(nbx::launch_force_strict(e->d_posm, e->n, e->lo, slab, e->d_f2, e->stream, &e->last, guard, e->strict_kernel));
    } else {
        bool on_device = false;
        if (e->use_device_tree()) {
            rc = build_tree_on_device(e, &on_device);
            if (rc != NBX_OK) return rc;
        }
        bool ordered = on_device;
        if (!on_device) {
            // host tree, big system: the Morton order of the bodies, as the stepping path takes it (engine_bh.cpp step_bh) -- in
            // particle order a force-only evaluation of a million bodies ran the per-lane walk: 4.4 ms and 6 GB of traffic against
            // 0.45 ms and 0.6 GB (round 5; what VERDICT r04 read as "12.8 x the algorithmic bytes" was this launch in the average)
            const bool want_order = e->bh_wave && e->world == 1 && e->n >= 65536;
            rc = build_and_upload_tree(e, nullptr, 0, want_order);
            if (rc != NBX_OK) return rc;
            ordered = want_order && e->d_perm != nullptr;
        }
        e->bh_last_tree_device = on_device ? 1 : 0;
        const unsigned* perm = (ordered && e->world == 1) ? e->d_perm : nullptr;
        if (e->force_mode == 0) {
            rc = launch_fast_walk(e, theta, perm, perm && e->bh_wave, on_device, nullptr, 0, 0, 0);
            if (rc != NBX_OK) return rc;
        } else {
            ProfScope ps(e, NBX_K_BH_EVAL);
            HIP_TRY(nbx::launch_bh_eval(e->d_posm, e->lo, slab, e->d_nodes, (int)e->n_flat, theta, e->force_mode, e->d_f2, e->stream, perm));
        }
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
    // list of the particles whose tail octant the device leaves to the host (draw.hip): a counter + up to n records
    const size_t amb_need = 16 + sizeof(nbx::DrawAmbiguous) * (size_t)std::max(e->n, 1);
    if (amb_need > e->amb_bytes) {
        if (e->d_amb) HIP_TRY(hipFree(e->d_amb));
        e->d_amb = nullptr;
        e->amb_bytes = 0;
        HIP_TRY(hipMalloc(&e->d_amb, amb_need + amb_need / 8));
        e->amb_bytes = amb_need + amb_need / 8;
    }
    unsigned* d_cnt = static_cast<unsigned*>(e->d_amb);
    nbx::DrawAmbiguous* d_rec = reinterpret_cast<nbx::DrawAmbiguous*>(static_cast<char*>(e->d_amb) + 16);
    HIP_TRY(nbx::launch_draw(e->d_posm, e->d_vel, e->n, w, h, x1, y1, scalex, scaley, e->d_counts, e->d_fb, d_cnt, d_rec, e->stream));
    // The caller's framebuffer is pageable (a mapped GL buffer in the reference's app): a copy straight into it goes through the
    // runtime's staging path.  Land the image -- and, right behind it, the count of ambiguous tails -- in pinned memory with one
    // DMA each, then hand it over with a host memcpy.  (Four pieces, each copied to the caller while the next is on the bus,
    // an event per piece: no gain -- 0.132 vs 0.130 ms per 512 x 512 draw; the event waits cost what the overlap saves.)
    if (px + 4 > e->h_fb_cap) {
        if (e->h_fb) HIP_TRY(hipHostFree(e->h_fb));
        e->h_fb = nullptr; e->h_fb_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_fb), (px + 4) * 4, hipHostMallocDefault));
        e->h_fb_cap = px + 4;
    }
    HIP_TRY(hipMemcpyAsync(e->h_fb, e->d_fb, px * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_fb + px, d_cnt, sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(wait_stream(e->stream, e->n <= kSpinMaxBodies));   // (a 512 x 512 frame is ~0.1 ms: poll before blocking, engine_internal.h)
    std::memcpy(fb, e->h_fb, px * 4);
    const unsigned n_amb = e->h_fb[px];
    e->draw_ambiguous = (int)n_amb;
    if (n_amb) {
        std::vector<nbx::DrawAmbiguous> rec(n_amb);
        HIP_TRY(hipMemcpy(rec.data(), d_rec, sizeof(nbx::DrawAmbiguous) * n_amb, hipMemcpyDeviceToHost));
        for (const nbx::DrawAmbiguous& r : rec) nbx::draw_add_tail(fb, w, h, r.xi, r.yi, r.vx, r.vy);
    }
    return NBX_OK;
}

int32_t nbx_draw(nbx_engine* e, int32_t w, int32_t h, uint32_t* fb)
{
    if (!e || !fb || w <= 0 || h <= 0) return fail(NBX_ERR_INVALID, "bad draw arguments");
    // -1 (default): on the device once the state lives there and has >= 4096 bodies (a frame would otherwise download
    // n x 32 B and splat on the host: 0.32 ms vs 0.10 ms at the reference's 10 000 bodies, profiles/r02_frame_loop_level1.txt)
    const bool on_device = e->draw_device == 1 || (e->draw_device < 0 && e->dev_valid && e->world == 1 && e->n >= 4096);
    if (on_device) {
        if (e->world != 1) return fail(NBX_ERR_STATE, "device draw needs the whole state on one GPU");
        return draw_on_device(e, w, h, fb);
    }
    int rc = download_positions(e);
    if (rc != NBX_OK) return rc;
    rc = download_velocities(e);
    if (rc != NBX_OK) return rc;
    nbx::draw_particles(e->host.px.data(), e->host.py.data(), e->host.vx.data(), e->host.vy.data(), e->n, w, h, fb);
    e->draw_ambiguous = -1;   // host draw
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
        rc0 = build_tree_on_device(e, &done, /*may_demote=*/false);   // (the class asked for, or nothing)
        if (rc0 != NBX_OK) return rc0;
        if (!done) return fail(NBX_ERR_STATE, "device tree build refused (NBX_STAT_BH_REFUSAL says why)");
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
    if (ok) {   // the header's count must agree with the file's size BEFORE anything is allocated from it
        long here = std::ftell(f);
        ok = here == 16 && std::fseek(f, 0, SEEK_END) == 0;
        const long size = ok ? std::ftell(f) : -1;
        ok = ok && size == 16 + 28L * (long)hdr[0] && std::fseek(f, 16, SEEK_SET) == 0;
    }
    nbx::HostState st;
    if (ok) {
        try {
            st.resize(hdr[0]);
        } catch (const std::bad_alloc&) {
            std::fclose(f);
            return fail(NBX_ERR_ALLOC, "out of memory for %d particles", (int)hdr[0]);
        }
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
    if (reinterpret_cast<uintptr_t>(device_ptr) & 15u) return fail(NBX_ERR_INVALID, "positions buffer must be 16-byte aligned (float4 records)");
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
    // the pointer is mutable and the caller may write positions through it (with world == 1 too): nothing derived from the old
    // positions -- the order-sorted copy the warm sort reads its keys from -- may be trusted any more
    e->positions_moved();
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
    e->positions_moved();   // (a caller with its own stream orders its own writes to the positions on it)
    return NBX_OK;
}

int32_t nbx_profile_reset(nbx_engine* e)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    if (e->dev_ready) {
        HIP_TRY(hipSetDevice(e->device));
        const int prc = resolve_pending(e);
        if (prc != NBX_OK) return prc;
        prof_fold(e, true);
    }
    for (int k = 0; k < NBX_K_COUNT; k++) { e->prof_ms[k] = 0.0; e->prof_n[k] = 0; }
    return NBX_OK;
}

int32_t nbx_profile_read(nbx_engine* e, int32_t kernel_id, double* total_ms, int32_t* launches)
{
    if (!e || kernel_id < 0 || kernel_id >= NBX_K_COUNT) return fail(NBX_ERR_INVALID, "bad kernel id");
    if (e->dev_ready) {
        HIP_TRY(hipSetDevice(e->device));
        const int prc = resolve_pending(e);
        if (prc != NBX_OK) return prc;
        prof_fold(e, true);
        if (!e->prof.empty()) return fail(NBX_ERR_HIP, "profiling events could not be read");
    }
    if (total_ms) *total_ms = e->prof_ms[kernel_id];
    if (launches) *launches = e->prof_n[kernel_id];
    return NBX_OK;
}

int32_t nbx_bh_work(nbx_engine* e, float theta, uint64_t* node_visits, uint64_t* pair_evals)
{
    uint64_t out[4] = {0, 0, 0, 0};
    const int32_t rc = nbx_bh_work_detail(e, theta, out);
    if (rc != NBX_OK) return rc;
    if (node_visits) *node_visits = out[0];
    if (pair_evals) *pair_evals = out[1];
    return NBX_OK;
}

int32_t nbx_bh_work_detail(nbx_engine* e, float theta, uint64_t* out4)
{
    if (!e || !out4) return fail(NBX_ERR_INVALID, "null argument");
    int rc = upload(e);
    if (rc != NBX_OK) return rc;
    bool on_device = false;
    if (e->use_device_tree()) {
        rc = build_tree_on_device(e, &on_device);
        if (rc != NBX_OK) return rc;
    }
    if (!on_device) {
        rc = build_and_upload_tree(e);
        if (rc != NBX_OK) return rc;
    }
    unsigned long long* d_tot = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_tot), 32));
    struct Free { unsigned long long* p; ~Free() { if (p) (void)hipFree(p); } } guard{d_tot};   // every return below releases it
    HIP_TRY(hipMemsetAsync(d_tot, 0, 32, e->stream));
    if (e->bh_walk != 0 && e->force_mode == 0 && nbx::bh_groups_addressable((int)e->n_flat)) {   // counted over the structure the selected walk uses
        rc = grow(&e->d_groups, &e->groups_cap, nbx::bh_groups_count((int)e->n_flat));
        if (rc != NBX_OK) return rc;
        HIP_TRY(nbx::launch_bh_groups(e->d_nodes, (int)e->n_flat, theta, e->d_groups, /*compact=*/on_device, e->stream));
        HIP_TRY(nbx::launch_bh_count_groups(e->d_posm, e->lo, e->slab(), e->d_groups, d_tot, e->stream));
    } else {
        HIP_TRY(nbx::launch_bh_count(e->d_posm, e->lo, e->slab(), e->d_nodes, (int)e->n_flat, theta, d_tot, e->stream));
    }
    unsigned long long h[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(h, d_tot, 32, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int i = 0; i < 4; i++) out4[i] = h[i];
    return NBX_OK;
}

int32_t nbx_bh_walk_trace(nbx_engine* e, float theta, int32_t cap_walks, uint64_t* out)
{
    if (!e || !out || cap_walks <= 0) return fail(NBX_ERR_INVALID, "bad arguments");
    if (e->force_mode != 0 || e->bh_walk == 0) return fail(NBX_ERR_STATE, "the trace is of the child-group walk (fast mode)");
    int rc = upload(e);
    if (rc != NBX_OK) return rc;
    rc = resolve_pending(e);
    if (rc != NBX_OK) return rc;
    const int slab = e->slab();
    const int walks = nbx::bh_walk_count(slab);
    if (walks > cap_walks) return walks;
    HIP_TRY(hipSetDevice(e->device));
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned long long) * 4 * (size_t)walks));
    hipError_t err = hipMemsetAsync(d, 0, sizeof(unsigned long long) * 4 * (size_t)walks, e->stream);
    e->d_walk_trace = d;
    e->walk_traced = false;
    std::vector<float> fx((size_t)slab), fy((size_t)slab);
    rc = err == hipSuccess ? nbx_forces(e, theta, slab, fx.data(), fy.data(), nullptr) : NBX_ERR_HIP;
    e->d_walk_trace = nullptr;
    const bool traced = e->walk_traced;
    if (rc >= 0 && traced) err = hipMemcpy(out, d, sizeof(unsigned long long) * 4 * (size_t)walks, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (rc < 0) return rc;
    HIP_TRY(err);
    // only the shared (one walk per wave) form writes a trace: a host tree below 65 536 bodies, NBX_OPT_BH_WAVE = 0 or a sharded
    // engine runs the per-lane form -- say so instead of returning rows of zeros
    if (!traced) return fail(NBX_ERR_STATE, "this evaluation ran the per-lane walk, which leaves no trace (host tree below 65 536 bodies, wave walk off, or world > 1)");
    return walks;
}

float nbx_bh_take_threshold(float s, float theta) { return nbx::bh_take_threshold(s, theta); }

int32_t nbx_bh_take_thresholds_device(nbx_engine* e, int32_t count, const float* s, const float* theta, float* out)
{
    if (!e || count < 0 || (count > 0 && (!s || !theta || !out))) return fail(NBX_ERR_INVALID, "bad arguments");
    if (count == 0) return NBX_OK;
    int rc = ensure_device(e);
    if (rc != NBX_OK) return rc;
    HIP_TRY(hipSetDevice(e->device));
    float* d = nullptr;
    const size_t bytes = sizeof(float) * (size_t)count;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), 3 * bytes));
    hipError_t err = hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, e->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(d + count, theta, bytes, hipMemcpyHostToDevice, e->stream);
    if (err == hipSuccess) err = nbx::launch_bh_thresholds(d, d + count, d + 2 * (size_t)count, count, e->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(out, d + 2 * (size_t)count, bytes, hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    (void)hipFree(d);
    HIP_TRY(err);
    return NBX_OK;
}

int32_t nbx_bh_host_timing(nbx_engine* e, double* ms4, int32_t* steps, int32_t* nodes)
{
    if (!e) return fail(NBX_ERR_INVALID, "null engine");
    if (e->dev_ready) {
        const int prc = resolve_pending(e);
        if (prc != NBX_OK) return prc;
    }
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


}  // extern "C"
