// level1.cpp -- the six reference symbols (rs-src/nbody.rs:34-35,:39-40,:73-74,:106-107,:186-187,:482-483)
// on a process-global engine (or multi-GPU group), serialised by a mutex like the reference's PARTICLES.
#include <mutex>

#include "engine_internal.h"

using namespace nbxi;

extern "C" {

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
    if (tree && std::strcmp(tree, "host") == 0) e->bh_tree_device = 0;
    const char* fold = std::getenv("NB_BH_FOLD");   // device tree interior nodes: reference (f32 running fold) | exact (sums rounded once)
    if (fold && std::strcmp(fold, "exact") == 0) e->bh_fold = 0;
    if (fold && std::strcmp(fold, "reference") == 0) e->bh_fold = 1;
    const char* bits = std::getenv("NB_SOURCE_BITS");   // 16: all-pairs sources from the half4 copy (BASELINE config #5)
    if (bits && std::atoi(bits) == 16) e->source_half = 1;
    const char* draw = std::getenv("NB_DRAW");
    if (draw && std::strcmp(draw, "device") == 0) e->draw_device = 1;
    if (draw && std::strcmp(draw, "host") == 0) e->draw_device = 0;
}

static nbx_engine* global_engine()   // engine 0 of the group when NB_GPUS > 1
{
    if (!g_engine) {
        const char* gpus = std::getenv("NB_GPUS");
        int want = gpus ? (std::strcmp(gpus, "all") == 0 ? nbx_device_count() : std::atoi(gpus)) : 1;
        if (want > 1) {
            // NBX_GROUP_EXCHANGE=copy (peer copies instead of RCCL) lets engines share devices: wrap the ordinals, so that the
            // group path can be exercised on a box with fewer GPUs than NB_GPUS (tests)
            const char* xc = std::getenv("NBX_GROUP_EXCHANGE");
            const int present = nbx_device_count();
            std::vector<int32_t> devs((size_t)want);
            for (int i = 0; i < want; i++) devs[(size_t)i] = (xc && std::strcmp(xc, "copy") == 0 && present > 0) ? i % present : i;
            if (nbx_group_create(&g_group, devs.data(), want) != NBX_OK) die("NB_GPUS group creation");
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
    g_group->fp32_stale = false;
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
    if (theta != 0.0f && nthreads <= 0) {
        // Reference: `(0..nthreads).map(..)` is empty -- no worker, no division by nthreads (it sits inside the closure,
        // nbody.rs:424-428), no particle touched -- but the quadtree has been built by then (:380-417), so its asserts
        // (depth > 50, mass <= 0, ...) still panic.  Same here: build the host tree for its checks, update nothing.
        // (group after fp16-source steps: engine 0's fp32 positions of the other slabs are re-gathered first)
        if (g_group && group_replicate_fp32(g_group) != NBX_OK) die("nb_step_barnes_hut");
        const int rc = nbx_bh_tree_dump(e, nullptr, 0);
        if (rc == NBX_ERR_TREE_DEPTH || rc == NBX_ERR_TREE) die("nb_step_barnes_hut");
        return;
    }
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
