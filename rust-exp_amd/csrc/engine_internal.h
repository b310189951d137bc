// engine_internal.h -- shared internals of libnbody_mi355x.so's host layer (NOT part of the ABI; the public
// interface is include/nbody_mi355x.h).  engine.cpp = state owner + step drivers, c_api.cpp = level-2 nbx_*,
// group.cpp = single-process multi-GPU group, level1.cpp = the six reference symbols.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only (ncclComm_t); RCCL itself is dlopen'ed by group.cpp

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nbody_mi355x.h"
#include "host_ops.h"
#include "kernels.h"

struct ProfRec {
    int kernel;
    hipEvent_t start, stop;
};

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
    // round 4: the fast walk's copy of the tree, one record of (x, y, m, T) x 4 + child words per opened node (bh_walk.hip);
    // rebuilt from d_nodes by every fast Barnes-Hut evaluation (T depends on the step's theta)
    nbx::BhGroup* d_groups = nullptr;
    size_t groups_cap = 0;
    // longest-first launch order of the walks: costs of the last walk, the order made from them, and what shape they belong to
    unsigned long long* d_walk_trace = nullptr;   // nbx_bh_walk_trace: set for the one evaluation it traces
    bool walk_traced = false;                     // ... and whether the walk that ran was the shared (wave) form, the one that writes the trace
    int bh_fuse_kick = 1;                // NBX_OPT_BH_FUSE_KICK: 1 (default) = the child-group walk applies the kick-drift itself, 0 = separate kernel
                                         // (measured, round 4: 0.449 vs 0.430 ms at 1 M bodies -- spatially adjacent walks no longer run side by side)
    int bh_walk = 1;                     // NBX_OPT_BH_WALK: 1 = child groups, hand-scheduled loop (default), 2 = child groups, compiled
                                         // loop, 0 = the node walk of rounds 1-3 (bh_eval.hip)
    unsigned* d_guard = nullptr;   // max|coord| word for the bit-exact kernels' short division
    size_t guard_cap = 0;
    void* d_tree_ws = nullptr;     // device tree build workspace (NBX_OPT_BH_TREE = 1)
    size_t tree_ws_bytes = 0;
    hipStream_t side_stream = nullptr;   // device tree build: the root's fold runs here beside the rest (NBX_OPT_BH_FOLD = 1)
    hipEvent_t ev_side_go = nullptr, ev_side_done = nullptr;
    int* h_counters = nullptr;     // pinned: per-level node counters of the device build
    const unsigned* d_perm = nullptr;   // spatial body order produced by the device build
    float4* d_sorted_pos = nullptr;     // the positions in the order d_perm names, as the last fused kick-drift left them (BhKick::sorted)
    size_t sorted_pos_cap = 0;
    bool sorted_pos_valid = false;      // ... and whether they still ARE the current positions in that order: set by the launch of a
                                        // fused walk + kick-drift, taken back by everything else that moves bodies on the device or
                                        // replaces the order (positions_moved()); never with a bound positions buffer or world > 1
    void positions_moved() { sorted_pos_valid = false; }
    const float4* sorted_positions() const { return (sorted_pos_valid && world == 1 && !posm_external && sort_warm_n == n) ? d_sorted_pos : nullptr; }
    int sort_warm_n = 0;                // > 0: the tree workspace holds the sorted order of this many bodies as of the last build /
                                        // spatial order of THIS state (one step old at most): the next sort starts from it (bh_build.hip,
                                        // round 5). 0 after an upload of host state, a refused build, a new workspace
    const unsigned* d_slab_perm = nullptr;  // d_perm restricted to this engine's slab (world > 1)
    void* d_slab_ws = nullptr;
    size_t slab_ws_bytes = 0;
    // NBX_OPT_BH_TREE: 0 host, 1 device, -1 (default) device in fast mode from 512 (exact sums) / 1 024 (reference fold) bodies on (measured crossover: a step with the
    // host build costs 0.141 / 0.208 / 0.303 / 0.509 ms at 512 / 1000 / 2000 / 4000 bodies, with the device build 0.124 / 0.135 /
    // 0.144 / 0.145 ms; below ~400 bodies the host build's few microseconds win: profiles/r02_bh_tree_crossover.txt)
    int bh_tree_device = -1;
    static constexpr int kDeviceTreeFrom = 512;
    // the device build has none of the reference's tree asserts: a non-positive or NaN mass (nbody.rs:304) always goes to the
    // host build, which reports NBX_ERR_TREE where the reference panics (mass_min is NaN when any mass is NaN / infinite)
    // The bit-exact mode builds on the host unless the caller asks for the device build (NBX_OPT_BH_TREE = 1) AND that build
    // carries the reference fold: its flattened tree is then the host tree bit for bit -- or the build refuses and the host
    // builds after all -- so the bit-exact walk over it gives the reference's results (a third of the host-tree step at 10 000
    // bodies).  Opt-in: the identity of the two trees rests on the replay's analysis and on fuzzing, not on construction.
    bool use_device_tree() const
    {
        if (!(mass_min > 0.0f)) return false;
        if (force_mode != 0) return bh_tree_device == 1 && effective_fold() == 1;
        // (by size: the exact-sum build pays from ~400 bodies on, the reference fold -- the root's chain -- from ~1 000:
        //  profiles/r02_bh_tree_crossover.txt, r03_bh_tree_crossover_reference_fold.txt)
        return bh_tree_device == 1 || (bh_tree_device < 0 && n >= (effective_fold() == 1 ? 2 * kDeviceTreeFrom : kDeviceTreeFrom));
    }
    // NBX_OPT_BH_FOLD: interior nodes of the DEVICE-built tree: 1 = the reference's f32 running fold in arrival order (the host
    // tree's records bit for bit), 0 = roundings of exact sums (round 2), -1 (default) = BY COST (round 6): the fast mode takes the
    // reference fold only while its build costs at most 1.5 x the exact-sum build's -- up to kFoldCostMax bodies (kernels.h: at no
    // size the device build serves, measured) -- the bit-exact mode, which only the reference fold can serve, up to kFoldFaithfulMax
    int bh_fold = -1;
    // >= 0 while a step (or force evaluation) runs on another class than the option names: a fast-mode step whose reference-fold
    // build refused is redone on the exact-sum DEVICE build (class 0), and so is the back-off run behind refusals in a row
    int fold_forced = -1;
    int effective_fold() const
    {
        if (fold_forced >= 0) return fold_forced;
        if (bh_fold >= 0) return bh_fold;
        return n <= (force_mode != 0 ? nbx::kFoldFaithfulMax : nbx::kFoldCostMax) ? 1 : 0;
    }
    int bh_wave = 1;               // wave-uniform traversal when a spatial body order is available
    // A Barnes-Hut step on the device tree is enqueued WITHOUT waiting for the build's verdict (node count, EPS clusters): walk
    // and kick-drift are gated on the device by the build's own counters.  The host reads the verdict at the next call that
    // needs the state (resolve_pending) and, in the rare case the build had to refuse, redoes the step on the host tree --
    // the gated kernels left the state untouched.  NBX_OPT_BH_ASYNC = 0 waits inside the step as rounds 1-2 did.
    struct PendingStep {
        bool active = false;
        float theta = 0.f, dt = 0.f;
        int node_cap = 0, fold = 0;
    } pending[2];                   // up to two steps in flight: the next one is enqueued before the previous verdict is read
    int pend_next = 0;              // slot the next step takes (= the OLDER one when both are active)
    int* h_verdict[2] = {nullptr, nullptr};       // pinned: each slot's build counters
    hipEvent_t ev_step[2] = {nullptr, nullptr};   // each slot's step is complete on the stream
    bool any_pending() const { return pending[0].active || pending[1].active; }
    int bh_async = 1;
    int bh_last_tree_device = 0;   // where the last evaluated tree was built
    int bh_fallbacks = 0;          // steps / evaluations the device tree was selected for but the host tree served
    int bh_class_switches = 0;     // NBX_STAT_BH_CLASS_SWITCHES: fast-mode steps / evaluations the reference-fold device build was selected
                                   // for but the exact-sum DEVICE build served (a refusal redone there, or its back-off run)
    // Consecutive refused device builds (a dense core keeps its EPS clusters for many steps): from the second refusal in a row
    // the next 2, 4, .. kBackoffMaxSteps steps go straight to the class below -- and then ONE step tries the refused class again.
    // Two ladders (round 6): [1] the reference-fold build's refusals, whose class below is the exact-sum DEVICE build in the
    // fast mode (counted in bh_class_switches) and the host build in the bit-exact mode; [0] the exact-sum build's, whose class
    // below is the host build (counted in bh_fallbacks like the refusals).  Any replaced state (after_host_state_change) resets
    // both, an accepted build its own.
    int bh_refusal_streak[2] = {0, 0}, bh_demoted_steps_left[2] = {0, 0};
    int bh_last_refusal = 0;       // NBX_STAT_BH_REFUSAL: reasons of the last refused device build (status and counter word 5)
    void note_why(int status, int why)
    {
        bh_last_refusal = status == 1 ? 0x10000 : (why ? why : 0x20000);
        if (status != 1 && (why & nbx::kWhySortOverflow)) {   // the warm sort overflowed: the next builds sort from scratch (below)
            warm_holdoff = warm_holdoff_next;
            warm_holdoff_next = std::min(32, 2 * warm_holdoff_next);
        }
    }
    // A warm sort whose bucket overflows (more than 4 096 bodies on one 62-bit key: coincident positions; or, at 2e-8 per bucket and
    // step, bad luck) says nothing against the TREE: that build is redone at once from a cold sort, same class, on the device
    // (NBX_STAT_BH_COLD_RESORTS; round 6 -- rounds 5 sent the step to the host build), and the next 2, 4 .. 32 builds sort cold
    // too (a clump of coincident bodies stays one for many steps and would overflow every warm sort).
    int bh_cold_resorts = 0;
    int bh_chain_merged = 0, bh_chain_approx = 0;   // NBX_STAT_BH_CHAIN_*: the chain replay's tallies of the last accepted exact-sum build
    void note_chains(const int* c) { bh_chain_approx = c[6]; bh_chain_merged = c[7]; }
    int warm_holdoff = 0, warm_holdoff_next = 2;
    bool sort_overflowed() const { return bh_last_refusal != 0x10000 && (bh_last_refusal & nbx::kWhySortOverflow) != 0; }
    void note_refusal(int fold, int max_steps)
    {
        const int c = fold == 1 ? 1 : 0;
        if (++bh_refusal_streak[c] >= 2 && max_steps > 0) {
            const int k = bh_refusal_streak[c] - 1;
            bh_demoted_steps_left[c] = k >= 30 || (1 << k) > max_steps ? max_steps : (1 << k);
        }
    }
    // (an accepted reference-fold build ends both streaks: that class is the stricter one)
    void note_accepted(int fold) { bh_refusal_streak[0] = 0; if (fold == 1) bh_refusal_streak[1] = 0; }
    void reset_backoff() { bh_refusal_streak[0] = bh_refusal_streak[1] = bh_demoted_steps_left[0] = bh_demoted_steps_left[1] = 0; warm_holdoff = 0; warm_holdoff_next = 2; }
    // a fast-mode build of the reference-fold class that refuses is redone one class down on the device, not on the host
    bool demotes_on_device(int fold) const { return fold == 1 && force_mode == 0; }
    void* d_counts = nullptr;      // device draw: uint2 hit counters per pixel
    size_t counts_cap = 0;         // pixels
    unsigned* d_fb = nullptr;
    size_t fb_cap = 0;
    unsigned* h_fb = nullptr;      // pinned landing buffer of the device draw (the caller's framebuffer is pageable)
    size_t h_fb_cap = 0;
    int draw_device = -1;          // nb_draw: 0 host, 1 device, -1 device for >= 4096 bodies resident on one GPU
    void* d_amb = nullptr;         // device draw: counter + list of particles whose tail octant the host decides
    size_t amb_bytes = 0;
    int draw_ambiguous = 0;        // how many the last device draw handed over
    void* d_posh = nullptr;        // half4 (x,y,z,m) source copy (NBX_OPT_SOURCE_PRECISION = 16)
    bool posh_external = false;
    size_t posh_cap = 0;           // records
    int source_half = 0;
    float4* d_src4 = nullptr;      // the fp16 source copy widened to float4, rewritten before every wave-split sweep (K4)
    size_t src4_cap = 0;

    // options
    int force_mode = 0, jsplit = 0, bpt = 0, dim_opt = 0, profile = 0, variant = -1, strict_kernel = 0;
    bool any_z = false;
    float mass_min = 0.0f, mass_max = 0.0f;   // over the current bodies (masses never change during a run)
    // "one common mass + a handful of exceptions" (the reference's own nb_stable_orbits: unit planets + a 1000-mass sun,
    // nbody.rs:85-102): the unit-mass sweep (variant 7) runs with every source weightless at mass_common and K2 adds the
    // exceptional sources with weight m_j - mass_common.  mass_common = 0: no usable common mass (variant 6 runs).
    float mass_common = 0.0f;
    std::vector<int> exc_idx;        // bodies whose mass differs from mass_common (at most exc_cap(n))
    int* d_exc_idx = nullptr;
    float4* d_exc_rec = nullptr;     // (x, y, z, m_j - mass_common) snapshot written by the sweep kernel for K2
    size_t exc_cap_dev = 0;
    static int exc_cap(int n) { return std::min(n / 64, std::max(32, n / 1024)); }   // 1 024 bodies: 16, 10 000: 32, 262 144: 256
    bool unit_sweep_ok() const { return n > 0 && mass_common > 0.0f; }

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
    float* h_xy = nullptr;            // pinned, device-visible: planar x[cap], y[cap] for the host tree build
    size_t h_xy_cap = 0;
    hipEvent_t ev_xy = nullptr;       // the planar (x, y) download has landed
    void* d_route_ws = nullptr;       // device workspace of the routing + scatter help for the host build
    size_t route_ws_bytes = 0;
    char* h_route = nullptr;          // pinned, device-visible: insert events | bucket per body | bucket offsets
    size_t h_route_bytes = 0;

    std::vector<ProfRec> prof;              // event pairs still in flight (profiling on)
    std::vector<hipEvent_t> ev_free;        // recycled events: a long profiled run creates no new ones
    double prof_ms[NBX_K_COUNT] = {};       // folded totals per kernel id since the last reset
    int prof_n[NBX_K_COUNT] = {};
    nbx::ForceLaunch last{0, 0, 0, 0, 0, 0};
    std::chrono::steady_clock::time_point tree_t0;   // start of the device tree build in flight
    double host_ms[4] = {0, 0, 0, 0};  // Barnes-Hut host phases: download, build, flatten, upload (cumulative)
    int host_steps = 0;

    int slab() const { return hi - lo; }
};

// a class of device tree build forced for the duration of a scope (fold < 0: nothing forced)
struct FoldForce {
    nbx_engine* e;
    int saved;
    FoldForce(nbx_engine* eng, int fold) : e(eng), saved(eng->fold_forced) { if (fold >= 0) e->fold_forced = fold; }
    void set(int fold) { e->fold_forced = fold; }   // (-1: the options decide again)
    ~FoldForce() { e->fold_forced = saved; }
    FoldForce(const FoldForce&) = delete;
    FoldForce& operator=(const FoldForce&) = delete;
};

struct GroupWorkers;   // group.cpp: one persistent enqueue thread per engine (optional)

struct nbx_group {
    std::vector<nbx_engine*> eng;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    int exchanges = 0;
    bool fp32_stale = false;                // only the fp16 source copy was exchanged: fp32 positions of other slabs are old
    bool copy_exchange = false;             // peer copies + events instead of RCCL (requested, or after RCCL failed)
    int exchange_kind = 0;                  // NBX_GROUP_INFO_EXCHANGE: 0 RCCL, 1 peer copies requested, 2 peer copies after an RCCL failure
    int rccl_ranks = 0;                     // ranks ncclCommInitAll was given (0: no communicator exists)
    int rccl_fail_hook = 0;                 // NBX_GROUP_RCCL_FAIL=init|gather (tests): 1 = communicator creation, 2 = first all-gather "fails"
    std::string exchange_note;              // why the group fell back to peer copies
    std::vector<hipEvent_t> ev_ready;       // per engine: its slab is updated
    std::vector<hipEvent_t> ev_copied;      // per engine: it has pulled every other slab
    GroupWorkers* workers = nullptr;        // NBX_GROUP_ENQUEUE=threads / nbx_group_set_enqueue_threads
};

namespace nbxi {

using nbx::kTile;

extern thread_local std::string g_last_error;
int fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return nbxi::fail(NBX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

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

// Host waits of the stepping path.  A 10 000-body step is ~0.1 ms of GPU work; the runtime's blocking wait adds its wake-up
// latency to every such step, so SHORT waits poll first (NBX_SPIN_US microseconds, default 400; 0 = block at once) and block
// after that.  Only short ones: a system above kSpinMaxBodies bodies steps in several tenths of a millisecond and more -- polling
// would burn a core per waiting thread (a group host has one per GPU) for nothing -- and blocks at once.
constexpr int kSpinMaxBodies = 32768;
inline int spin_budget_us()
{
    static const int us = [] {
        const char* s = std::getenv("NBX_SPIN_US");
        return s ? std::max(0, std::atoi(s)) : 400;
    }();
    return us;
}
// hipSuccess: done; hipErrorNotReady: the budget ran out (block now); anything else: the query's own error, handed on as it is
template <typename Query>
inline hipError_t spin_until_done(Query query, bool short_work)
{
    const int budget = short_work ? spin_budget_us() : 0;
    if (budget <= 0) return hipErrorNotReady;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int k = 0; k < 16; k++) {
            const hipError_t st = query();
            if (st != hipErrorNotReady) return st;
        }
        (void)hipGetLastError();   // (hipErrorNotReady is sticky in hipGetLastError: it is not an error here)
        if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= budget) return hipErrorNotReady;
    }
}
inline hipError_t wait_stream(hipStream_t s, bool short_work)
{
    const hipError_t st = spin_until_done([&] { return hipStreamQuery(s); }, short_work);
    if (st != hipErrorNotReady) return st;
    (void)hipGetLastError();
    return hipStreamSynchronize(s);
}
inline hipError_t wait_event(hipEvent_t ev, bool short_work)
{
    const hipError_t st = spin_until_done([&] { return hipEventQuery(ev); }, short_work);
    if (st != hipErrorNotReady) return st;
    (void)hipGetLastError();
    return hipEventSynchronize(ev);
}

// Finished event pairs -> per-kernel totals; their events go back to the free list.  wait = false folds only the pairs
// that have already completed (no synchronisation inside a step).
inline void prof_fold(nbx_engine* e, bool wait)
{
    size_t k = 0;
    for (; k < e->prof.size(); k++) {
        ProfRec& r = e->prof[k];
        if (wait) {
            if (hipEventSynchronize(r.stop) != hipSuccess) break;
        } else if (hipEventQuery(r.stop) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess && r.kernel >= 0 && r.kernel < NBX_K_COUNT) {
            e->prof_ms[r.kernel] += ms;
            e->prof_n[r.kernel]++;
        }
        e->ev_free.push_back(r.start);
        e->ev_free.push_back(r.stop);
    }
    e->prof.erase(e->prof.begin(), e->prof.begin() + (long)k);
}

inline bool prof_event(nbx_engine* e, hipEvent_t* ev)
{
    if (!e->ev_free.empty()) {
        *ev = e->ev_free.back();
        e->ev_free.pop_back();
        return true;
    }
    return hipEventCreate(ev) == hipSuccess;
}

// HIP event pair on the engine's stream around a launch (NBX_OPT_PROFILE). The caller has made e->device current.
struct ProfScope {
    nbx_engine* e;
    hipEvent_t stop = nullptr;
    ProfScope(nbx_engine* eng, int kernel) : e(eng)
    {
        if (!e->profile) return;
        if (e->prof.size() >= 256) prof_fold(e, false);
        ProfRec r{kernel, nullptr, nullptr};
        if (!prof_event(e, &r.start)) return;
        if (!prof_event(e, &r.stop)) { e->ev_free.push_back(r.start); return; }
        (void)hipEventRecord(r.start, e->stream);
        e->prof.push_back(r);
        stop = r.stop;
    }
    ~ProfScope()
    {
        if (stop) (void)hipEventRecord(stop, e->stream);
    }
};

void compute_slab(nbx_engine* e);
int ensure_device(nbx_engine* e);
int refresh_half_sources(nbx_engine* e, int first, int count);
int upload(nbx_engine* e);
int download_positions(nbx_engine* e);
int download_velocities(nbx_engine* e);
void choose_launch(const nbx_engine* e, int n_targets, int tiles_total, int* variant, int* bpt, int* jsplit, int* dim);
int launch_forces_fast(nbx_engine* e);
bool log_enabled();   // NBX_LOG=1
nbx::MassExceptions exceptions_of(const nbx_engine* e);
nbx::SelfImage self_image_of(const nbx_engine* e);
int step_brute(nbx_engine* e, float dt);
int build_and_upload_tree(nbx_engine* e, nbx_engine* const* also = nullptr, int n_also = 0, bool order_bodies = false);
int build_tree_on_device_begin(nbx_engine* e, int* host_counters = nullptr, bool publish_by_kernel = false);
int build_tree_on_device_end(nbx_engine* e, bool* done);
int build_tree_on_device(nbx_engine* e, bool* done, bool may_demote = true);
int bh_eval_and_integrate(nbx_engine* e, float theta, float dt, bool on_device, bool have_perm, bool gated = false,
                          int* gate_host_out = nullptr);
// the fast traversal of e->d_nodes for this engine's slab into e->d_f2 (accelerations): child-group walk or node walk (NBX_OPT_BH_WALK)
int launch_fast_walk(nbx_engine* e, float theta, const unsigned* perm, bool wave, bool on_device, int* gate, int gate_node_cap,
                     int gate_crowd_limit, int gate_queue_limit, const nbx::BhKick* kick = nullptr);
bool walk_takes_kick(const nbx_engine* e, const unsigned* perm, bool wave, int nodes_or_cap);
int step_bh_group(nbx_engine* const* eng, int count, float theta, float dt);
int spatial_order(nbx_engine* e);
int slab_order(nbx_engine* e);
int step_bh(nbx_engine* e, float theta, float dt);
int resolve_pending(nbx_engine* e);   // the verdicts of the speculatively enqueued Barnes-Hut steps, oldest first (no-op when none)
void free_device(nbx_engine* e);
uint64_t entropy_seed();
void after_host_state_change(nbx_engine* e);
int group_replicate_fp32(nbx_group* g);   // group.cpp: fp32 positions of every slab current on every engine

}  // namespace nbxi
