// engine_bh.cpp -- the Barnes-Hut step drivers of libnbody_mi355x.so (nb_step_barnes_hut, nbody.rs:186-480): host tree build and
// upload, device tree build (two halves), traversal + kick-drift, the steps that are enqueued without waiting for the build's
// verdict, the single-process group's step.  State owner and the all-pairs drivers: engine.cpp.
#include <chrono>
#include <cmath>

#include "engine_internal.h"

namespace nbxi {


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
    // (bh_front.hip) while the host threads fold; any device-side problem just leaves both to the host
    nbx::QuadTree::RouteFn route = [e](const nbx::QuadTree::TopView& v, int warm, int rest, int* pbucket,
                                       nbx::QuadTree::Event* sorted, size_t* offset) -> bool {
        struct TopRec { float x1, y1, x2, y2; int32_t first_child, bucket; };
        static_assert(sizeof(TopRec) == 24 && sizeof(nbx::QuadTree::Event) == 16, "layout shared with bh_front.hip");
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
static constexpr int kBackoffMaxSteps = 32;   // see nbx_engine::note_refusal (both ladders)

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
    const bool warm = e->sort_warm_n == e->n && e->warm_holdoff == 0;   // (hold-off: cold sorts behind an overflowed warm one, engine_internal.h)
    if (e->warm_holdoff > 0) e->warm_holdoff--;
    HIP_TRY(nbx::device_tree_build_begin(e->d_posm, e->n, e->d_tree_ws, e->tree_ws_bytes, node_cap, e->d_nodes,
                                         publish_by_kernel ? nullptr : (host_counters ? host_counters : e->h_counters), &e->d_perm,
                                         e->stream, fold, e->side_stream, e->ev_side_go, e->ev_side_done,
                                         /*depth_panic_guard=*/e->force_mode != 0, warm, warm ? e->sorted_positions() : nullptr));
    e->positions_moved();    // the workspace holds a NEW order now: the order-sorted copy of the positions belongs to the old one
                             // (the fused kick-drift of the step this build serves, if there is one, leaves a current copy again)
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
    const int fold = e->effective_fold();
    int n_nodes = 0, status = 0;
    HIP_TRY(nbx::device_tree_build_end(e->n, node_cap, e->h_counters, &n_nodes, &status, e->stream, fold));
    if (status != 0) {
        // (who serves the step instead -- the exact-sum device build or the host build -- and which counter that is, is the caller's)
        e->note_refusal(fold, backoff_max_steps());
        e->note_why(status, e->h_counters[5]);
        e->d_perm = nullptr;
        e->sort_warm_n = 0;
        if (std::getenv("NBX_LOG"))
            std::fprintf(stderr, "[nbx] device tree build of %d bodies (%s) refused: status %d (1 = pool / queue overflow, 2 = EPS "
                                 "clusters), nodes %d of %d, left-behind bodies %d (why 0x%x), queued folds %d\n", e->n, fold == 1 ? "reference fold" : "exact sums",
                         status, e->h_counters[0], node_cap, e->h_counters[1], (unsigned)e->h_counters[5], e->h_counters[2]);
        return NBX_OK;   // caller takes the class below
    }
    e->note_accepted(fold);
    e->note_chains(e->h_counters);
    if (fold == 0 && std::getenv("NBX_LOG_CHAINS"))   // (waited-for builds only: the pipelined steps hand over fewer words)
        std::fprintf(stderr, "[nbx] chain replay, %d bodies: %d segments listed, %d bodies merged, %d approximate\n", e->n, e->h_counters[4], e->h_counters[7], e->h_counters[6]);
    e->n_flat = (size_t)n_nodes;
    e->host_ms[1] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - e->tree_t0).count();
    e->host_steps++;
    *done = true;
    return NBX_OK;
}

// The device build of the class the engine is set to, waited for.  *done = false: the caller builds on the host (counted here).
// may_demote (round 6): a FAST-mode build of the reference-fold class that refuses is tried once more as an exact-sum build --
// the class the fast mode defaults to anyway, 8 x faster than the host build at 65 536 bodies and inside the fast mode's stated
// tolerance -- before the host is asked; the bit-exact mode, and a refusal of the exact-sum class itself, go to the host build.
int build_tree_on_device(nbx_engine* e, bool* done, bool may_demote)
{
    *done = false;
    if (e->n > kDeviceTreeMaxBodies) return NBX_OK;   // caller takes the host path
    int rc = build_tree_on_device_begin(e);
    if (rc != NBX_OK) return rc;
    rc = build_tree_on_device_end(e, done);
    if (rc != NBX_OK || *done) return rc;
    if (e->sort_overflowed()) {   // the warm sort's order failed, not the tree: once more from a cold sort (the refusal took the order back)
        e->bh_cold_resorts++;
        rc = build_tree_on_device_begin(e);
        if (rc != NBX_OK) return rc;
        rc = build_tree_on_device_end(e, done);
        if (rc != NBX_OK || *done) return rc;
    }
    if (may_demote && e->demotes_on_device(e->effective_fold())) {
        e->bh_class_switches++;
        const int first = e->bh_last_refusal;
        FoldForce down(e, 0);
        rc = build_tree_on_device_begin(e);
        if (rc != NBX_OK) return rc;
        rc = build_tree_on_device_end(e, done);
        if (rc != NBX_OK || *done) return rc;
        e->bh_last_refusal |= first;   // both classes refused: the reasons of the class asked for stay on record
    }
    e->bh_fallbacks++;
    return NBX_OK;
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
    // ALWAYS the library sort here: the warm sort may overflow a bucket (more than kBucketCap bodies on one 62-bit key --
    // coincident positions, exactly what sends a step to the host tree -- or, at 2e-8 per bucket and step, a reshuffled system) and then
    // leaves a non-permutation; the device build's gate catches that, this path has no gate.  The sort overlaps the host build
    // (milliseconds), so nothing is lost; its order is still a good sampling frame for the next warm DEVICE build.
    HIP_TRY(nbx::device_spatial_order(e->d_posm, e->n, e->d_tree_ws, e->tree_ws_bytes, &e->d_perm, e->stream, /*warm=*/false, nullptr));
    e->positions_moved();    // (as in build_tree_on_device_begin: a new order)
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
        if (e->d_walk_trace && wave && perm) e->walk_traced = true;
        HIP_TRY(nbx::launch_bh_walk_groups(e->d_posm, e->lo, slab, e->d_groups, e->d_f2, e->stream, perm, wave, e->bh_walk == 1, gate,
                                           gate_node_cap, gate_crowd_limit, gate_queue_limit, e->d_walk_trace, kick));
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
    e->positions_moved();   // this step moves the slab's bodies; only a fused kick-drift below leaves the order-sorted copy current again
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
        kicked = walk_takes_kick(e, perm, wave, gate ? node_cap : (int)e->n_flat);
        // the new positions once more in the walks' order, for the next build's sort: one GPU, engine-owned positions, the
        // whole system in one slab, the order that of the tree workspace (d_perm)
        const bool leave_sorted = kicked && e->world == 1 && !e->posm_external && slab == e->n && perm == e->d_perm && e->sort_warm_n == e->n;
        if (leave_sorted) {
            rc = grow(&e->d_sorted_pos, &e->sorted_pos_cap, (size_t)e->n);
            if (rc != NBX_OK) return rc;
        }
        const nbx::BhKick kick{e->d_vel, e->d_posm, dt, gated ? gate_host_out : nullptr, leave_sorted ? e->d_sorted_pos : nullptr};
        rc = launch_fast_walk(e, theta, perm, wave, on_device, gate, node_cap, crowd_limit, queue_limit, kicked ? &kick : nullptr);
        if (rc != NBX_OK) return rc;
        if (leave_sorted) e->sorted_pos_valid = true;
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
    HIP_TRY(wait_event(e->ev_step[slot], e->n <= kSpinMaxBodies));
    const nbx_engine::PendingStep p = e->pending[slot];
    e->pending[slot].active = false;
    const int status = verdict_of(e, slot);
    if (status == 0) {
        e->note_accepted(p.fold);
        e->note_chains(e->h_verdict[slot]);
        e->n_flat = (size_t)e->h_verdict[slot][0];
        e->bh_last_tree_device = 1;
        e->host_steps++;
        return NBX_OK;
    }
    // refused: this step's gated kernels did nothing and poisoned the step behind it (if one is in flight)
    e->note_refusal(p.fold, backoff_max_steps());
    e->note_why(status, e->h_verdict[slot][5]);
    e->d_perm = nullptr;
    e->sort_warm_n = 0;
    const int other = slot ^ 1;
    const bool redo_later = e->pending[other].active;
    const nbx_engine::PendingStep later = e->pending[other];
    e->pending[other].active = false;
    HIP_TRY(hipStreamSynchronize(e->stream));
    const bool on_device_first = e->demotes_on_device(p.fold);   // fast mode, reference fold refused: the exact-sum DEVICE build first
    if (std::getenv("NBX_LOG"))
        std::fprintf(stderr, "[nbx] device tree build of %d bodies (%s) refused (status %d: nodes %d of %d, left-behind bodies %d (why 0x%x), queued folds %d): "
                             "step redone on the %s%s\n", e->n, p.fold == 1 ? "reference fold" : "exact sums", status, e->h_verdict[slot][0], p.node_cap,
                     e->h_verdict[slot][1], (unsigned)e->h_verdict[slot][5], e->h_verdict[slot][2],
                     e->sort_overflowed() ? "device tree, cold sort" : on_device_first ? "exact-sum device tree" : "host tree", redo_later ? ", the step behind it enqueued again" : "");
    HIP_TRY(hipMemsetAsync(nbx::device_tree_counters(e->d_tree_ws) + nbx::kTreePoisonWord, 0, sizeof(int), e->stream));
    int rc = NBX_OK;
    bool on_device = false;
    if (e->sort_overflowed()) {   // only the warm sort's order failed: the same class once more, from a cold sort, on the device
        e->bh_cold_resorts++;
        FoldForce same(e, p.fold);
        rc = build_tree_on_device(e, &on_device);   // (demotes / counts the fallback itself if the tree is refused as well)
        if (rc != NBX_OK) return rc;
    } else if (on_device_first) {
        e->bh_class_switches++;
        const int first = e->bh_last_refusal;
        FoldForce down(e, 0);
        rc = build_tree_on_device(e, &on_device, /*may_demote=*/false);   // (counts the fallback itself if this class refuses too)
        if (rc != NBX_OK) return rc;
        if (!on_device) e->bh_last_refusal |= first;
    } else {
        e->bh_fallbacks++;
    }
    bool have_perm = on_device;
    if (!on_device) {
        const bool want_order = e->bh_wave && e->n >= 65536;
        rc = build_and_upload_tree(e, nullptr, 0, want_order);
        if (rc != NBX_OK) return rc;
        have_perm = want_order && e->d_perm != nullptr;
    }
    rc = bh_eval_and_integrate(e, p.theta, p.dt, on_device, have_perm);
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
    if (e->any_pending() && (e->bh_refusal_streak[0] > 0 || e->bh_refusal_streak[1] > 0)) {   // a verdict that may start a back-off: read it before choosing the path
        rc = resolve_pending(e);
        if (rc != NBX_OK) return rc;
    }
    // the tree class of this step: the one the options name, or -- while a back-off run lasts (engine_internal.h) -- the class below
    // (a step re-enqueued from inside another step's redo chooses for itself: the outer step's class is not this one's)
    FoldForce this_step(e, -1);
    this_step.set(-1);
    bool device_tree = e->use_device_tree() && e->n <= kDeviceTreeMaxBodies;
    int forced = -1;
    if (device_tree) {
        int fold = e->effective_fold();
        if (fold == 1 && e->bh_demoted_steps_left[1] > 0) {
            e->bh_demoted_steps_left[1]--;
            if (e->force_mode == 0) { forced = fold = 0; e->bh_class_switches++; }   // fast mode: the exact-sum DEVICE build serves the run
            else { device_tree = false; e->bh_fallbacks++; }                        // bit-exact mode: only the host build can
        }
        if (device_tree && fold == 0 && e->bh_demoted_steps_left[0] > 0) {
            e->bh_demoted_steps_left[0]--;
            e->bh_fallbacks++;
            device_tree = false;
        }
    }
    this_step.set(forced);
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
            if (!done) { on_device = false; eng[d]->bh_fallbacks++; }   // same bodies, same tree: if one build refuses, all do (the group goes to the host build)
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

}  // namespace nbxi
