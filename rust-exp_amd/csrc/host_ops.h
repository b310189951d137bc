// host_ops.h -- host-side parts of the nb_* surface that stay on the CPU by design:
// presets (nbody.rs:39-104), framebuffer splat (nbody.rs:482-617) and the Barnes-Hut quadtree
// build (nbody.rs:203-331, :388-415; north_star: "octree build stays on host").
// Internal header; the public ABI is include/nbody_mi355x.h.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

#include "kernels.h"

namespace nbx {

// Persistent host worker threads (created on first use, one set per process).  Spawning 31 threads costs ~0.8 ms on
// the target hosts and a Barnes-Hut step with a host tree has eight parallel regions: with fresh threads a third of
// the step was thread creation.  Tasks of a group may run on any worker or on the thread that waits for the group.
class TaskGroup {
public:
    TaskGroup() = default;
    ~TaskGroup() { wait(); }
    TaskGroup(const TaskGroup&) = delete;
    TaskGroup& operator=(const TaskGroup&) = delete;
    void run(std::function<void()> fn);   // enqueue one task
    void wait();                          // returns when every task of this group has finished (helps running tasks)
private:
    friend class WorkerPool;
    int pending_ = 0;                     // guarded by the pool's mutex
};
// fn(t) for t = 0 .. count-1, t = 0 on the calling thread; returns when all are done
void parallel_for(int count, const std::function<void(int)>& fn);
int host_threads();                       // worker threads a parallel host phase may use (NBX_HOST_THREADS overrides)

// Host mirror of the particle state, SoA. z/vz are zero for everything that comes through the
// reference's 2-D surface.
struct HostState {
    std::vector<float> px, py, pz, vx, vy, vz, m;
    int n() const { return (int)px.size(); }
    void resize(int n);
    void clear() { resize(0); }
};

// splitmix64 -> top 24 bits -> [0,1) f32 (rand 0.3 `next_f32` construction)
struct Rng {
    uint64_t s;
    float next_f32();
    float range(float lo, float hi) { return lo + (hi - lo) * next_f32(); }
};

void preset_random_disk(HostState& st, int n, Rng& rng);                        // nbody.rs:39-71
void preset_stable_orbits(HostState& st, int n, float rmin, float rmax, Rng& rng);  // nbody.rs:73-104

// benchmark workloads of BASELINE.json's configs; stateless in the seed
void workload_plummer_sphere(HostState& st, int n, uint64_t seed, int dim);
void workload_two_galaxies(HostState& st, int n, uint64_t seed);

// nbody.rs:482-617; fb is w*h ABGR words, cleared here
void draw_particles(const float* px, const float* py, const float* vx, const float* vy, int n, int32_t w,
                    int32_t h, uint32_t* fb);
// One particle's TAIL (nbody.rs:541-565) added to an already drawn framebuffer: octant from the reference's f32 expression
// with this host's atan2f, bounds check, per-channel saturating add; the five centre-cross pixels stay magenta (the
// reference writes them last). Used by the device draw for the few particles whose octant it leaves to the host.
void draw_add_tail(uint32_t* fb, int32_t w, int32_t h, int32_t xi, int32_t yi, float vx, float vy);


// Quadtree with index-linked nodes. Node k's children (if any) are first_child[k] .. +3 in the
// reference's order [UL, UR, LL, LR] (nbody.rs:295-300).
struct QuadTree {
    struct Node {
        float x1, y1, x2, y2;  // AABB
        float px, py, m;       // COM + total mass (interior) / particle (exterior)
        int32_t first_child;   // -1 = exterior
    };
    struct Event { float x, y, m; unsigned depth; };   // a pending Node::insert(px, py, m, depth)
    std::vector<Node> nodes;   // nodes[0] = root. Whole tree (forest == false) or its top levels.
    // Threaded build leaves a forest: the subtree under top node root_of[b] lives in pools[b] with
    // pool-local child indices; pools[b][0] is that bucket root and supersedes nodes[root_of[b]].
    bool forest = false;
    int n_buckets = 0;
    std::vector<int> bucket_of;                 // top node -> bucket id or -1
    std::vector<int> root_of;                   // bucket id -> top node
    std::vector<std::vector<Node>> pools;       // capacity reused from step to step
    std::vector<size_t> pool_live;              // per pool: nodes that survive flattening (non-empty)
    std::vector<std::vector<BhNode>> flat_pools; // per pool: its flattened form with pool-local skips (preflatten)
    bool preflattened = false;
    std::vector<std::vector<Event>> queues;     // per-bucket insert queues of the last build (reused)
    std::vector<int> pbucket;                   // scratch of the threaded build (reused)
    std::vector<Event> sorted;                  // scratch: particles grouped by bucket, index order kept
    size_t node_count() const;
    // status: 0 ok, else the NBX_ERR_* code standing in for the reference panic
    // preflatten: the threaded build also flattens every bucket subtree right after replaying it (while it is hot in
    // the cache) into flat_pools; flatten_write then only copies the pieces into place and rebases their skip pointers
    // route (optional): an external implementation of phase 1a/1c of the threaded build -- given the frozen top
    // levels it must fill, for the bodies warm .. n-1, pbucket[i - warm] = bucket of body i, `sorted` = their insert
    // events grouped by bucket with the index order kept inside every bucket (depth = level of the bucket root), and
    // offset[b] .. offset[b+1] = bucket b's range in `sorted`.  Returns false to make the build do it itself.
    // (The engine routes and scatters on the GPU, where the positions already are: bh_front.hip.)
    struct TopView {
        const Node* top;            // nodes of the top levels; top[k].first_child >= 0 for pass-through nodes
        const int* bucket_of;       // per top node: bucket id, or -1 for a pass-through node
        int ntop;
        const int* bucket_depth;    // per bucket: level of its root (the depth its inserts start at)
        int nb;
    };
    using RouteFn = std::function<bool(const TopView& view, int warm, int rest, int* pbucket, Event* sorted, size_t* offset)>;
    int build(const float* px, const float* py, const float* m, int n, bool preflatten = false, const RouteFn* route = nullptr);
    // pre-order, all nodes (including empty exteriors), rows of 8 floats (see nbx_bh_tree_dump)
    int dump_preorder(float* rows, int cap) const;
    // pre-order with empty exterior nodes dropped + skip pointers, for the GPU traversal (serial)
    void flatten(std::vector<BhNode>& out) const;
    // the same for a tree built sequentially (forest == false), straight into a caller buffer that holds at least
    // nodes.size() records (e.g. pinned memory); returns the number of records written
    size_t flatten_into(BhNode* out) const;
    // the same array produced by host threads straight into a caller buffer (e.g. pinned memory):
    // prepare() returns the node count, write() fills out[0..count)
    struct FlatPlan {
        struct Item { int node; int piece; int end_item; size_t offset; };
        std::vector<Item> items;                  // top-level nodes and bucket subtrees, in pre-order
        std::vector<size_t> piece_size;           // live nodes of every bucket subtree (its span in the array)
        size_t total = 0;
    };
    size_t flatten_prepare(FlatPlan& plan) const;
    // chunk_done (optional): called on the calling thread with [first, last) node ranges of `out` as soon as a
    // prefix of at least chunk_nodes more nodes is complete (the ranges tile [0, total) in order)
    void flatten_write(const FlatPlan& plan, BhNode* out, const std::function<void(size_t, size_t)>& chunk_done = {},
                       size_t chunk_nodes = 262144) const;
};

}  // namespace nbx
