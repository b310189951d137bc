// host_tree.cpp -- the Barnes-Hut quadtree of the reference on the host (nbody.rs:203-331, :388-415): node-for-node its
// sequential insertion (threaded below a frozen top, result-identical for any thread count) and the flattening into the
// pre-order records the device walks.  COMPILED WITH -ffp-contract=off: the centre-of-mass update must round exactly like the
// reference (rustc never contracts a*b+c).
#include <sched.h>

#include "host_ops.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include <unistd.h>

#include "../../include/nbody_mi355x.h"

namespace nbx {

// ---- quadtree --------------------------------------------------------------------------------
//
// The reference inserts recursively into a Box-linked tree (nbody.rs:226-284). Here the same
// decisions run as one iterative descent over an index-linked node pool:
//   interior  -> add_mass, descend to quadrant_from_point, depth+1                 (:234-240)
//   exterior, empty or within EPS of the resident -> add_mass (merge)              (:249-260)
//   exterior, occupied -> split: children created, the resident re-inserted (it lands in an empty
//             child at depth+2), then the newcomer continues from THIS node at depth+1 (:271-281)
// Depth accounting follows the reference exactly (insert(..., depth+1) on the same node).

namespace {

inline bool add_mass(QuadTree::Node& nd, float px, float py, float m)   // nbody.rs:303-320
{
    if (!(m > 0.0f)) return false;                                       // :304
    if (nd.m == 0.0f) {                                                  // :305 exact copy
        nd.px = px; nd.py = py; nd.m = m;
    } else {
        const float inv_msum = 1.0f / (nd.m + m);                        // :315
        nd.px = (nd.px * nd.m + px * m) * inv_msum;                      // :316
        nd.py = (nd.py * nd.m + py * m) * inv_msum;                      // :317
        nd.m += m;                                                       // :318
    }
    return true;
}

inline int quadrant(const QuadTree::Node& nd, float x, float y)         // nbody.rs:322-331 -> [UL,UR,LL,LR]
{
    const float cx = (nd.x1 + nd.x2) * 0.5f;
    const float cy = (nd.y1 + nd.y2) * 0.5f;
    return (y < cy ? 2 : 0) + (x < cx ? 0 : 1);
}

}  // namespace

namespace {

using Event = QuadTree::Event;   // one pending Node::insert(px, py, m, depth) call

// The reference insertion (nbody.rs:226-284) as an iterative descent over an index-linked pool.
// `TOP` mode is the first phase of the threaded build: nodes carry their tree level, and an insert that
// is about to enter a node at level == limit is not executed but queued on that node's bucket, in
// arrival order.  Replaying a bucket's queue later, on its own, reproduces the sequential result
// exactly: a subtree's state depends only on the ordered sequence of inserts that reach its root.
struct Builder {
    std::vector<QuadTree::Node>& pool;
    std::vector<uint8_t>* level = nullptr;               // TOP mode only
    std::vector<int>* bucket_of = nullptr;               // TOP mode only: node -> bucket id or -1
    std::vector<std::vector<QuadTree::Event>>* buckets = nullptr;  // TOP mode only (capacity reused across builds)
    size_t* used = nullptr;                                        // TOP mode only: buckets in use
    int limit = -1;

    int split(int k)                                                      // create_children, :286-301
    {
        const QuadTree::Node nd = pool[k];
        const float cx = (nd.x1 + nd.x2) * 0.5f;
        const float cy = (nd.y1 + nd.y2) * 0.5f;
        if (!(cx > nd.x1 || cx < nd.x2 || cy > nd.y1 || cy < nd.y2)) return NBX_ERR_TREE;  // :293
        const int c = (int)pool.size();
        pool.push_back(QuadTree::Node{nd.x1, cy, cx, nd.y2, 0.0f, 0.0f, 0.0f, -1});      // UL :296
        pool.push_back(QuadTree::Node{cx, cy, nd.x2, nd.y2, 0.0f, 0.0f, 0.0f, -1});      // UR :297
        pool.push_back(QuadTree::Node{nd.x1, nd.y1, cx, cy, 0.0f, 0.0f, 0.0f, -1});      // LL :298
        pool.push_back(QuadTree::Node{cx, nd.y1, nd.x2, cy, 0.0f, 0.0f, 0.0f, -1});      // LR :299
        pool[k].first_child = c;
        if (level) {
            const uint8_t lv = (uint8_t)((*level)[k] + 1);
            for (int q = 0; q < 4; q++) {
                level->push_back(lv);
                if (lv == limit) {
                    bucket_of->push_back((int)*used);
                    if (*used == buckets->size()) buckets->emplace_back();
                    ++*used;
                } else {
                    bucket_of->push_back(-1);
                }
            }
        }
        return NBX_OK;
    }

    template <bool TOP>
    int insert(int k, const Event ev)
    {
        const float EPS = kEps;
        const float qx = ev.x, qy = ev.y, qm = ev.m;
        unsigned depth = ev.depth;
        for (;;) {
            if (TOP && (*bucket_of)[k] >= 0) {                            // hand over to the subtree's owner
                (*buckets)[(*bucket_of)[k]].push_back(Event{qx, qy, qm, depth});
                return NBX_OK;
            }
            if (depth > 50) return NBX_ERR_TREE_DEPTH;                    // :230
            if (pool[k].first_child >= 0) {                               // :234
                if (!add_mass(pool[k], qx, qy, qm)) return NBX_ERR_TREE;  // :236
                k = pool[k].first_child + quadrant(pool[k], qx, qy);      // :237-240
                depth += 1;
                continue;
            }
            QuadTree::Node& nd = pool[k];
            const bool too_close = std::fabs(nd.px - qx) < EPS && std::fabs(nd.py - qy) < EPS;  // :249
            if (nd.m == 0.0f || too_close) {                              // :250
                if (!add_mass(nd, qx, qy, qm)) return NBX_ERR_TREE;       // :260
                return NBX_OK;
            }
            if (!(nd.px != qx || nd.py != qy)) return NBX_ERR_TREE;       // :267
            const float ox = nd.px, oy = nd.py, om = nd.m;                // :271-273
            pool[k].px = 0.0f; pool[k].py = 0.0f; pool[k].m = 0.0f;       // :274-276
            const int rc = split(k);                                      // :277 (invalidates `nd`)
            if (rc != NBX_OK) return rc;
            // self.insert(original, depth+1)  :278 -> interior branch: add_mass on the emptied node
            // (exact copy), then the child at depth+2, which is empty -> exact copy again.
            if (depth + 1 > 50) return NBX_ERR_TREE_DEPTH;
            if (!add_mass(pool[k], ox, oy, om)) return NBX_ERR_TREE;
            const int child = pool[k].first_child + quadrant(pool[k], ox, oy);
            if (TOP && (*bucket_of)[child] >= 0) {
                (*buckets)[(*bucket_of)[child]].push_back(Event{ox, oy, om, depth + 2});
            } else {
                if (depth + 2 > 50) return NBX_ERR_TREE_DEPTH;
                if (!add_mass(pool[child], ox, oy, om)) return NBX_ERR_TREE;
            }
            depth += 1;                                                   // :281 self.insert(new, depth+1)
        }
    }
};


}  // namespace

static int flatten_subtree_into(const std::vector<QuadTree::Node>& nodes, int root, BhNode* out, int base);

int QuadTree::build(const float* px, const float* py, const float* m, int n, bool preflatten, const RouteFn* route)
{
    nodes.clear();
    forest = false;
    preflattened = false;
    n_buckets = 0;
    float x1 = 3.40282347e+38f, y1 = 3.40282347e+38f, x2 = -3.40282347e+38f, y2 = -3.40282347e+38f;  // :388-391
    const int threads = host_threads();
    auto minmax = [&](int a, int b, float* r) {                           // :392-398 strict < / >
        float lx = r[0], ly = r[1], hx = r[2], hy = r[3];
        for (int i = a; i < b; i++) {
            lx = px[i] < lx ? px[i] : lx;
            ly = py[i] < ly ? py[i] : ly;
            hx = px[i] > hx ? px[i] : hx;
            hy = py[i] > hy ? py[i] : hy;
        }
        r[0] = lx; r[1] = ly; r[2] = hx; r[3] = hy;
    };
    if (n >= 65536 && threads > 1) {   // min / max do not depend on the order: same box from any split
        const int parts = std::min(threads, 16);
        std::vector<float> part((size_t)parts * 16);   // one cache line per part
        parallel_for(parts, [&](int t) {
            float* r = &part[(size_t)t * 16];
            r[0] = x1; r[1] = y1; r[2] = x2; r[3] = y2;
            minmax((int)((long long)n * t / parts), (int)((long long)n * (t + 1) / parts), r);
        });
        for (int t = 0; t < parts; t++) {
            const float* r = &part[(size_t)t * 16];
            x1 = r[0] < x1 ? r[0] : x1; y1 = r[1] < y1 ? r[1] : y1;
            x2 = r[2] > x2 ? r[2] : x2; y2 = r[3] > y2 ? r[3] : y2;
        }
    } else {
        float r[4] = {x1, y1, x2, y2};
        minmax(0, n, r);
        x1 = r[0]; y1 = r[1]; x2 = r[2]; y2 = r[3];
    }
    if (n < 4096 || threads < 2) {
        // sequential: exactly the reference's loop (nbody.rs:413-415, particle-index order)
        nodes.reserve((size_t)n * 3 + 8);
        nodes.push_back(Node{x1, y1, x2, y2, 0.0f, 0.0f, 0.0f, -1});      // :410
        Builder b{nodes};
        for (int i = 0; i < n; i++) {
            const int rc = b.insert<false>(0, Event{px[i], py[i], m[i], 0});
            if (rc != NBX_OK) return rc;
        }
        return NBX_OK;
    }

    // ---- threaded, result-identical build --------------------------------------------------------
    // A subtree's final state depends only on the ordered sequence of insert() calls that reach its root, and
    // an interior node only folds the passing particle into its centre of mass and forwards it by geometry.
    // So:
    //   phase 0 (sequential, exact): the first `warm` particles run the real algorithm on the top `limit`
    //            levels; inserts reaching a level-`limit` node are queued on it ("bucket").
    //   freeze : every top node that is still exterior becomes a bucket too (its state = the pool root).  All
    //            remaining top nodes are interior and stay interior: they are pure pass-through from now on.
    //   phase 1 (parallel): (a) route every remaining particle by geometry to its bucket; (b) per top level, a
    //            thread folds the particles into that level's pass-through nodes IN INDEX ORDER (nodes of one
    //            level are disjoint, levels are independent); (c) a stable counting sort by bucket appends the
    //            particles to the bucket queues in index order with depth = bucket level (no split can happen
    //            on the way, so the reference's depth counter is just the number of descents).
    //   phase 2 (parallel over buckets): replay each queue on a private pool (node 0 = the bucket root).
    const bool timing = std::getenv("NBX_TIMING") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    // bucket level, sequential warm-up and worker count by size (measured on the target host, profiles/README.md):
    // small systems want few workers (waking 31 threads for 10 000 bodies costs more than it buys) and a short warm-up
    const int limit = n >= 262144 ? 8 : (n >= 16384 ? 5 : 4);   // 8 levels under a 2 048-body warm-up: ~4 000 buckets at 1 M bodies
    // (the warm-up is sequential: with the routing on the device from 16 384 bodies on, 1 024 bodies of it instead of 2 048 / 8 192
    //  are 5-15 % of a host-tree step at 30 000 .. 131 072 bodies -- discs, orbits, Plummer spheres, a collapsed core alike)
    const int warm = std::min(n, n >= 262144 ? 2048 : 1024);   // (262 144-body Plummer disc: 6.4-7.5 ms per host-tree step with 8 192 / 7 levels, 5.0-5.6 like this)
    std::vector<Node>& top = nodes;
    std::vector<uint8_t> level;
    bucket_of.clear();
    for (auto& q : queues) q.clear();
    size_t used_queues = 0;
    top.reserve(8192);
    top.push_back(Node{x1, y1, x2, y2, 0.0f, 0.0f, 0.0f, -1});
    level.push_back(0);
    bucket_of.push_back(-1);
    {
        Builder tb{top, &level, &bucket_of, &queues, &used_queues, limit};
        for (int i = 0; i < warm; i++) {
            const int rc = tb.insert<true>(0, Event{px[i], py[i], m[i], 0});
            if (rc != NBX_OK) return rc;
        }
    }
    // freeze: exterior top nodes above the limit become buckets as they are
    const int ntop = (int)top.size();
    for (int k = 0; k < ntop; k++) {
        if (bucket_of[k] < 0 && top[k].first_child < 0) {
            bucket_of[k] = (int)used_queues;
            if (used_queues == queues.size()) queues.emplace_back();
            ++used_queues;
        }
    }
    const int nb = (int)used_queues;
    root_of.assign(nb, -1);
    for (int k = 0; k < ntop; k++)
        if (bucket_of[k] >= 0) root_of[bucket_of[k]] = k;
    const auto tp1 = std::chrono::steady_clock::now();

    const int rest = n - warm;
    pbucket.resize((size_t)rest);                    // bucket of particle warm+i (scratch reused across builds)
    std::vector<int> parent(ntop, -1);
    for (int k = 0; k < ntop; k++)
        if (top[k].first_child >= 0 && bucket_of[k] < 0)
            for (int c = 0; c < 4; c++) parent[top[k].first_child + c] = k;
    std::atomic<int> bad_mass{0};
    const int nt = std::max(1, n >= 32768 ? threads : std::min(threads, n >= 16384 ? 8 : 4));
    auto run_threads = [&](int count, const std::function<void(int)>& fn) { parallel_for(count, fn); };
    // ancestors of every bucket root, per level (anc[b][l] = pass-through node at level l, or -1)
    int max_level = 0;
    for (int k = 0; k < ntop; k++) max_level = std::max<int>(max_level, level[k]);
    std::vector<int> anc((size_t)nb * (size_t)(max_level + 1), -1);
    for (int b2 = 0; b2 < nb; b2++)
        for (int k = parent[root_of[b2]]; k >= 0; k = parent[k]) anc[(size_t)b2 * (max_level + 1) + level[k]] = k;
    const int fold_levels = max_level;                                   // pass-through nodes live on levels 0..max_level-1
    const int slices = std::max(1, std::min(4, nt / std::max(1, fold_levels)));
    const std::vector<Node> top0(top.begin(), top.begin() + ntop);   // the top levels as the warm-up left them
    // (b) folds: one task per (level, slice of that level's nodes) adds the particles to that level's pass-through
    // nodes in index order.  They run beside everything below and are only waited for at the very end.
    auto fold = [&](int lvl, int slice) {
        const size_t stride = (size_t)(max_level + 1);
        // work on a private copy: 32-byte nodes of different levels share cache lines in `top`, and every fold
        // thread writes its nodes a million times (false sharing cost 5x here).  The copy comes from a snapshot taken
        // before any fold started, so no task ever reads what another one is writing back.
        std::vector<Node> mine(top0);
        for (int i = 0; i < rest; i++) {
            const int k = anc[(size_t)pbucket[i] * stride + lvl];
            if (k < 0 || (k % slices) != slice) continue;
            add_mass(mine[k], px[warm + i], py[warm + i], m[warm + i]);   // interior: nbody.rs:236, index order
        }
        for (int k = 0; k < ntop; k++)
            if (level[k] == lvl && bucket_of[k] < 0 && top[k].first_child >= 0 && (k % slices) == slice) {
                top[k].px = mine[k].px; top[k].py = mine[k].py; top[k].m = mine[k].m;   // only what the fold changed
            }
    };
    // the root sees EVERY particle, a 1 M-long chain of dependent adds (~4 ms at 1 M bodies: the longest task of the
    // build, the reference's own serial fold).  It needs no routing result, so it starts before anything else.
    auto fold_root = [&]() {
        Node mine = top0[0];
        for (int i = 0; i < rest; i++)
            if (!add_mass(mine, px[warm + i], py[warm + i], m[warm + i])) bad_mass.store(1);   // nbody.rs:304
        top[0].px = mine.px; top[0].py = mine.py; top[0].m = mine.m;
    };
    TaskGroup folds;   // declared after everything its tasks capture: its destructor waits for them on every exit path
    if (fold_levels > 0) folds.run(fold_root);
    std::vector<size_t> offset((size_t)nb + 1, 0);
    sorted.resize((size_t)rest);
    bool routed = false;
    if (route && *route) {   // phases (a) and (c) elsewhere (the engine does them on the GPU)
        std::vector<int> bucket_depth((size_t)nb);
        for (int b2 = 0; b2 < nb; b2++) bucket_depth[(size_t)b2] = level[root_of[b2]];
        const TopView view{top.data(), bucket_of.data(), ntop, bucket_depth.data(), nb};
        routed = (*route)(view, warm, rest, pbucket.data(), sorted.data(), offset.data());
        if (routed && fold_levels == 0)   // nobody else looks at the masses then
            for (int i = 0; i < rest; i++)
                if (!(m[warm + i] > 0.0f)) bad_mass.store(1);
    }
    auto tpa = std::chrono::steady_clock::now();
    if (!routed) {
        // (a) routing
        run_threads(nt, [&](int t) {
            const int lo = (int)((long long)rest * t / nt), hi = (int)((long long)rest * (t + 1) / nt);
            for (int i = lo; i < hi; i++) {
                const float qx = px[warm + i], qy = py[warm + i];
                if (!(m[warm + i] > 0.0f)) bad_mass.store(1);
                int k = 0;
                while (bucket_of[k] < 0) k = top[k].first_child + quadrant(top[k], qx, qy);
                pbucket[i] = bucket_of[k];
            }
        });
        if (bad_mass.load()) return NBX_ERR_TREE;                              // nbody.rs:304
        tpa = std::chrono::steady_clock::now();
    }
    for (int lvl = 1; lvl < fold_levels; lvl++)
        for (int sl = 0; sl < slices; sl++) folds.run([&fold, lvl, sl] { fold(lvl, sl); });
    if (!routed) {
        // (c) histogram + stable scatter of the particles into per-bucket queues (index order kept)
        std::vector<std::vector<size_t>> hist(nt, std::vector<size_t>((size_t)nb, 0));
        run_threads(nt, [&](int t) {
            const int lo = (int)((long long)rest * t / nt), hi = (int)((long long)rest * (t + 1) / nt);
            for (int i = lo; i < hi; i++) hist[t][(size_t)pbucket[i]]++;
        });
        size_t run = 0;
        for (int b2 = 0; b2 < nb; b2++) {
            offset[b2] = run;
            for (int t = 0; t < nt; t++) { const size_t c = hist[t][(size_t)b2]; hist[t][(size_t)b2] = run; run += c; }
        }
        offset[nb] = run;
        run_threads(nt, [&](int t) {
            const int lo = (int)((long long)rest * t / nt), hi = (int)((long long)rest * (t + 1) / nt);
            for (int i = lo; i < hi; i++) {
                const int b2 = pbucket[i];
                sorted[hist[t][(size_t)b2]++] = Event{px[warm + i], py[warm + i], m[warm + i], (unsigned)level[root_of[b2]]};
            }
        });
    }
    const auto tp2 = std::chrono::steady_clock::now();

    // Phase 2 (parallel over buckets): replay each queue on a private pool whose node 0 is the bucket root.
    if ((int)pools.size() < nb) pools.resize(nb);
    pool_live.assign((size_t)nb, 0);
    if (preflatten && (int)flat_pools.size() < nb) flat_pools.resize(nb);
    std::vector<int> status(nb, NBX_OK);
    std::vector<int> order(nb);
    for (int b2 = 0; b2 < nb; b2++) order[b2] = b2;
    auto qsize = [&](int b2) { return queues[b2].size() + (offset[b2 + 1] - offset[b2]); };
    std::sort(order.begin(), order.end(), [&](int a2, int b2) { return qsize(a2) > qsize(b2); });
    std::atomic<int> next{0};
    run_threads(std::max(1, std::min(nt, nb)), [&](int) {
        for (;;) {
            const int t = next.fetch_add(1);
            if (t >= nb) return;
            const int b2 = order[t];
            std::vector<Node>& pool = pools[b2];
            pool.clear();
            pool.reserve(qsize(b2) * 3 + 8);
            pool.push_back(top[root_of[b2]]);
            Builder lb{pool};
            int rc = NBX_OK;
            for (const Event& ev : queues[b2]) {                          // phase-0 arrivals first ...
                rc = lb.insert<false>(0, ev);
                if (rc != NBX_OK) break;
            }
            for (size_t i = offset[b2]; rc == NBX_OK && i < offset[b2 + 1]; i++) rc = lb.insert<false>(0, sorted[i]);  // ... then the rest
            status[b2] = rc;
            size_t live = 0;   // what the flattened subtree will hold (empty exterior nodes are dropped); counted while hot
            for (const Node& nd : pool) live += (nd.first_child >= 0 || nd.m != 0.0f) ? 1 : 0;
            pool_live[(size_t)b2] = live;
            if (preflatten && rc == NBX_OK) {
                std::vector<BhNode>& fp = flat_pools[(size_t)b2];
                fp.resize(live);
                if (live) flatten_subtree_into(pool, 0, fp.data(), 0);
            }
        }
    });
    const auto tp3a = std::chrono::steady_clock::now();
    folds.wait();   // the pass-through nodes (touched by nobody else) are final now
    if (bad_mass.load()) return NBX_ERR_TREE;                                  // nbody.rs:304
    for (int b2 = 0; b2 < nb; b2++)
        if (status[b2] != NBX_OK) return status[b2];
    // The tree stays a forest: `nodes` = top levels, pools[b] = subtree of bucket b (local indices,
    // node 0 = the bucket root, which supersedes nodes[root_of[b]]).  Traversals below understand both.
    forest = true;
    preflattened = preflatten;
    n_buckets = nb;
    if (timing) {
        const auto tp3 = std::chrono::steady_clock::now();
        auto ms = [](auto a2, auto b2) { return std::chrono::duration<double, std::milli>(b2 - a2).count(); };
        size_t big = 0;
        for (int b2 = 0; b2 < nb; b2++) big = std::max(big, qsize(b2));
        std::fprintf(stderr,
                     "[nbx] tree build n=%d threads=%d limit=%d buckets=%d (largest %zu): warm-up %.2f ms, route %.2f ms, scatter %.2f ms, "
                     "subtrees %.2f ms, folds (remaining) %.2f ms\n",
                     n, nt, limit, nb, big, ms(tp0, tp1), ms(tp1, tpa), ms(tpa, tp2), ms(tp2, tp3a), ms(tp3a, tp3));
    }
    return NBX_OK;
}

size_t QuadTree::node_count() const
{
    size_t c = nodes.size();
    if (forest)
        for (int b = 0; b < n_buckets; b++) c += pools[b].size() - 1;
    return c;
}

int QuadTree::dump_preorder(float* rows, int cap) const
{
    if (nodes.empty()) return 0;
    int count = 0;
    struct Ref { int pool; int idx; };   // pool -1 = `nodes`
    std::vector<Ref> stack;
    stack.push_back(Ref{-1, 0});
    while (!stack.empty()) {
        Ref r = stack.back();
        stack.pop_back();
        if (r.pool < 0 && forest && bucket_of[r.idx] >= 0) r = Ref{bucket_of[r.idx], 0};
        const Node& nd = r.pool < 0 ? nodes[r.idx] : pools[r.pool][r.idx];
        if (count < cap && rows) {
            float* o = rows + 8 * (size_t)count;
            o[0] = nd.x1; o[1] = nd.y1; o[2] = nd.x2; o[3] = nd.y2;
            o[4] = nd.px; o[5] = nd.py; o[6] = nd.m; o[7] = nd.first_child >= 0 ? 1.0f : 0.0f;
        }
        count++;
        if (nd.first_child >= 0)
            for (int c = 3; c >= 0; c--) stack.push_back(Ref{r.pool, nd.first_child + c});
    }
    return count;
}

// Pre-order flattening with skip pointers, empty exterior nodes dropped (they contribute (0,0),
// nbody.rs:368).  `base` is added to every skip pointer (position of the subtree in the final array).
static void flatten_subtree(const std::vector<QuadTree::Node>& nodes, int root, std::vector<BhNode>& out)
{
    struct Frame { int node; int slot; int next_child; };
    std::vector<Frame> st;
    auto emit = [&](int k) -> int {
        const QuadTree::Node& nd = nodes[k];
        BhNode b;
        b.px = nd.px; b.py = nd.py; b.m = nd.m; b.s = nd.x2 - nd.x1;   // s = x-extent, nbody.rs:341
        b.skip = 0; b.interior = nd.first_child >= 0 ? 1 : 0; b.q = bh_node_q(b.s, b.interior != 0); b.pad1 = 0;
        out.push_back(b);
        return (int)out.size() - 1;
    };
    st.push_back(Frame{root, emit(root), 0});
    while (!st.empty()) {
        Frame& f = st.back();
        const QuadTree::Node& nd = nodes[f.node];
        if (nd.first_child < 0 || f.next_child == 4) {
            out[f.slot].skip = (int)out.size();
            st.pop_back();
            continue;
        }
        const int c = nd.first_child + f.next_child++;
        const QuadTree::Node& ch = nodes[c];
        if (ch.first_child < 0 && ch.m == 0.0f) continue;
        const int slot = emit(c);
        st.push_back(Frame{c, slot, 0});
    }
}

// same walk, written straight into `out` (which must hold the subtree's live-node count); skips are absolute:
// `base` = position of the subtree root in the final array
static int flatten_subtree_into(const std::vector<QuadTree::Node>& nodes, int root, BhNode* out, int base)
{
    struct Frame { int node; int slot; int next_child; };
    Frame st[128];   // depth <= 52 (the build rejects deeper trees)
    int sp = 0, count = 0;
    auto emit = [&](int k) -> int {
        const QuadTree::Node& nd = nodes[k];
        BhNode b;
        b.px = nd.px; b.py = nd.py; b.m = nd.m; b.s = nd.x2 - nd.x1;
        b.skip = 0; b.interior = nd.first_child >= 0 ? 1 : 0; b.q = bh_node_q(b.s, b.interior != 0); b.pad1 = 0;
        out[count] = b;
        return count++;
    };
    st[sp++] = Frame{root, emit(root), 0};
    while (sp > 0) {
        Frame& f = st[sp - 1];
        const QuadTree::Node& nd = nodes[f.node];
        if (nd.first_child < 0 || f.next_child == 4) {
            out[f.slot].skip = base + count;
            sp--;
            continue;
        }
        if (f.next_child == 0) {
            // the pool is in creation order, the walk is depth-first: every children block (4 nodes = 2 cache
            // lines) is a likely miss. Ask for the four grandchildren blocks now; all but the first are only
            // needed after whole subtrees have been written.
            for (int g = 0; g < 4; g++) {
                const int gc = nodes[nd.first_child + g].first_child;
                if (gc >= 0) {
                    __builtin_prefetch(&nodes[gc]);
                    __builtin_prefetch(&nodes[gc + 2]);
                }
            }
        }
        const int c = nd.first_child + f.next_child++;
        const QuadTree::Node& ch = nodes[c];
        if (ch.first_child < 0 && ch.m == 0.0f) continue;
        const int slot = emit(c);
        st[sp++] = Frame{c, slot, 0};
    }
    return count;
}

size_t QuadTree::flatten_into(BhNode* out) const
{
    if (forest || nodes.empty()) return 0;
    const Node& root = nodes[0];
    if (root.first_child < 0 && root.m == 0.0f) return 0;   // empty tree
    return (size_t)flatten_subtree_into(nodes, 0, out, 0);
}

void QuadTree::flatten(std::vector<BhNode>& out) const
{
    out.clear();
    if (nodes.empty()) return;
    if (forest) {   // generic path: lay the pieces out serially
        FlatPlan plan;
        const size_t count = flatten_prepare(plan);
        out.resize(count);
        if (count) flatten_write(plan, out.data());
        return;
    }
    const Node& root = nodes[0];
    if (root.first_child < 0 && root.m == 0.0f) return;   // empty tree
    out.reserve(nodes.size());
    flatten_subtree(nodes, 0, out);
}

// Threaded flattening of a forest: prepare() walks the top levels serially and counts the live nodes of every
// bucket subtree in parallel, which fixes each piece's position in the pre-order array; write() then flattens
// every bucket subtree straight into its span of the destination (e.g. pinned memory), in parallel.
size_t QuadTree::flatten_prepare(FlatPlan& plan) const
{
    plan.items.clear();
    plan.total = 0;
    if (nodes.empty()) return 0;
    auto eff = [&](int k) -> const Node& { return (forest && bucket_of[k] >= 0) ? pools[bucket_of[k]][0] : nodes[k]; };
    const Node& root = eff(0);
    if (root.first_child < 0 && root.m == 0.0f) return 0;
    struct Frame { int node; int item; int next_child; };
    std::vector<Frame> st;
    auto add_item = [&](int node) -> int {
        FlatPlan::Item it;
        it.node = node; it.end_item = -1; it.offset = 0;
        it.piece = (forest && bucket_of[node] >= 0) ? bucket_of[node] : -1;
        plan.items.push_back(it);
        return (int)plan.items.size() - 1;
    };
    st.push_back(Frame{0, add_item(0), 0});
    while (!st.empty()) {
        Frame& f = st.back();
        const Node& nd = nodes[f.node];
        const bool leaf_like = plan.items[f.item].piece >= 0 || nd.first_child < 0;
        if (leaf_like || f.next_child == 4) {
            plan.items[f.item].end_item = (int)plan.items.size();
            st.pop_back();
            continue;
        }
        const int c = nd.first_child + f.next_child++;
        const Node& ch = eff(c);
        if (ch.first_child < 0 && ch.m == 0.0f) continue;   // empty exterior: dropped
        const int item = add_item(c);
        st.push_back(Frame{c, item, 0});
    }
    // live-node count of every referenced bucket (what its flattened piece will hold): taken by the build
    plan.piece_size.assign((size_t)n_buckets, 0);
    for (const auto& it : plan.items)
        if (it.piece >= 0) plan.piece_size[(size_t)it.piece] = pool_live[(size_t)it.piece];
    size_t off = 0;
    for (auto& it : plan.items) {
        it.offset = off;
        off += it.piece >= 0 ? plan.piece_size[(size_t)it.piece] : 1;
    }
    plan.total = off;
    return off;
}

void QuadTree::flatten_write(const FlatPlan& plan, BhNode* out, const std::function<void(size_t, size_t)>& chunk_done,
                             size_t chunk_nodes) const
{
    const int ni = (int)plan.items.size();
    std::atomic<int> next{0};
    // items finish out of order; done[] lets the calling thread find the finished PREFIX of the array and hand it to
    // chunk_done (the engine starts the host-to-device copy of that range while the rest is still being written)
    std::unique_ptr<std::atomic<unsigned char>[]> done;
    if (chunk_done) {
        done.reset(new std::atomic<unsigned char>[(size_t)ni]);
        for (int i = 0; i < ni; i++) done[(size_t)i].store(0, std::memory_order_relaxed);
    }
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= ni) return;
            const FlatPlan::Item& it = plan.items[i];
            if (it.piece < 0) {
                const Node& nd = nodes[it.node];
                BhNode b;
                b.px = nd.px; b.py = nd.py; b.m = nd.m; b.s = nd.x2 - nd.x1;
                b.interior = nd.first_child >= 0 ? 1 : 0; b.q = bh_node_q(b.s, b.interior != 0); b.pad1 = 0;
                b.skip = (int)(it.end_item < ni ? plan.items[it.end_item].offset : plan.total);
                out[it.offset] = b;
            } else if (preflattened) {
                const std::vector<BhNode>& src = flat_pools[(size_t)it.piece];
                BhNode* dst = out + it.offset;
                const int base = (int)it.offset;
                for (size_t j = 0; j < src.size(); j++) {
                    BhNode b = src[j];
                    b.skip += base;
                    dst[j] = b;
                }
            } else {
                flatten_subtree_into(pools[it.piece], 0, out + it.offset, (int)it.offset);
            }
            if (done) done[(size_t)i].store(1, std::memory_order_release);
        }
    };
    int nt = std::max(1, std::min(host_threads(), ni));
    if (plan.total < 262144) nt = std::min(nt, plan.total < 65536 ? 2 : 4);   // small trees: waking the whole pool costs more than the copy
    if (!chunk_done || nt == 1) {   // nt == 1: nobody to write while this thread watches the prefix
        parallel_for(nt, [&work](int) { work(); });
        if (chunk_done && plan.total > 0) chunk_done(0, plan.total);
        return;
    }
    TaskGroup writers;
    for (int t = 0; t < nt - 1; t++) writers.run(work);
    int w = 0;            // items [0, w) are finished
    size_t sent = 0;      // nodes [0, sent) were handed over
    while (w < ni) {
        if (!done[(size_t)w].load(std::memory_order_acquire)) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));   // not a worker: do not burn the CPU quota
            continue;
        }
        while (w < ni && done[(size_t)w].load(std::memory_order_acquire)) w++;
        const size_t upto = w < ni ? plan.items[(size_t)w].offset : plan.total;
        if (upto - sent >= chunk_nodes || w == ni) {
            if (upto > sent) chunk_done(sent, upto);
            sent = upto;
        }
    }
    writers.wait();
}

}  // namespace nbx
