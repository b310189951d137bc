// host_ops.h -- host-side parts of the nb_* surface that stay on the CPU by design:
// presets (nbody.rs:39-104), framebuffer splat (nbody.rs:482-617) and the Barnes-Hut quadtree
// build (nbody.rs:203-331, :388-415; north_star: "octree build stays on host").
// Internal header; the public ABI is include/nbody_mi355x.h.
#pragma once
#include <cstdint>
#include <vector>

#include "kernels.h"

namespace nbx {

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

// nbody.rs:482-617; fb is w*h ABGR words, cleared here
void draw_particles(const float* px, const float* py, const float* vx, const float* vy, int n, int32_t w,
                    int32_t h, uint32_t* fb);

// Quadtree with index-linked nodes. Node k's children (if any) are first_child[k] .. +3 in the
// reference's order [UL, UR, LL, LR] (nbody.rs:295-300).
struct QuadTree {
    struct Node {
        float x1, y1, x2, y2;  // AABB
        float px, py, m;       // COM + total mass (interior) / particle (exterior)
        int32_t first_child;   // -1 = exterior
    };
    std::vector<Node> nodes;   // nodes[0] = root
    // status: 0 ok, else the NBX_ERR_* code standing in for the reference panic
    int build(const float* px, const float* py, const float* m, int n);
    // pre-order, all nodes (including empty exteriors), rows of 8 floats (see nbx_bh_tree_dump)
    int dump_preorder(float* rows, int cap) const;
    // pre-order with empty exterior nodes dropped + skip pointers, for the GPU traversal
    void flatten(std::vector<BhNode>& out) const;
};

}  // namespace nbx
