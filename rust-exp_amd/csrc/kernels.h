// kernels.h -- launch wrappers of the gfx950 kernels (internal; the public ABI is include/nbody_mi355x.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nbx {

constexpr int kTile = 256;   // sources per LDS tile == threads per workgroup (4 wave64)
constexpr float kEps = 0.0001f;  // nbody.rs:17

// Flattened Barnes-Hut node (pre-order, skip pointers). 32 bytes, two 16-B loads.
struct BhNode {
    float px, py, m, s;      // COM / particle position, mass, x-extent (x2-x1; nbody.rs:341)
    int32_t skip;            // index of the next node when this subtree is not opened
    int32_t interior;        // 1 = has children (nbody.rs:338), 0 = exterior (leaf)
    float q;                 // opening threshold of the fast walks: s*s for an interior node, -1 for a leaf, so that
                             // "q < theta^2 * d^2" is the MAC for interior nodes and always true for leaves
    int32_t pad1;            // device build: interior nodes before this one in pre-order (where the fast walk files its child group)
};
inline __host__ __device__ float bh_node_q(float s, bool interior) { return interior ? s * s : -1.0f; }

// Round 4: the fast walk's copy of the tree (bh_walk.hip).  One 80-byte record per OPENED node: the (x, y, m, T) of its up to four
// children -- 64 bytes, one s_load_dwordx16 -- present ones first, in the reference's child order (nbody.rs:295-300), then four
// child words.  T = the opening threshold of bh_threshold.h for an interior child (take <=> dist_sq > T: the reference's
// s/sqrt(dist_sq) < theta exactly), -1 for a leaf (always evaluated), +inf for an absent slot.  kid = BYTE offset of that child's
// own record (>= 0), -1 = leaf, -2 = absent.  Record 0 holds the root; the children of the r-th interior node in pre-order are
// record r + 1 -- the records follow the depth-first order of the walk, a node's first opened child is the next record in memory
// (device-built trees: BhNode::pad1 carries r; host-built trees use r = the node's pre-order index: gaps, never touched).
struct alignas(16) BhGroup {
    float4 c[4];
    int4 kid;
};
size_t bh_groups_count(int node_cap);        // records a tree of node_cap nodes needs
bool bh_groups_addressable(int node_cap);    // their byte offsets fit the 31 bits a child word has
// records from the flattened tree, with the step's theta.  n_nodes_or_cap = the node count, or (gated: the count is still on the
// device, bh_gate.h) the capacity of the node array -- the kernel then reads the count itself
// compact: the interior nodes carry their pre-order rank among interior nodes in pad1 (device build)
hipError_t launch_bh_groups(const BhNode* nodes, int n_nodes_or_cap, float theta, BhGroup* groups, bool compact, hipStream_t stream,
                            int* gate_counters = nullptr, int gate_node_cap = 0, int gate_crowd_limit = 0, int gate_queue_limit = 0);
// accelerations of the slab's bodies (fast mode).  wave && perm: one walk per wave (bodies in the spatial order perm; hand_scheduled:
// the assembly loop, else the compiler's), else one per lane; bit-identical results whichever runs
// kick (optional, wave form only): the kick-drift of the step (nbody.rs:453-471, what k_integrate_f2 does) applied by the walk itself
// as soon as a body's acceleration is complete -- legitimate because a walk reads no other body's position from posm (the group
// records hold copies) -- so a small system's step is one dependent kernel shorter.  out is not written then.  host_out: see BuildGate.
struct BhKick {
    float4* vel;      // [n_targets] this slab's velocities; nullptr = no kick (the walk writes accelerations to out)
    float4* posm;     // the same array the walk reads its bodies from
    float dt;         // (the reference's velocity kill outside +-55, nbody.rs:466-471, is always applied: this is the Barnes-Hut step)
    int* host_out;    // gated step: pinned words the build's counters are handed to (nullptr: none)
    float4* sorted;   // optional [n_targets]: the new positions once more, in the order the walks took the bodies (entry t = body
                      // perm[t]): the next build's sort walks the bodies in exactly that order and reads them from here, coalesced,
                      // instead of gathering 16 MB of random 16-byte records (round 5)
};
hipError_t launch_bh_walk_groups(const float4* posm, int lo, int n_targets, const BhGroup* groups, float2* out, hipStream_t stream,
                                 const unsigned* perm, bool wave, bool hand_scheduled, int* gate_counters = nullptr,
                                 int gate_node_cap = 0, int gate_crowd_limit = 0, int gate_queue_limit = 0,
                                 unsigned long long* trace = nullptr, const BhKick* kick = nullptr);
// trace (optional, 4 words per walk = workgroup): s_memrealtime (10 ns ticks) at its start and end, groups loaded (bit 31: redone with the LDS spill) | chunk << 32, HW_ID | XCC_ID << 32
int bh_walk_count(int n_targets, int* bodies_per_walk = nullptr);   // walks (workgroups) of the wave form, a multiple of 8
hipError_t launch_bh_count_groups(const float4* posm, int lo, int n_targets, const BhGroup* groups, unsigned long long* totals,
                                  hipStream_t stream);   // totals[0] children visited, [1] pair laws, [2] opening tests (visits of
                                                         // interior nodes), [3] groups loaded (per body)
hipError_t launch_bh_thresholds(const float* s, const float* theta, float* out, int count, hipStream_t stream);   // test hook

struct ForceLaunch {
    int grid, block, jsplit, bpt, dim, variant;
};

// K1: all-pairs accelerations for slab targets [lo, lo+n_targets) against tiles_total*kTile sources.
// acc_partial: [jsplit][acc_stride] float4 (ax, ay, az, unused).
// Variant 1 (k_force_tile_pk: packed math, sources through LDS tiles), bpt = 2 or 4 targets per thread.
hipError_t launch_force_tile(const float4* posm, int lo, int n_targets, int tiles_total, int jsplit, int bpt,
                             int dim, float4* acc_partial, int acc_stride, hipStream_t stream, ForceLaunch* info);

// K1, variants 6 / 7 (k_force_smem_pkw): the four waves of a workgroup share 256 targets and split the workgroup's source
// range; partial sums meet in LDS, so only `jsplit` slabs are written for 4 * jsplit source ranges. unit_mass: every body
// has mass `mass` (the per-interaction multiply leaves the loop); n_sources = true body count (no padding swept).
// exc_idx / exc_rec (unit_mass only, exc_count > 0): workgroup 0 also copies the source record of body exc_idx[k], with the
// weight (its mass - mass) the sweep leaves out, into exc_rec[k] -- the snapshot K2 / the force readout add afterwards.
// widened (K4, fp16 sources): the sources are read from this float4 array -- the half4 copy widened by launch_widen_half --
// instead of posm (targets stay posm); every target's interaction with its own image is then taken out by K2 / the
// force readout (SelfImage); info->variant = 17 / 18.
hipError_t launch_force_wave_split(const float4* posm, int lo, int n_targets, int tiles_total, int n_sources, int jsplit, int dim,
                                   bool unit_mass, float mass, float4* acc_partial, int acc_stride, hipStream_t stream,
                                   ForceLaunch* info, const int* exc_idx = nullptr, float4* exc_rec = nullptr, int exc_count = 0,
                                   const float4* widened = nullptr);
// widened[i] = float4(posh[i]) for i < count: the fp16 source copy as fp32 records, once per step (K4)
hipError_t launch_widen_half(const void* posh, float4* widened, int count, hipStream_t stream);

// K4: the packed sweep with sources read from a half4 (x,y,z,m) copy (8 B/body), targets fp32.
hipError_t launch_force_tile_half(const float4* posm, const void* posh, int lo, int n_targets, int tiles_total,
                                  int jsplit, int bpt, int dim, float4* acc_partial, int acc_stride, hipStream_t stream,
                                  ForceLaunch* info);
float half_image(float v);   // (float)(_Float16)v, on the host: the value a source's fp16 copy carries
// posh[first..first+count) = half(posm[...]) (round to nearest even)
hipError_t launch_pack_half(const float4* posm, void* posh, int first, int count, hipStream_t stream);

// Sources whose mass differs from the common mass of a unit-mass sweep (variant 7): after the sweep gave every source the
// common mass, such a body still owes (m - m_common) * d / (|d|^2 + eps) to every target.  rec[k] = (x, y, z, m - m_common)
// of exceptional body k, SNAPSHOT by the sweep kernel itself (K2 moves bodies in place while it reads these).  count = 0: none.
struct MassExceptions {
    const float4* rec;
    const int* idx;      // body index of every record (a target skips its own)
    int count;
    int dim;
};

// K4 on the wave-split kernels: the sources were a widened fp16 copy (src), so every target's sum holds its interaction with
// its OWN image; K2 / the force readout take it out. unit_mass > 0: the sweep weighted every source with it; else with src.w.
struct SelfImage {
    const float4* src;
    float unit_mass;
    int dim;
};

// K2: reduce partials in fixed order (- the self image, + the exceptional sources), kick-drift, write positions in place.
hipError_t launch_integrate(float4* posm, int lo, int n_targets, float4* vel, const float4* acc_partial,
                            int jsplit, int acc_stride, float dt, hipStream_t stream,
                            MassExceptions exc = MassExceptions{nullptr, nullptr, 0, 3},
                            SelfImage si = SelfImage{nullptr, 0.0f, 3});

// forces-only readout: F_i = m_i * a_i into float4 out[n_targets]
hipError_t launch_reduce_forces(const float4* posm, int lo, int n_targets, const float4* acc_partial, int jsplit,
                                int acc_stride, float4* out, hipStream_t stream,
                                MassExceptions exc = MassExceptions{nullptr, nullptr, 0, 3},
                                SelfImage si = SelfImage{nullptr, 0.0f, 3});

// strict (bit-exact) pair: ascending j per target, IEEE divide, no contraction. 2-D.
// kernel: 16 or 8 = workgroups of that many waves per 64 targets (term producers + one summing wave), 1 = one thread per body,
// anything else = by size (strict_kernel_choice); info->variant = -(kernel). Results are bit-identical whichever runs.
// guard (optional device word): when given, max|coordinate| of the sources is reduced into it first and the kernel takes
// the short correctly-rounded division whenever that maximum is <= 1e5 -- the caller passes it only if
// strict_fastdiv_ok(min mass, max mass); null = always the compiler's IEEE division. Results are identical either way.
hipError_t launch_force_strict(const float4* posm, int n, int lo, int n_targets, float2* force_out,
                               hipStream_t stream, ForceLaunch* info = nullptr, unsigned* guard = nullptr, int kernel = 0);
int strict_kernel_choice(int n_targets);
bool strict_fastdiv_ok(float mass_min, float mass_max);
// *guard = float bits of max(|x|, |y|, |z|) over posm[0..n_records) (NaN counts as +inf)
hipError_t launch_max_coord(const float4* posm, int n_records, unsigned* guard, hipStream_t stream);
// kick-drift from a per-body force (v += (dt*F)/m, nbody.rs:155) or acceleration (is_accel: v += dt*a),
// optional velocity kill box (nbody.rs:466-471).
// gate_* (optional; see bh_eval.hip BuildGate): the kernel runs only if the device tree build whose counters these are
// produced a usable tree -- for steps enqueued before the host has read the build's outcome
hipError_t launch_integrate_f2(float4* posm, int lo, int n_targets, float4* vel, const float2* force, float dt,
                               int is_accel, int killbox, hipStream_t stream, int* gate_counters = nullptr,
                               int gate_node_cap = 0, int gate_crowd_limit = 0, int gate_queue_limit = 0, int* gate_host_out = nullptr);

// K3: Barnes-Hut traversal. mode 0 = fast (sequential pre-order accumulation, rcp),
// mode 1 = strict (hierarchical summation order of nbody.rs:354-360 via an explicit frame stack,
// IEEE sqrt/divide): bit-exact with the reference traversal.  mode 2 / 3 = the wave-uniform forms of 0 / 1
// (need perm: bodies in a spatial order); same results as 0 / 1.
// perm (optional): thread t evaluates body perm[t] -- a GLOBAL body index inside the slab
// [lo, lo + n_targets) -- instead of body lo + t (spatial order => coherent waves); force_out is indexed by body - lo
// gate_* (fast walks only): as launch_integrate_f2; n_nodes is then read from gate_counters[0] on the device
hipError_t launch_bh_eval(const float4* posm, int lo, int n_targets, const BhNode* nodes, int n_nodes, float theta,
                          int mode, float2* force_out, hipStream_t stream, const unsigned* perm = nullptr,
                          int* gate_counters = nullptr, int gate_node_cap = 0, int gate_crowd_limit = 0,
                          int gate_queue_limit = 0);

// planar (x, y) of posm[0..n) into device-visible pinned host arrays (input of the host quadtree build)
hipError_t launch_split_xy(const float4* posm, int n, float* xs_host_pinned, float* ys_host_pinned, hipStream_t stream);

hipError_t launch_bh_count(const float4* posm, int lo, int n_targets, const BhNode* nodes, int n_nodes, float theta,
                           unsigned long long* totals, hipStream_t stream);

// Quadtree build on the device (bh_build.hip): same node set as the host build, flattened straight into `out`.
size_t device_tree_workspace_bytes(int n, int node_cap, size_t* sort_tmp_bytes);
// call once after every (re)allocation of the workspace, before its first use (clears the header's self-clearing ticket)
hipError_t device_tree_workspace_init(void* workspace, hipStream_t stream);
// warm (round 5): the workspace still holds the order an earlier call (this one or device_tree_build_begin) left for the SAME n
// bodies, give or take a step's motion -- the sort then starts from it (bh_sort.hip: k_sample_rank / k_keys_scatter / k_bucket_sort) instead
// of from scratch; a warm sort whose buckets overflow refuses the build (counters[1], see bh_sort.hip)
// sorted_pos (optional, warm only): posm in the order the workspace holds (entry t = posm[order[t]], as the last fused kick-drift
// left it, BhKick::sorted): the warm sort then reads the bodies from there, coalesced, instead of gathering them
hipError_t device_spatial_order(const float4* posm, int n, void* workspace, size_t workspace_bytes, const unsigned** perm_dev,
                                hipStream_t stream, bool warm = false, const float4* sorted_pos = nullptr);
// Routing + stable scatter for the host quadtree build (see bh_front.hip): top_host = ntop records of (x1, y1, x2, y2,
// first_child, bucket); pbucket_host / events_host / offset_host are pinned, device-visible host arrays of rest ints,
// rest 16-byte insert events and nb + 1 64-bit offsets.  Enqueues on `stream`; the caller waits.
size_t device_route_workspace_bytes(int rest, int ntop, int nb);
hipError_t device_route_and_scatter(const float4* posm, int warm, int rest, const void* top_host, int ntop,
                                    const int* bucket_depth_host, int nb, void* workspace, size_t workspace_bytes,
                                    int* pbucket_host, void* events_host, unsigned long long* offset_host, hipStream_t stream);
// perm restricted to the bodies of one slab [lo, hi), order kept (global body indices); *slab_perm points into workspace
size_t device_slab_order_workspace_bytes(int n);
hipError_t device_slab_order(const unsigned* perm, int n, int lo, int hi, void* workspace, size_t workspace_bytes,
                             const unsigned** slab_perm, hipStream_t stream);
// fold: 0 = interior masses / centres are roundings of exact fp64 sums (own tolerance class); 1 = the reference's f32 running
// fold in arrival order (nbody.rs:303-320), EPS clusters of any size replayed in arrival order: the host tree bit for bit; what
// the replay cannot reproduce reports status 2 (caller builds on the host; the reasons are in the build's counter word 5).
constexpr int kFoldFaithfulMax = 65536;   // bit-exact mode (only this class serves it): faithful fold up to this many bodies (the root's chain is n serial steps)
// Fast mode, default class BY COST (round 6): the reference fold only while its build takes at most 1.5 x the
// exact-sum build's.  Measured (profiles/r06_bh_sizes.jsonl, build ms reference / exact): 2 000 bodies 0.114 / 0.044, 10 000
// 0.19 / 0.057, 65 536 1.09 / 0.086 -- 2.6 x at the smallest size listed and growing (the root's fold is n serial f32 steps), and
// below 1 024 bodies this class is served by the host build anyway: at NO size the device build serves.  0 = never by default;
// NBX_OPT_BH_FOLD = 1 still asks for it at any size.
constexpr int kFoldCostMax = 0;
// side / ev_go / ev_done (optional, fold = 1): a second stream and two events of the same device -- the root's fold then runs
// on `side` from the start of the build, beside everything else (it is the longest chain and needs only the bodies);
// host_counters: pinned words the build's counters are copied to at the end; null = the caller's gated kick-drift
// (launch_integrate_f2 gate_host_out) hands them over instead, no copy command
hipError_t device_tree_build_begin(const float4* posm, int n, void* workspace, size_t workspace_bytes, int node_cap, BhNode* out,
                                   int* host_counters, const unsigned** perm_dev, hipStream_t stream, int fold = 0,
                                   hipStream_t side = nullptr, hipEvent_t ev_go = nullptr, hipEvent_t ev_done = nullptr,
                                   bool depth_panic_guard = false, bool warm = false, const float4* sorted_pos = nullptr);
hipError_t device_tree_build_end(int n, int node_cap, const int* host_counters, int* n_nodes_host, int* status,
                                 hipStream_t stream, int fold = 0);
// the device-side view of the same verdict: where the build's counters live (for launch_bh_eval / launch_integrate_f2 gates)
// and the limits device_tree_build_end applies to counters[1] (left-behind bodies) and counters[2] (queued folds)
int* device_tree_counters(void* workspace);
constexpr int kWhySortOverflow = 1 << 20;   // refusal reason (counter word 5): a bucket of the warm sort outgrew its slots -- the ORDER failed, not the tree
constexpr int kTreePoisonWord = 9;   // device_tree_counters()[9]: set by a gated kick-drift whose build was refused; while it is set
                                     // every gated kernel is a no-op (the host clears it when it redoes the refused step)
void device_tree_limits(int n, int fold, int* crowd_limit, int* queue_limit);

// nb_draw on the device: counts (uint2 per pixel: body hits, tail hits) -> ABGR framebuffer. Particles whose tail octant
// cannot be decided safely on the device (see draw.hip) are appended to amb[0..*amb_count) (capacity n) with their body
// pixel and velocity; their body hit is counted, their tail is the host's to add.
struct DrawAmbiguous {
    int32_t xi, yi;
    float vx, vy;
};
hipError_t launch_draw(const float4* posm, const float4* vel, int n, int w, int h, float x1, float y1, float scalex,
                       float scaley, void* counts, unsigned* fb, unsigned* amb_count, DrawAmbiguous* amb, hipStream_t stream);

}  // namespace nbx
