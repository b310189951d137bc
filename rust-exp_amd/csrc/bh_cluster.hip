// bh_cluster.hip -- the reference's EPS merge in full (nbody.rs:249-260), replayed on the device in arrival order: the part of
// the device tree build (bh_build.hip) that lets the reference-fold class promise the reference's tree node for node.
#include "bh_build_internal.h"

namespace nbx {

// ---- 3c. the reference's EPS merge in full (reference fold: that class promises the reference's tree node for node) ---------
//
// Sequential insertion (nbody.rs:226-284) decides a body B's fate when it ARRIVES: among the entities in the tree at that moment
// -- single bodies and blobs of merged bodies -- B walks down to the leaf of the one entity A that shares the most leading path
// digits with it (a tie, or none: B opens a leaf of its own); if A's current centre is closer than EPS in both axes B is folded
// into A (add_mass), else the leaf splits and B becomes an entity.  A blob's centre moves with every member, later arrivals are
// tested against the moved centre, and the blob travels down by its centre whenever its leaf splits (nbody.rs:271-281).
// All of this involves only bodies within 2 EPS of one another: B within EPS of a centre is within 2 EPS of one of the members.
// So:
//   * k_cells   a hash table of the occupied cells of a grid (the quadtree level whose cells are >= 2.5 EPS wide; a cell is a
//               contiguous range of the sorted keys) -> "who is within 2 EPS of this point" is nine probes, not a search;
//   * k_blobs   every entity-by-key (a run of identical keys; usually one body) looks around; one with company that arrived
//               before all of its neighbours collects its connected component (chains of < 2 EPS links; usually 2-5 bodies) and,
//               if it is the component's first arrival, REPLAYS the component's arrivals in index order, by the rule above:
//               entities in LDS, exact f32 folds, the nearest-entity rule from the keys, outside bodies that arrived earlier and
//               share the cell taken into account (k_merge_links' rival scan).  A blob's path is its centre's; its members take
//               the path of its last centre as their key ("ghosts" when that is not their own: listed for k_place);
//   * k_place   the bodies in the order of their ENTITY keys (a ghost moves next to its entity: usually by a slot or two, but by
//               any distance when a coarse cell boundary runs between the two) -- keys, indices, records, out of place.
// The tree files a blob under the path of its LAST centre.  In the reference the path is made of stretches, each laid down by the
// centre the blob had while its leaf went from one depth to the next (the first by the opener's own position); the last centre's
// path is that path down to the blob's final leaf iff every centre the blob ever had shares it that far, and the opener's
// position shares it as far as the opener's leaf went before it took in its first body (at most the digits those two share).
// What that, or the replay, cannot reproduce soundly is not guessed: the step then goes to the host build (counted in
// counters[1], the reasons in counters[5]):
//   * a blob whose successive centres do not share one path down to its final leaf (k_emit compares pmin with the leaf depth;
//     blobs of three or more bodies with an unmerged body a fraction of EPS away, mostly).  Telling WHICH centre laid down which
//     stretch was built too: every entity's depth over time then hinges on its nearest earlier-arrived neighbours in key order,
//     and in the dense cores where blobs form those are members of OTHER components more often than not (the 10 000-body
//     nb_random_disk: 695 of 1 000 steps refused, against none like this);
//   * a body of ANOTHER component among the rivals of a merge (its entity may sit elsewhere),
//   * any body outside the component within EPS of any centre a blob ever had (the 2 EPS argument holds for exact arithmetic;
//     this checks the computed centres),
//   * two entities in one level-31 cell that do not merge (the reference goes deeper than the keys do),
//   * components of more than kBlobRuns entities / kBlobBodies bodies, more than kGhostCap ghosts, crowded neighbourhoods.
struct CellGrid {
    int D, sh;                         // D digits of a key name a grid cell; key >> sh = the cell's prefix
    const unsigned long long* hk;      // open addressing: prefix + 1 (0 = free) ...
    const int* hv;                     // ... -> the first sorted slot of the cell
    unsigned mask;
};

__device__ __forceinline__ CellGrid make_grid(const unsigned* __restrict__ box, const unsigned long long* hk, const int* hv,
                                              const unsigned mask)
{
    // cells at least 2.5 EPS wide in both axes (widths halve per level; the f32 midpoints move them by rounding only)
    float wx = dec_f32(box[2]) - dec_f32(box[0]), wy = dec_f32(box[3]) - dec_f32(box[1]);
    int D = 0;
    while (D < kLevels && wx * 0.5f >= 2.5f * kEps && wy * 0.5f >= 2.5f * kEps) { wx *= 0.5f; wy *= 0.5f; D++; }
    return CellGrid{D, 2 * (kLevels - D), hk, hv, mask};
}

__device__ __forceinline__ unsigned hash_cell(unsigned long long c)
{
    c ^= c >> 33; c *= 0xff51afd7ed558ccdull;
    c ^= c >> 33; c *= 0xc4ceb9fe1a85ec53ull;
    c ^= c >> 33;
    return (unsigned)c;
}

// Gathers the bodies into sorted order (sb[j] = posm[idx[j]]), starts every body as its own entity, and enters the first body
// of every grid cell into the table.
__global__ __launch_bounds__(kTile) void k_cells(const float4* __restrict__ posm, float4* __restrict__ sb,
                                                 const unsigned long long* __restrict__ keys, const unsigned* __restrict__ idx,
                                                 const unsigned* __restrict__ box, const int n, unsigned long long* __restrict__ hk,
                                                 int* __restrict__ hv, const unsigned mask, unsigned long long* __restrict__ ekey,
                                                 unsigned char* __restrict__ pmin, const int sb_ready, const int* __restrict__ poison)
{
    if (*poison) return;   // (see launch_cluster_replay)
    const int j = blockIdx.x * kTile + threadIdx.x;
    if (j >= n) return;
    if (!sb_ready) sb[j] = posm[idx[j]];      // (the warm sort delivers the records itself, bh_sort.hip)
    const unsigned long long k = keys[j];
    ekey[j] = k;
    pmin[j] = (unsigned char)kLevels;
    const CellGrid g = make_grid(box, hk, hv, mask);
    const unsigned long long prefix = k >> g.sh;
    if (j > 0 && (keys[j - 1] >> g.sh) == prefix) return;
    unsigned h = hash_cell(prefix) & mask;
    for (;;) {
        const unsigned long long old = atomicCAS(&hk[h], 0ull, prefix + 1ull);   // (a prefix is entered once: by its first slot)
        if (old == 0ull) { hv[h] = j; return; }
        h = (h + 1u) & mask;
    }
}

// every second bit of a word: bit b of v -> bit 2b (and back)
__device__ __forceinline__ unsigned long long spread_bits(const unsigned v)
{
    unsigned long long x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
__device__ __forceinline__ unsigned compact_bits(unsigned long long x)
{
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return (unsigned)x;
}

// f(first slot of an entity-by-key) for every one in the 3 x 3 block of grid cells around the cell of `at` (a path key), until f
// returns false.  0: all visited; 1: stopped by f; 2: more than kCloseScanCap of them (a collinear or collapsed system).
// own >= 0: `at` is the key of the body in sorted slot `own` -- its own cell is then found by walking left from that slot (the
// neighbouring keys are in the cache of the wave's coalesced loads) instead of through the table.  The table probes of the other
// cells are issued together, then the hits' slots, before anything is looked at: one round trip each instead of nine in a row
// (nearly every probe finds an empty cell).
template <class F>
__device__ __forceinline__ int visit_entities_near(const CellGrid& g, const unsigned long long at,
                                                   const unsigned long long* __restrict__ keys, const int n, const int own, F&& f)
{
    const unsigned long long centre = at >> g.sh;     // digit = (lower << 1) | right, one per level
    const unsigned ix = compact_bits(centre), iy = compact_bits(centre >> 1);
    const long long lim = 1ll << g.D;
    unsigned long long prefix[9], found[9];
    unsigned h[9];
#pragma unroll
    for (int c9 = 0; c9 < 9; c9++) {
        const long long cx = (long long)ix + (c9 % 3 - 1), cy = (long long)iy + (c9 / 3 - 1);
        const bool inside = !(cx < 0 || cy < 0 || cx >= lim || cy >= lim);
        const unsigned long long p = (spread_bits((unsigned)cy) << 1) | spread_bits((unsigned)cx);
        prefix[c9] = p;
        h[c9] = hash_cell(p) & g.mask;
        found[c9] = !inside ? 0ull : (c9 == 4 && own >= 0) ? p + 1ull : g.hk[h[c9]];
    }
    int start[9];
#pragma unroll
    for (int c9 = 0; c9 < 9; c9++) start[c9] = (found[c9] == prefix[c9] + 1ull && !(c9 == 4 && own >= 0)) ? g.hv[h[c9]] : -1;
    if (own >= 0) {
        int t = own, steps = 0;
        while (t > 0 && (keys[t - 1] >> g.sh) == centre && ++steps <= 64) t--;
        start[4] = steps > 64 ? g.hv[h[4]] : t;       // (a crowded cell: the table knows where it starts -- it holds every cell)
        if (steps > 64) found[4] = g.hk[h[4]];
    }
    int seen = 0;
#pragma unroll 1
    for (int c9 = 0; c9 < 9; c9++) {
        if (found[c9] == 0ull) continue;              // outside the grid, or nobody there
        int t = start[c9];
        if (found[c9] != prefix[c9] + 1ull) {         // the slot held another cell: probe on
            unsigned hh = h[c9];
            unsigned long long kk = found[c9];
            while (kk != 0ull && kk != prefix[c9] + 1ull) { hh = (hh + 1u) & g.mask; kk = g.hk[hh]; }
            if (kk == 0ull) continue;
            t = g.hv[hh];
        }
        const unsigned long long pc = prefix[c9];
        while (t < n && (keys[t] >> g.sh) == pc) {
            if (++seen > kCloseScanCap) return 2;
            if (!f(t)) return 1;
            t = run_end(keys, t, n);
        }
    }
    return 0;
}

__device__ __forceinline__ bool within(const float4 a, const float4 b, const float r)
{
    return fabsf(__fsub_rn(a.x, b.x)) < r && fabsf(__fsub_rn(a.y, b.y)) < r;
}

struct BlobShared {
    int run_first[kBlobRuns], run_last[kBlobRuns];   // the component: entities by key, as ranges of sorted slots
    int mem_slot[kBlobBodies];                       // its bodies in arrival order ...
    unsigned mem_idx[kBlobBodies];
    unsigned char mem_ent[kBlobBodies];              // ... and the entity each of them ended in
    unsigned char ent_pmin[kBlobBodies];             // entities of the replay: fewest digits two successive centres' paths shared
    unsigned char ent_c1[kBlobBodies];               //   digits the opener shared with the first body it took in (kLevels + 1: none yet)
    unsigned long long ent_key[kBlobBodies];         //   path: the opener's key, then the path of the current centre
    float ent_x[kBlobBodies], ent_y[kBlobBodies], ent_m[kBlobBodies];
    int ent_first[kBlobBodies];                      //   the opener's sorted slot
};

__device__ __forceinline__ bool in_component(const BlobShared& s, const int nruns, const int slot)
{
    for (int u = 0; u < nruns; u++)
        if (slot >= s.run_first[u] && slot < s.run_last[u]) return true;
    return false;
}

struct ReplayView {
    const CellGrid& g;
    const float4* __restrict__ sb;
    const unsigned long long* __restrict__ keys;
    const unsigned* __restrict__ idx;
    const unsigned* __restrict__ box;
    int n;
};

// Is a body that does not belong to the component, arrived before body (slot, ib) and shares at least c digits with it in the
// tree when that body arrives?  0 no, 1 yes (the body then never reaches the component's entity), 2 cannot tell -> host build
__device__ __forceinline__ int outside_rival(const BlobShared& s, const int nruns, const ReplayView& v, const int slot,
                                             const unsigned long long kb, const unsigned ib, const int c)
{
    int steps = 0;
    for (int dir = -1; dir <= 1; dir += 2) {
        for (int x = slot + dir; x >= 0 && x < v.n; x += dir) {
            if (common_digits(kb, v.keys[x]) < c) break;
            if (++steps > kRivalScanCap) return 2;
            if (v.idx[x] >= ib || in_component(s, nruns, x)) continue;   // arrives later / the replay knows it
            // an outsider that was there first.  It is an entity under its own key unless it belongs to a component of its own
            // (then its entity may carry another member's key): anybody within 2 EPS of it?
            const float4 px = v.sb[x];
            const unsigned long long kx = v.keys[x];
            bool company = false;
            const int st = visit_entities_near(v.g, kx, v.keys, v.n, x, [&](const int t) {
                if (v.keys[t] == kx) return true;
                if (within(px, v.sb[t], 2.0f * kEps)) { company = true; return false; }
                return true;
            });
            return (st == 2 || company) ? 2 : 1;
        }
    }
    return 0;
}

// The component's arrivals replayed in index order by ONE lane.  0, or why the host build has to do this step.
__device__ int replay_component(BlobShared& s, const int nruns, const ReplayView& v, unsigned long long* __restrict__ ekey,
                                unsigned char* __restrict__ pmin, int* __restrict__ ghosts, int* __restrict__ counters)
{
    int k = 0;
    for (int r = 0; r < nruns; r++)
        for (int slot = s.run_first[r]; slot < s.run_last[r]; slot++) {   // (the caller made sure they fit)
            const unsigned a = v.idx[slot];
            int pos = k++;
            while (pos > 0 && s.mem_idx[pos - 1] > a) { s.mem_idx[pos] = s.mem_idx[pos - 1]; s.mem_slot[pos] = s.mem_slot[pos - 1]; pos--; }
            s.mem_idx[pos] = a;
            s.mem_slot[pos] = slot;
        }
    int ne = 0;
    for (int t = 0; t < k; t++) {
        // somebody has refused this build already (this class tolerates nothing: device_tree_limits): whatever is replayed from here
        // on is thrown away.  In a collapsed core -- where refusals come from -- the replays left are the expensive ones (rival scans
        // of a thousand slots, 3 x 3 cell visits of hundreds of entities, per member): round 6 measured 0.3-0.7 s for one refused
        // build of a collapsing 65 536-body disc (profiles/r06_bh_sizes.jsonl, first run) against ~1 ms for a kept one
        if (__hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return 0;
        const int slot = s.mem_slot[t];
        const unsigned long long kb = v.keys[slot];
        const float4 pb = v.sb[slot];
        int best = -1, cbest = -1;
        bool tie = false;
        for (int e = 0; e < ne; e++) {
            const int c = common_digits(kb, s.ent_key[e]);
            if (c > cbest) { cbest = c; best = e; tie = false; }
            else if (c == cbest) tie = true;
        }
        bool fresh = best < 0 || tie;   // no entity of the component yet / two equally near: a leaf of its own (nbody.rs:234-240)
        if (!fresh) {
            const int st = outside_rival(s, nruns, v, slot, kb, s.mem_idx[t], cbest);
            if (st == 2) return kWhyRival;
            fresh = st == 1;
        }
        if (!fresh) {
            // arrives at entity `best`'s leaf (nbody.rs:249-260)
            if (fabsf(__fsub_rn(s.ent_x[best], pb.x)) < kEps && fabsf(__fsub_rn(s.ent_y[best], pb.y)) < kEps) {
                float x = s.ent_x[best], y = s.ent_y[best], m = s.ent_m[best];
                fold_mass(x, y, m, pb.x, pb.y, pb.w);
                s.ent_x[best] = x; s.ent_y[best] = y; s.ent_m[best] = m;
                s.mem_ent[t] = (unsigned char)best;
                // the blob's path from here on is its centre's (nbody.rs:271-281: a split re-inserts it by its position)
                const unsigned long long kc = path_key(v.box, x, y);
                if ((int)s.ent_c1[best] > kLevels) s.ent_c1[best] = (unsigned char)cbest;   // the leaf was at most this deep
                else {
                    const int c = common_digits(kc, s.ent_key[best]);
                    if (c < (int)s.ent_pmin[best]) s.ent_pmin[best] = (unsigned char)c;
                }
                s.ent_key[best] = kc;
                // nobody outside the component may ever be within EPS of this centre
                const float4 centre = make_float4(x, y, 0.0f, 0.0f);
                const int st = visit_entities_near(v.g, kc, v.keys, v.n, -1, [&](const int u) {
                    return !within(centre, v.sb[u], kEps) || in_component(s, nruns, u);
                });
                if (st != 0) return kWhyOutsider;
                continue;
            }
            if (cbest >= kLevels) return kWhyLevel31;   // the same level-31 cell and not close: the reference splits deeper than the keys go
        }
        s.ent_key[ne] = kb;
        s.ent_x[ne] = pb.x; s.ent_y[ne] = pb.y; s.ent_m[ne] = pb.w;
        s.ent_first[ne] = slot;
        s.ent_pmin[ne] = (unsigned char)kLevels;
        s.ent_c1[ne] = (unsigned char)(kLevels + 1);
        s.mem_ent[t] = (unsigned char)ne;
        ne++;
    }
    // A blob is filed under the path of its LAST centre.  That is its path in the reference's tree down to its final leaf iff
    // every centre it ever had shares that path that far (each stretch of the path was laid down by the centre of its time; the
    // fewest digits two successive centres share is the fewest any shares with the last) and the opener's own position shares
    // it down to the depth its leaf had when it took in its first body -- at most the digits the two shared.  k_emit knows the
    // final leaf depth and compares (pmin); two entities that end on one 62-bit path would need a deeper tree than the keys hold.
    // (Bounding every centre's stretch like the opener's -- it ends above the digits the NEXT body shared with the path -- was
    //  tried: the same 167 of 300 steps of the collapsing 65 536-body disc refused, 3 of 600 fuzz cases more kept.  Not kept.)
    for (int e = 0; e < ne; e++) {
        if ((int)s.ent_c1[e] > kLevels) continue;            // never took anybody in: its own key, nothing to check
        const int ca = common_digits(v.keys[s.ent_first[e]], s.ent_key[e]);
        if (ca < (int)s.ent_c1[e] && ca < (int)s.ent_pmin[e]) s.ent_pmin[e] = (unsigned char)ca;
        for (int o = 0; o < ne; o++)
            if (o != e && s.ent_key[o] == s.ent_key[e]) return kWhyLevel31;
    }
    int why = 0;
    for (int t = 0; t < k; t++) {
        const int slot = s.mem_slot[t];
        const int e = s.mem_ent[t];
        pmin[slot] = s.ent_pmin[e];
        const unsigned long long ke = s.ent_key[e];
        if (ke == v.keys[slot]) continue;
        ekey[slot] = ke;                                     // a ghost: filed under its entity's path
        const int gi = atomicAdd(&counters[4], 1);
        if (gi < kGhostCap) ghosts[gi] = slot; else why = kWhyGhosts;
    }
    return why;
}

__global__ __launch_bounds__(kTile) void k_blobs(const float4* __restrict__ sb, const unsigned long long* __restrict__ keys,
                                                 const unsigned* __restrict__ idx, const unsigned* __restrict__ box, const int n,
                                                 const unsigned long long* __restrict__ hk, const int* __restrict__ hv,
                                                 const unsigned mask, unsigned long long* __restrict__ ekey,
                                                 unsigned char* __restrict__ pmin, int* __restrict__ ghosts, int* __restrict__ counters)
{
    __shared__ BlobShared bs[kTile / 64];             // one component at a time per wave
    if (counters[kTreePoisonWord]) return;            // (see launch_cluster_replay)
    const int j = blockIdx.x * kTile + threadIdx.x;
    const CellGrid g = make_grid(box, hk, hv, mask);
    bool root = false;
    unsigned mine = 0;
    if (j < n && !(j > 0 && keys[j - 1] == keys[j])) {   // the first body of an entity-by-key speaks for it
        const unsigned long long kj = keys[j];
        const float4 p = sb[j];
        mine = idx[j];
        bool company = false, later = true;
        const int st = visit_entities_near(g, kj, keys, n, j, [&](const int t) {
            if (keys[t] == kj || !within(p, sb[t], 2.0f * kEps)) return true;
            company = true;
            if (idx[t] < mine) later = false;
            return true;
        });
        if (st == 2) refuse(counters, kWhyCrowdedScan);
        // alone, or a neighbour arrived first (the component's first arrival replays it): nothing to do
        root = st != 2 && company && later;
        const int last = run_end(keys, j, n);
        if (st != 2 && !company && last - j > 1) {
            // Several bodies of one level-31 cell and nobody else around: one leaf -- as long as every arrival is within EPS of the
            // centre the earlier ones have folded to.  Where an ulp of the coordinates is no longer small against EPS (|x| in the
            // thousands) the folded centre of even IDENTICAL positions can sit more than EPS away (nbody.rs:315-317 round three
            // times): the reference then splits, 31 levels are not enough, and the host build has to do it.
            float cx = 0.0f, cy = 0.0f, cm = 0.0f;
            bool one_leaf = true;
            for (int t = j; t < last && one_leaf; t++) {          // (the stable sort left them in index order)
                const float4 q = sb[t];
                if (t > j && !(fabsf(__fsub_rn(cx, q.x)) < kEps && fabsf(__fsub_rn(cy, q.y)) < kEps)) one_leaf = false;
                fold_mass(cx, cy, cm, q.x, q.y, q.w);
            }
            if (!one_leaf) refuse(counters, kWhyLevel31);
        }
    }
    // the wave's candidates one after the other (they share the wave's LDS record; a lane cannot wait for another lane)
    BlobShared& s = bs[threadIdx.x >> 6];
    unsigned long long todo = __ballot(root);
    while (todo) {
        if (__hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;   // refused already (see replay_component)
        const int lane = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        if ((int)(threadIdx.x & 63) != lane) continue;
        int nruns = 1;
        s.run_first[0] = j;
        s.run_last[0] = run_end(keys, j, n);
        int bodies = s.run_last[0] - j;
        bool first = true, fits = bodies <= kBlobBodies;
        for (int r = 0; r < nruns && first && fits; r++) {
            const int fr = s.run_first[r];
            const float4 pr = sb[fr];
            const unsigned long long kr = keys[fr];
            const int st = visit_entities_near(g, kr, keys, n, fr, [&](const int t) {
                if (keys[t] == kr || !within(pr, sb[t], 2.0f * kEps)) return true;
                for (int u = 0; u < nruns; u++)
                    if (s.run_first[u] == t) return true;
                if (idx[t] < mine) { first = false; return false; }
                const int e = run_end(keys, t, n);
                bodies += e - t;
                if (nruns == kBlobRuns || bodies > kBlobBodies) { fits = false; return false; }
                s.run_first[nruns] = t; s.run_last[nruns] = e;
                nruns++;
                return true;
            });
            if (st == 2) fits = false;
        }
        if (first) {
            const ReplayView v{g, sb, keys, idx, box, n};
            const int why = fits ? replay_component(s, nruns, v, ekey, pmin, ghosts, counters) : kWhyBigComponent;
            if (why) refuse(counters, why);
        }
    }
}

// The bodies in the order of their entity keys.  Everybody but the ghosts keeps its relative order (their keys are sorted); a
// ghost goes behind the bodies that carry its entity's key themselves.  Keys, indices, records and pmin, out of place.
__global__ __launch_bounds__(kTile) void k_place(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ ekey,
                                                 const unsigned* __restrict__ idx, const float4* __restrict__ sb,
                                                 const unsigned char* __restrict__ pmin, const int* __restrict__ ghosts,
                                                 const int* __restrict__ counters, const int n, unsigned long long* __restrict__ keys2,
                                                 unsigned* __restrict__ idx2, float4* __restrict__ sb2, unsigned char* __restrict__ pmin2)
{
    __shared__ int gslot[kGhostCap];
    __shared__ unsigned long long gkey[kGhostCap];
    if (counters[kTreePoisonWord]) return;            // (see launch_cluster_replay)
    int G = counters[4];
    const bool overflow = G > kGhostCap;              // the step is refused then: everybody stays where it is, under its own
    if (overflow) G = 0;                              // key (the arrays below must hold a permutation whatever happens)
    for (int t = threadIdx.x; t < G; t += kTile) {
        const int sl = ghosts[t];
        gslot[t] = sl;
        gkey[t] = ekey[sl];
    }
    __syncthreads();
    const int j = blockIdx.x * kTile + threadIdx.x;
    if (j >= n) return;
    const unsigned long long own = keys[j], ek = overflow ? own : ekey[j];
    int pos = j;
    if (G > 0) {
        if (ek == own) {
            int before = 0, ahead = 0;
            for (int t = 0; t < G; t++) { before += gslot[t] < j ? 1 : 0; ahead += gkey[t] < own ? 1 : 0; }
            pos = j - before + ahead;
        } else {
            int lo = 0, hi = n;                       // first slot whose key is above the entity's
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (keys[mid] <= ek) lo = mid + 1; else hi = mid;
            }
            int before = 0, ahead = 0;
            for (int t = 0; t < G; t++) {
                before += gslot[t] < lo ? 1 : 0;
                ahead += (gkey[t] < ek || (gkey[t] == ek && gslot[t] < j)) ? 1 : 0;
            }
            pos = lo - before + ahead;
        }
    }
    if (pos < 0 || pos >= n) return;
    keys2[pos] = ek;
    idx2[pos] = idx[j];
    sb2[pos] = sb[j];
    pmin2[pos] = pmin[j];
}


// The poison word (kernels.h kTreePoisonWord): a build enqueued BEHIND a refused step starts from what that step left -- the warm sort's
// kernels then do nothing (bh_sort.hip launch_inc_sort), and with them the clearing of the cell table; a replay on a table still full of the
// refused build's cells took 33-49 ms (k_cells) + 270-380 ms (k_blobs) at 65 536 bodies (every refusal of a run of pipelined steps:
// profiles/r06_bh_sizes.jsonl, 38.8 ms per step) for a tree nobody may read: the three kernels leave at once, like the sort's.
hipError_t launch_cluster_replay(const float4* posm, int n, const Workspace& k, hipStream_t stream, bool sb_ready)
{
    const int nb = (n + kTile - 1) / kTile;
    hipLaunchKernelGGL(k_cells, dim3(nb), dim3(kTile), 0, stream, posm, k.sb, k.keys1, k.idx1, k.box, n, k.hk, k.hv, k.hmask, k.ekey, k.link, sb_ready ? 1 : 0,
                       k.counters + kTreePoisonWord);
    hipLaunchKernelGGL(k_blobs, dim3(nb), dim3(kTile), 0, stream, k.sb, k.keys1, k.idx1, k.box, n, k.hk, k.hv, k.hmask, k.ekey, k.link,
                       k.ghosts, k.counters);
    hipLaunchKernelGGL(k_place, dim3(nb), dim3(kTile), 0, stream, k.keys1, k.ekey, k.idx1, k.sb, k.link, k.ghosts, k.counters, n,
                       k.keys0, k.idx0, k.sb2, k.pmin2);
    return hipGetLastError();
}

}  // namespace nbx
