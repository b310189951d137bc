/*
 * nbody_mi355x.h -- C ABI of libnbody_mi355x.so, the MI355X (gfx950) engine that replaces the
 * N-body hot path of blitzcode/rust-exp (rs-src/nbody.rs) behind that crate's own extern "C"
 * surface.  Plain C: fixed-width ints, floats, raw pointers and sizes; no C++/torch types.
 *
 * Two levels:
 *   Level 1 (nb_*)   the SIX symbols the reference exports and hs-src/RustNBodyExperiment.hs
 *                    imports (RustNBodyExperiment.hs:101-106).  Same names, argument meaning and
 *                    (absent) error reporting: they return void / a count; a fatal device error
 *                    prints a diagnostic and abort()s, the analogue of the reference's panic!.
 *   Level 2 (nbx_*)  additive, handle based, returns status codes.  This is what a Rust
 *                    `nbody.rs` shim or any other host binds: state
 *                    injection/extraction (the reference has none: its state is a private global,
 *                    nbody.rs:28-32, and its RNG is OS-seeded, nbody.rs:46,:90), force-only
 *                    evaluation, sharding across GPUs, profiling.
 *
 * There is NO CPU fallback: every step entry point needs a gfx950 device and fails loudly
 * (NBX_ERR_NO_DEVICE / abort) without one.  Host-side entry points (presets, set/get, draw, tree
 * build) work without a device.
 *
 * Citations `nbody.rs:A-B` are relative to /root/reference/rs-src/.
 */
#ifndef NBODY_MI355X_H
#define NBODY_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------- */
/* Level 1: drop-in replacements (process-global engine, internally serialised by a mutex like  */
/* the reference's `PARTICLES: Mutex<Vec<Particle>>`, nbody.rs:28-32).                          */
/* Environment knobs read once at first use: NB_DEVICE (ordinal, default 0), NB_GPUS (n | all:   */
/* single-process multi-GPU group, see nbx_group_*), NB_SEED (u64,                               */
/* default: OS entropy, as the reference's thread_rng), NB_FORCE_MODE=fast|strict,                */
/* NB_DRAW=host|device, NB_BH_TREE=host|device (both default: by size), NB_BH_FOLD=reference|exact  */
/* (NBX_OPT_BH_FOLD; default: by cost -- exact sums in the fast mode), NB_SOURCE_BITS=16 (fp16      */
/* source copy for the all-pairs sweep, BASELINE config #5; default 32).                          */
/* Both levels: NBX_HOST_THREADS (workers of the host quadtree build / flatten / draw; default   */
/* min(hardware threads, 32)), NBX_GROUP_EXCHANGE=copy (see nbx_group_*), NBX_LOG=1 (one stderr   */
/* line per step; a device tree build that hands its step to the host build says why),            */
/* NBX_TIMING=1 (host tree-build phases).                                                          */
/* NBX_SPIN_US (default 400: a host wait of the stepping path -- nbx_synchronize, the verdict of a */
/* pipelined Barnes-Hut step -- polls this many microseconds before it blocks; 0 = block at once;  */
/* systems above 32 768 bodies block at once). NBX_INC_SORT=0: the device tree build sorts from     */
/* scratch every step (the library sort) instead of from last step's order -- an A/B knob, the       */
/* results are bit-identical.  NBX_BH_BACKOFF_MAX: see NBX_STAT_BH_FALLBACKS.                        */

/* replaces nbody.rs:34-37   pub extern fn nb_num_particles() -> i32 */
int32_t nb_num_particles(void);

/* replaces nbody.rs:39-64   pub extern fn nb_random_disk(num_particles: i32) */
void nb_random_disk(int32_t num_particles);

/* replaces nbody.rs:73-104  pub extern fn nb_stable_orbits(num_particles: i32, rmin: f32, rmax: f32) */
void nb_stable_orbits(int32_t num_particles, float rmin, float rmax);

/* replaces nbody.rs:106-162 pub extern fn nb_step_brute_force(dt: f32)
 * O(N^2) pairwise force + kick-drift on the GPU; returns after the step completed (the Haskell
 * caller wall-clocks this call, RustNBodyExperiment.hs:55-57). */
void nb_step_brute_force(float dt);

/* replaces nbody.rs:186-480 pub extern fn nb_step_barnes_hut(theta: f32, dt: f32, nthreads: i32)
 * theta == 0.0 delegates to brute force (nbody.rs:197-200).  Otherwise: quadtree built on the
 * host exactly as the reference builds it, force evaluation + integration + velocity-kill on
 * the GPU.  `nthreads` (CPU worker count in the reference, nbody.rs:424-428) is accepted and
 * ignored by the GPU evaluation; nthreads <= 0 updates no particle, as in the reference (its workers are
 * `(0..nthreads).map(..)`: an empty iterator; the tree build and its asserts still run). */
void nb_step_barnes_hut(float theta, float dt, int32_t nthreads);

/* replaces nbody.rs:482-583 pub extern fn nb_draw(w: i32, h: i32, fb: *mut u32)
 * fb: caller-owned w*h little-endian ABGR words, cleared and filled here, not retained. */
void nb_draw(int32_t w, int32_t h, uint32_t *fb);

/* ------------------------------------------------------------------------------------------- */
/* Level 2: handle API                                                                          */

/* An engine is NOT internally synchronised: use one engine from one thread at a time (different engines may be
 * used from different threads; the level-1 symbols serialise themselves with a mutex like the reference). */
typedef struct nbx_engine nbx_engine;

enum nbx_status {
    NBX_OK = 0,
    NBX_ERR_INVALID = -1,    /* bad argument */
    NBX_ERR_NO_DEVICE = -2,  /* no usable gfx950 device: step entry points refuse (no CPU fallback) */
    NBX_ERR_HIP = -3,        /* HIP runtime error; text in nbx_last_error() */
    NBX_ERR_TREE_DEPTH = -4, /* quadtree depth > 50: the reference panics (nbody.rs:230-232) */
    NBX_ERR_TREE = -5,       /* other reference tree assert (nbody.rs:267,:293,:304) */
    NBX_ERR_ALLOC = -6,
    NBX_ERR_STATE = -7       /* call not valid in the engine's current configuration */
};

/* 14 options (numbers are stable; 10-12 and 17 moved to nbx_stat, 16 and 19 -- measured losers -- were removed in round 5) */
enum nbx_option {
    /* 0 = fast (default): a_i = sum_j m_j d/(|d|^2+eps) with v_rcp_f32 + FMA, tile order,
     *     parity within the stated fp32 tolerance.
     * 1 = strict: reference expression order, IEEE divide, ascending-j sequential sum, no FMA
     *     contraction: BIT-EXACT with the reference arithmetic (2-D only, no j-split). */
    NBX_OPT_FORCE_MODE = 0,
    NBX_OPT_JSPLIT = 1,            /* source-range split factor S; 0 = auto */
    NBX_OPT_BODIES_PER_THREAD = 2, /* variant 1: targets per thread, 2 or 4 (packed pairs); 0 = auto */
    NBX_OPT_DIM = 3,               /* 2 or 3; 0 = auto (2 when every z and vz is zero) */
    NBX_OPT_PROFILE = 4,           /* 1 = record a HIP event pair around every kernel launch */
    NBX_OPT_KERNEL_VARIANT = 5,    /* fast force kernel; -1 = auto (default: 7 / 6 for >= 16384 sources, else 1):
                                    *  7 = 6 for systems whose bodies all have the SAME mass: the per-interaction multiply
                                    *      by m_j leaves the loop (a = m * sum d/(r^2+eps)); falls back to 6 otherwise
                                    *  6 = packed fp32, sources through the scalar cache as SGPR operands (no LDS in the
                                    *      loop); the 4 waves of a workgroup share 256 targets, split the source range and
                                    *      reduce through LDS once
                                    *  1 = packed fp32, sources staged through LDS tiles
                                    * (0, 2, 3, 4, 5 of rounds 1-4 -- compiler-scheduled tiles, scalar-cache scalar math, 4-source
                                    *  batches, batched reciprocals, 6 without the wave split -- lost every A/B and were removed in
                                    *  round 5: docs/rounds/r01.md, r02.md hold their numbers) */
    NBX_OPT_DRAW_DEVICE = 7,       /* 1: nbx_draw/nb_draw splat on the GPU (count + resolve kernels, one w*h*4 B
                                    * download) instead of downloading the state. Pixel-identical to the host draw: the
                                    * few tails whose octant sits within 1e-5 of a step of the reference's f32 expression
                                    * (diagonal or near-diagonal velocities) are decided by the host's own atan2f.
                                    * Default (-1): host below 4096 bodies or while the state is not on the GPU, device otherwise */
    NBX_OPT_BH_TREE = 8,           /* Barnes-Hut tree: 0 = built on the host exactly like the reference (the bit-exact mode's
                                    * default), 1 = built on the device (bh_build.hip: same node set and leaf records incl. the
                                    * reference's EPS merge; interior centres of mass are roundings of the exact mean, or -- on
                                    * request, and in the bit-exact mode -- the reference's running f32 fold: NBX_OPT_BH_FOLD).  The bit-exact mode
                                    * honours 1 only while the device tree carries the reference fold (that tree IS the host
                                    * tree bit for bit, or the build refuses and the host builds: the same results at a third of
                                    * the step time at 10 000 bodies).  -1 (default) = device in the fast mode from 512 bodies on (1 024 with NBX_OPT_BH_FOLD = 1),
                                    * else host */
    NBX_OPT_BH_WAVE = 9,           /* 1 (default): with the device-built tree, walk the tree once per WAVE (node
                                    * records through the scalar cache, lanes park on accepted subtrees); 0: one
                                    * independent walk per lane. Bit-identical results either way */
    NBX_OPT_STRICT_KERNEL = 13,    /* bit-exact all-pairs kernel: 0 = by targets per GPU (default), 16 or 8 = workgroups of that
                                    * many waves per 64 targets (term producers + one summing wave), 1 = one thread per body.
                                    * Bit-identical results whichever runs */
    NBX_OPT_BH_FOLD = 14,          /* device-built tree, interior nodes: 1 = the reference's own f32 running fold of masses and
                                    * centres in ARRIVAL order (nbody.rs:303-320) and its EPS merge (nbody.rs:249-260: blobs of any
                                    * size, grown in arrival order) replayed on the device -- the host tree's records bit for bit;
                                    * what the replay cannot reproduce node for node is refused (NBX_LOG says why): the fast mode
                                    * then serves the step from the exact-sum DEVICE build (NBX_STAT_BH_CLASS_SWITCHES; round 6 --
                                    * the host build only if that refuses too), the bit-exact mode from the host build;
                                    * 0 = roundings of the exact sums (own tolerance class);
                                    * -1 (default) = BY COST (round 6): the fast mode takes 1 only while its build costs at most
                                    * 1.5 x the exact-sum build's -- at no size the device build serves (2.6 x at 2 000 bodies,
                                    * 12.7 x at 65 536: the root's fold is n serial f32 steps; profiles/r06_bh_sizes.jsonl), so the
                                    * fast mode's default is 0 at every size; the bit-exact mode (which only 1 can serve, and only
                                    * with NBX_OPT_BH_TREE = 1) keeps 1 up to 65 536 bodies.  Rounds 3-5 defaulted to 1 up to
                                    * 65 536 bodies in both modes: the reference's own 10 000-body scene then stepped in 0.22 ms
                                    * against 0.09, and nb_random_disk(65536) spent 63 % of its steps on the host tree */
    NBX_OPT_BH_ASYNC = 15,         /* 1 (default): a Barnes-Hut step on the device-built tree is enqueued without waiting for the
                                    * build's verdict (node count, EPS clusters); walk and kick-drift check it on the device, the
                                    * host at the next call that needs the state (nbx_synchronize, get, draw, the next step) and
                                    * redoes the step on the host tree if the build had to refuse. 0: wait inside the step */
    NBX_OPT_BH_WALK = 18,          /* fast-mode Barnes-Hut traversal: 1 (default) = over CHILD GROUPS (round 4, bh_walk.hip): the tree is
                                    * re-laid every step as one record per opened node -- its children's (x, y, m, T), T = the
                                    * reference's opening test s/sqrt(d^2) < theta turned into one exact threshold on d^2
                                    * (bh_threshold.h) -- so a walk loads once per OPENED node, decides with one compare per child
                                    * and keeps who-is-inside as a scalar lane mask; the loop is hand-scheduled gfx950 assembly.
                                    * 2 = the same walk as the compiler schedules it (bit-identical results; the A/B of DESIGN.md
                                    * K3).  0 = the node-by-node walk of rounds 1-3 (bh_eval.hip).  All make the reference's
                                    * decision for every body and node; 0 differs from 1 / 2 in the order the terms are added
                                    * (the fast mode's stated tolerance) */
    NBX_OPT_BH_FUSE_KICK = 20,     /* child-group walk, wave form, one GPU: 1 (default) = the walk kernel applies the step's kick-drift
                                    * itself as soon as a body's acceleration is complete (same operations, bit-identical state: a walk
                                    * reads no other body's position from the particle array -- the group records hold copies);
                                    * 0 = separate kick-drift kernel (the A/B: 0.0932 -> 0.0899 ms per step at 10 000 bodies,
                                    * 0.8375 -> 0.8252 at 1 048 576) */
    NBX_OPT_SOURCE_PRECISION = 6   /* 32 (default) or 16: all-pairs SOURCES read from a half4 (x,y,z,m) copy, 8 B/body;
                                    * targets, accumulators and the integrated state stay fp32 (fast mode only) */
};

/* What an engine has done so far (read only; nbx_get_stat).  Rounds 1-4 carried these among the options (10, 11, 12, 17). */
enum nbx_stat {
    NBX_STAT_BH_FALLBACKS = 0,     /* Barnes-Hut evaluations since the engine was created that the
                                    * device tree was selected for but the HOST tree served: builds the device refused (node
                                    * pool exhausted; a warm sort whose buckets overflowed; more unmerged EPS-chain bodies than the
                                    * exact-sum class tolerates; in the bit-exact mode: EPS clusters the reference-fold replay cannot
                                    * reproduce) plus the steps sent straight to the host build after refusals in a row (2, 4 .. 32
                                    * steps, then the device is tried again; env NBX_BH_BACKOFF_MAX = longest run, 0 = always try
                                    * the device).  A fast-mode refusal of the reference-fold class is NOT one of these: see
                                    * NBX_STAT_BH_CLASS_SWITCHES */
    NBX_STAT_BH_LAST_TREE = 1,     /* where the tree of the last Barnes-Hut evaluation was built: 0 host, 1 device */
    NBX_STAT_DRAW_AMBIGUOUS = 2,   /* tails the last device draw left to the host; -1 = the last draw ran on the host */
    NBX_STAT_BH_REFUSAL = 3,       /* why the last device tree build that was refused (its evaluation served by another class) was
                                    * (0: none has yet) -- 0x10000 = the node pool or the fold queue overflowed; 0x100000 = a bucket of the warm sort outgrew its
                                    * slots (bh_sort.hip, round 5); else bits of the
                                    * cluster replay (NBX_OPT_BH_FOLD = 1): 1 more than 512 entities around one point, 4 a cluster
                                    * of more than 48 entities / 96 bodies, 8 a merge hinges on another cluster, 16 an outsider
                                    * within EPS of a blob's centre, 32 two entities in one level-31 cell that do not merge,
                                    * 64 more than 4 096 bodies to move, 128 a blob's successive centres part ways above its
                                    * leaf, 512 (bit-exact mode) a leaf deeper than 25 levels, where the reference may panic, 256 a leaf of more bodies than the leaf fold orders; exact-sum class: 0x20000 = more
                                    * bodies than it tolerates in chains of close bodies that its replay only approximates (NBX_STAT_BH_CHAIN_APPROX; rounds 2-5:
                                    * bodies left unmerged by the pairs-only merge) */
    NBX_STAT_BH_CLASS_SWITCHES = 4, /* fast mode, NBX_OPT_BH_FOLD = 1 (round 6): evaluations the reference-fold device build was
                                    * selected for but the exact-sum DEVICE build served -- a refused build redone there, and the
                                    * back-off run (2, 4 .. 32 steps) behind refusals in a row.  Such a step stays on the GPU and
                                    * inside the fast mode's stated tolerance (the exact-sum class's) */
    NBX_STAT_BH_COLD_RESORTS = 5,  /* device builds whose warm sort overflowed a bucket (NBX_STAT_BH_REFUSAL 0x100000: more than 4 096
                                    * bodies on one 62-bit key, or a reshuffled system) and that were redone at once from a cold sort,
                                    * same class, on the device (round 6; round 5 sent such a step to the host build); the next
                                    * 2, 4 .. 32 builds then sort cold as well */
    NBX_STAT_BH_CHAIN_MERGED = 6,  /* exact-sum device build (round 6): bodies the last accepted build merged into another body's leaf
                                    * by replaying chains of bodies within EPS in arrival order (nbody.rs:249-260; rounds 2-5 merged
                                    * pairs only and handed crowded systems to the host build) */
    NBX_STAT_BH_CHAIN_APPROX = 7   /* ... and what of that build was approximate: bodies of blobs that end at a cut of a chain of more than 60
                                    * linked bodies (such chains are replayed in pieces) + merges decided with the search for an earlier arrival deeper in
                                    * the newcomer's path cut short (more than 256 sorted neighbours share the cell: taken as merges).
                                    * At most max(16, n/2000) in a build that was accepted: beyond that the step goes to the host build
                                    * (NBX_STAT_BH_REFUSAL 0x20000) */
};

enum nbx_kernel_id {
    NBX_K_FORCE = 0,     /* all-pairs force tile kernel (fast or strict) */
    NBX_K_INTEGRATE = 1, /* partial-sum reduce + kick-drift */
    NBX_K_BH_EVAL = 2,   /* Barnes-Hut traversal + kick-drift + velocity-kill */
    NBX_K_EXCHANGE = 3,  /* nbx_group_*: the per-step all-gather as seen from this engine's stream (includes the wait
                          * for the slowest peer) */
    NBX_K_TREE_BUILD = 4, /* the device quadtree build, first launch to last (GPU time on the engine's stream) */
    NBX_K_COUNT = 5
};

typedef struct nbx_device_info {
    char name[128];
    char arch[64];
    int32_t compute_units;
    int32_t clock_khz;         /* max engine clock */
    int32_t wavefront_size;
    int32_t lds_bytes_per_cu;
    double peak_fp32_flops;    /* CUs * clock * 256 flop/clk/CU (vector FMA roofline) */
    uint64_t hbm_bytes;
} nbx_device_info;

const char *nbx_last_error(void); /* thread-local text of the most recent failure */
const char *nbx_version(void);
int32_t nbx_device_count(void);   /* 0 when no device / no driver */
int32_t nbx_device_info_get(int32_t device, nbx_device_info *out);

/* Lifetime. Device resources are created lazily at the first device operation, so an engine can
 * be created and used for host-side work (presets, set/get, draw, tree build) without a GPU. */
int32_t nbx_create(nbx_engine **out, int32_t device);
void nbx_destroy(nbx_engine *e);

int32_t nbx_set_option(nbx_engine *e, int32_t option, int64_t value);
int64_t nbx_get_option(const nbx_engine *e, int32_t option);
/* as nbx_get_option with the status apart from the value: -1 is a legitimate value of some options ("by size" / "by cost" of
 * NBX_OPT_DRAW_DEVICE / NBX_OPT_BH_TREE / NBX_OPT_BH_FOLD) and also NBX_ERR_INVALID */
int32_t nbx_query_option(const nbx_engine *e, int32_t option, int64_t *value);
/* enum nbx_stat.  INT64_MIN for an unknown stat or a null engine (-1 is a legitimate value: NBX_STAT_DRAW_AMBIGUOUS, "the last
 * draw ran on the host"), and INT64_MIN as well when reading the verdict of a step still in flight failed (a refused step's redo
 * hit an error: nbx_last_error() has the text) -- the counters would be stale */
int64_t nbx_get_stat(const nbx_engine *e, int32_t stat);

/* Presets: same sampling as nbody.rs:39-104, but from a seedable generator (splitmix64 -> top 24
 * bits -> [0,1) f32, the rand 0.3 `next_f32` construction). */
int32_t nbx_seed(nbx_engine *e, uint64_t seed);
int32_t nbx_random_disk(nbx_engine *e, int32_t n);
int32_t nbx_stable_orbits(nbx_engine *e, int32_t n, float rmin, float rmax);

/* The synthetic workloads BASELINE.json's configs are quoted on (NOT in the reference, whose only
 * generators are the two presets above), so that any host behind this ABI can run them. Stateless in `seed`: sample k of
 * the splitmix64 stream -> top 24 bits -> [0,1) f32; IEEE double arithmetic + the C library's sqrt/pow/cos/sin; one
 * rounding to f32 per stored value. Golden values: tests/golden/workload_*.npz.
 *   plummer_sphere: scale radius 5, r = 5/sqrt(u^(-2/3) - 1) clipped to 45, isotropic, v = 0, m = 1000/n; dim 2 sets z = 0
 *   two_galaxies:   two stable_orbits-style disks of n/2 bodies (1000-mass core + unit planets, r in [0.5, 12)), centres
 *                   (-+15, 0), bulk velocities (+-3, -+1), 2-D */
#define NBX_SEED_PLUMMER 0x5EED0001ull
#define NBX_SEED_TWO_GALAXIES 0x5EED0002ull
int32_t nbx_plummer_sphere(nbx_engine *e, int32_t n, uint64_t seed, int32_t dim);
int32_t nbx_two_galaxies(nbx_engine *e, int32_t n, uint64_t seed);

/* State in/out (host SoA buffers of n floats each). The 2-D forms set z = vz = 0.
 * get: `cap` = capacity of each output array; returns the particle count or a negative status.
 * Any output pointer may be NULL. */
int32_t nbx_num_particles(const nbx_engine *e);
int32_t nbx_set_particles(nbx_engine *e, int32_t n, const float *px, const float *py, const float *vx,
                          const float *vy, const float *m);
int32_t nbx_set_particles3(nbx_engine *e, int32_t n, const float *px, const float *py, const float *pz,
                           const float *vx, const float *vy, const float *vz, const float *m);
int32_t nbx_get_particles(nbx_engine *e, int32_t cap, float *px, float *py, float *vx, float *vy, float *m);
int32_t nbx_get_particles3(nbx_engine *e, int32_t cap, float *px, float *py, float *pz, float *vx, float *vy,
                           float *vz, float *m);

/* Checkpoint to / resume from a file (additive; the reference has no persistence). Format: "NBXCKPT1",
 * int32 n, int32 0, then px py pz vx vy vz m as n little-endian f32 each. load returns the particle count. */
int32_t nbx_save(nbx_engine *e, const char *path);
int32_t nbx_load(nbx_engine *e, const char *path);

/* Steps. Asynchronous on the engine's stream; nbx_synchronize / get / draw wait. */
int32_t nbx_step_brute_force(nbx_engine *e, float dt);
int32_t nbx_step_barnes_hut(nbx_engine *e, float theta, float dt, int32_t nthreads);
int32_t nbx_synchronize(nbx_engine *e);

/* Force evaluation without state update, for parity tests on accelerations.
 * Writes F_i = m_i * a_i (the quantity nbody.rs:140-142 accumulates) for this engine's slab
 * targets; fz may be NULL. theta == 0 -> all-pairs, else Barnes-Hut traversal. cap >= slab size. */
int32_t nbx_forces(nbx_engine *e, float theta, int32_t cap, float *fx, float *fy, float *fz);

int32_t nbx_draw(nbx_engine *e, int32_t w, int32_t h, uint32_t *fb);

/* Host quadtree exactly as nbody.rs:388-415 builds it, dumped in pre-order (children UL,UR,LL,LR)
 * as rows of 8 floats: x1,y1,x2,y2,px,py,m,has_children. Returns node count (may exceed cap; only
 * cap rows are written) or a negative status. Runs without a device. */
int32_t nbx_bh_tree_dump(nbx_engine *e, float *rows, int32_t cap);

/* The flattened tree the GPU traversal walks: pre-order, empty exterior nodes dropped, 32-byte records
 * {float px, py, m, s; int32 skip, interior, pad, pad}. threaded = 1 uses the host-thread flattener
 * (same bytes), 2 dumps the DEVICE-built tree (needs a GPU). Returns the node count; writes only when cap >= count. Runs without a device. */
int32_t nbx_bh_flat_dump(nbx_engine *e, void *rows, int32_t cap, int32_t threaded);

/* ---- multi-GPU: bodies shard as contiguous slabs of targets, the reference's own thread split  */
/* (range = N/world, last rank takes the remainder; nbody.rs:426-428).  One process per GPU; every */
/* rank holds the full (x,y,z,m) source array and its own slab's velocities.  Per step:            */
/*   nbx_step_local()  ->  caller all-gathers the positions buffer (one RCCL all-gather, in place) */
int32_t nbx_set_shard(nbx_engine *e, int32_t rank, int32_t world); /* call before set_particles */
int32_t nbx_get_slab(const nbx_engine *e, int32_t *lo, int32_t *hi);
/* Use a caller-owned DEVICE buffer (e.g. a torch tensor handed to torch.distributed) as the
 * (x,y,z,m) float4 array instead of an engine-owned one. bytes >= nbx_positions_bytes(), 16-byte aligned. Call
 * after set_particles; the engine copies its current positions into it. */
int32_t nbx_bind_positions(nbx_engine *e, void *device_ptr, size_t bytes);
/* fp16 source copy (NBX_OPT_SOURCE_PRECISION = 16): in sharded runs THIS buffer is what the per-step
 * all-gather moves (half the bytes); bind a caller-owned device buffer of >= nbx_half_sources_bytes(). */
size_t nbx_half_sources_bytes(const nbx_engine *e); /* n_padded * 8 */
int32_t nbx_bind_half_sources(nbx_engine *e, void *device_ptr, size_t bytes);
/* device pointer of the float4 (x,y,z,m) array.  The caller may WRITE through it (sharded runs: the per-step all-gather lands
 * here); the engine therefore treats every call as "positions may have moved" (anything it derived from them is rebuilt) */
void *nbx_positions_device(nbx_engine *e);
size_t nbx_positions_bytes(const nbx_engine *e); /* n_padded * 16 */
int32_t nbx_set_stream(nbx_engine *e, void *hip_stream); /* run on a caller-owned hipStream_t */
/* force + integrate for this rank's slab only, writing the new positions into the slab's slot of
 * the positions buffer. The caller then performs the all-gather on the same stream. */
int32_t nbx_step_local(nbx_engine *e, float dt);

/* ---- single-process multi-GPU group: G engines (one per device) behind one handle, for hosts that cannot run
 * one process per GPU (the unmodified Haskell caller: set NB_GPUS=<n>|all and the six nb_* symbols use it).
 * Same slab sharding and the same single exchange per step as above, but the all-gather is issued by the
 * library itself through RCCL (ncclCommInitAll; librccl is dlopen'ed on first use). devices may be NULL
 * (0..count-1). Per-engine calls (options, profiling, forces) remain available through nbx_group_engine.
 * Barnes-Hut steps build the quadtree once per step for the whole group (host build: engine 0's copy, node
 * array sent to every device; device build: all devices concurrently).
 * fp16 sources (NBX_OPT_SOURCE_PRECISION = 16 on every engine, fast mode): the per-step all-gather moves the half4
 * source copy instead (ncclFloat16, half the bytes; BASELINE config #5). The fp32 positions of the other slabs are then
 * re-gathered lazily, when a group call needs them (get_particles, draw, a Barnes-Hut or bit-exact step); per-engine
 * calls through nbx_group_engine see them only after such a call.
 * NBX_GROUP_EXCHANGE=copy (environment, read at nbx_group_create): replace the RCCL all-gather by
 * event-ordered hipMemcpyPeerAsync pulls -- no communicator, no librccl, and a device may then be listed more
 * than once (several engines sharing one GPU: how the group logic is tested on a single-GPU box).
 * If RCCL cannot be used (librccl missing, ncclCommInitAll or a collective fails) the group does NOT die: it switches to
 * those peer copies (peer access enabled where the devices allow it), redoes the exchange, prints one line on stderr and
 * reports it through nbx_group_info / nbx_group_exchange_note.
 * NBX_GROUP_ENQUEUE=threads (or nbx_group_set_enqueue_threads): one persistent host thread per device enqueues that
 * device's kernels and its share of the exchange, instead of one thread walking the devices in turn. */
typedef struct nbx_group nbx_group;
int32_t nbx_group_create(nbx_group **out, const int32_t *devices, int32_t count);
void nbx_group_destroy(nbx_group *g);
int32_t nbx_group_size(const nbx_group *g);
nbx_engine *nbx_group_engine(nbx_group *g, int32_t i);
int32_t nbx_group_set_option(nbx_group *g, int32_t option, int64_t value);
int32_t nbx_group_num_particles(const nbx_group *g);
int32_t nbx_group_set_particles3(nbx_group *g, int32_t n, const float *px, const float *py, const float *pz,
                                 const float *vx, const float *vy, const float *vz, const float *m);
int32_t nbx_group_get_particles3(nbx_group *g, int32_t cap, float *px, float *py, float *pz, float *vx, float *vy,
                                 float *vz, float *m);
int32_t nbx_group_step_brute_force(nbx_group *g, float dt);
int32_t nbx_group_step_barnes_hut(nbx_group *g, float theta, float dt, int32_t nthreads);
int32_t nbx_group_synchronize(nbx_group *g);
int32_t nbx_group_draw(nbx_group *g, int32_t w, int32_t h, uint32_t *fb);
int32_t nbx_group_exchanges(const nbx_group *g); /* all-gathers issued so far */
enum nbx_group_info_id {
    NBX_GROUP_INFO_EXCHANGE = 0,        /* 0 = RCCL all-gather, 1 = peer copies (NBX_GROUP_EXCHANGE=copy), 2 = peer copies
                                         * because RCCL failed (nbx_group_exchange_note says what failed) */
    NBX_GROUP_INFO_RCCL_RANKS = 1,      /* ranks ncclCommInitAll was given; 0 while / when no communicator exists */
    NBX_GROUP_INFO_ENQUEUE_THREADS = 2, /* enqueue threads in use (0 = the caller's thread walks the devices) */
    NBX_GROUP_INFO_FP32_STALE = 3       /* 1 = only the fp16 source copy was exchanged last; fp32 positions re-gather lazily */
};
int64_t nbx_group_info(const nbx_group *g, int32_t what);
const char *nbx_group_exchange_note(const nbx_group *g); /* "" unless the group fell back to peer copies */
int32_t nbx_group_set_enqueue_threads(nbx_group *g, int32_t on);

/* ---- profiling: HIP event pairs on the engine's stream around each kernel launch -------------- */
int32_t nbx_profile_reset(nbx_engine *e);
/* total milliseconds and launch count recorded for `kernel_id` since the last reset (synchronises) */
int32_t nbx_profile_read(nbx_engine *e, int32_t kernel_id, double *total_ms, int32_t *launches);
/* Barnes-Hut host-side phases since the last call (cumulative ms, then reset): ms4 = download of
 * positions, quadtree build, flatten, upload of the node array; steps = tree builds; nodes = size of
 * the last flattened tree. */
int32_t nbx_bh_host_timing(nbx_engine *e, double *ms4, int32_t *steps, int32_t *nodes);
/* Work of one Barnes-Hut evaluation on the current state: tree nodes visited and pair laws evaluated, summed
 * over this engine's slab (for roofline accounting; runs a counting traversal, no state change). */
int32_t nbx_bh_work(nbx_engine *e, float theta, uint64_t *node_visits, uint64_t *pair_evals);
/* The same counting traversal in more detail: out4 = { node visits, pair laws evaluated (nbody.rs:164-184: 12 flops each in 2-D),
 * opening tests = visits of INTERIOR nodes (nbody.rs:341-345: 2 sub, 2 mul, 1 add, 1 sqrt, 1 div = 7 flops each as written),
 * child groups loaded per body summed over the slab (NBX_OPT_BH_WALK = 1; 0 for the node walk) }. */
int32_t nbx_bh_work_detail(nbx_engine *e, float theta, uint64_t *out4);
/* Timeline of one traversal by the child-group walk (fast mode; tools/bh_walk_trace.py): 4 words per walk (= workgroup, in launch
 * order) -- s_memrealtime (10 ns ticks) at its start and end, groups loaded (bit 31: the walk outgrew its register stack and was redone with the LDS spill) | first-body chunk << 32, HW_ID | XCC_ID << 32.  Returns the number of
 * walks (may exceed cap_walks: nothing is written then) or a negative status. */
int32_t nbx_bh_walk_trace(nbx_engine *e, float theta, int32_t cap_walks, uint64_t *out);
/* The opening threshold of bh_threshold.h: the float T with  (s / sqrt(d2) < theta, nbody.rs:344-345)  <=>  d2 > T  for every
 * float d2 >= 0.  Host evaluation; nbx_bh_take_thresholds_device evaluates count of them on the engine's GPU (test hook: the walk
 * relies on host and device agreeing with the reference's own arithmetic). */
float nbx_bh_take_threshold(float s, float theta);
int32_t nbx_bh_take_thresholds_device(nbx_engine *e, int32_t count, const float *s, const float *theta, float *out);
/* launch geometry the last force launch used (for DESIGN/bench reporting); any pointer may be NULL.
 * Bit-exact kernel: jsplit = 1 (the source loop is never split), bodies_per_thread = 1 and
 * variant = -(NBX_OPT_STRICT_KERNEL actually used): -16 / -8 = waves per workgroup of 64 targets, -1 = one thread per body. */
int32_t nbx_last_launch(const nbx_engine *e, int32_t *grid, int32_t *block, int32_t *jsplit,
                        int32_t *bodies_per_thread, int32_t *dim, int32_t *variant);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_MI355X_H */
