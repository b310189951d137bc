/*
 * nbody_oracle.c -- CPU restatement of the reference N-body hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * What this is: a line-faithful plain-C restatement of blitzcode/rust-exp `rs-src/nbody.rs`
 * (the O(N^2) pairwise force + kick-drift integrator, the Barnes-Hut quadtree step, the two
 * presets and nb_draw).  It exists to CHECK the MI355X HIP path, never to be the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
 *  and cannot be compiled here (Rust 2016-era crate; no rustc/cargo in the
 * image), so this restatement cannot be checked against reference OUTPUT.  It is pinned instead
 * by (1) independent restatements written from the Rust source in another language --
 * oracle/nbody_numpy.py (brute-force step), oracle/nbody_bh_py.py (Barnes-Hut step),
 * oracle/nbody_draw_py.py (nb_draw) and oracle/nbody_presets_py.py (presets) -- that must agree
 * with this file bit for bit
 * (tests/test_oracle_golden.py, tests/test_oracle_restatements.py),
 * (2) known-answer tests implied by the source semantics (tests/test_oracle_kat.py),
 * (3) an fp64 arbiter.  Golden vectors under tests/golden/ are produced by THIS file.
 *
 * Build (recipe: oracle/Makefile):
 *   gcc -O2 -std=c11 -fno-fast-math -ffp-contract=off -fPIC -shared -pthread
 * `-ffp-contract=off` matters: rustc never fuses a*b+c, gcc does by default in GNU mode.
 * All arithmetic is IEEE binary32 evaluated in source order, as rustc emits it on x86-64 (SSE).
 *
 * Every function cites the reference lines (relative to /root/reference/) it follows.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* nbody.rs:13-17 */
static const float VP_WDH = 100.0f;
static const float VP_ORG_X = 0.0f;
static const float VP_ORG_Y = 0.0f;
static const float EPS = 0.0001f;

/* nbody.rs:19-26 -- struct Particle {px,py,vx,vy,m}; AoS, 20 bytes */
typedef struct {
    float px, py, vx, vy, m;
} orc_particle;

/* error codes standing in for the reference's panics */
enum {
    ORC_OK = 0,
    ORC_PANIC_DEPTH = -1,      /* nbody.rs:230-232 */
    ORC_PANIC_SAME_POS = -2,   /* nbody.rs:267 */
    ORC_PANIC_SUBDIVIDE = -3,  /* nbody.rs:293 */
    ORC_PANIC_MASS = -4,       /* nbody.rs:304 */
    ORC_PANIC_NTHREADS = -5,   /* NOT a reference panic: returned by the helpers of this file that need >= 1 worker
                                * (orc_brute_forces_mt, orc_bh_forces*). nb_step_barnes_hut with nthreads <= 0 does not panic:
                                * the division `len / nthreads` sits inside the (0..nthreads).map closure, nbody.rs:424-428,
                                * which never runs -- see orc_step_barnes_hut */
    ORC_PANIC_ALLOC = -6
};

/* ------------------------------------------------------------------------------------------ */
/* nbody.rs:164-184  fn force(px1,py1,m1,px2,py2,m2) -> (f32,f32)                              */
/* Un-normalised direction: magnitude ~ 1/r.  One IEEE divide per pair.                        */
void orc_force(float px1, float py1, float m1, float px2, float py2, float m2, float *fx, float *fy)
{
    float dx = px2 - px1;                 /* :174 */
    float dy = py2 - py1;                 /* :175 */
    float dist_sq = dx * dx + dy * dy;    /* :176 */
    float f = m1 * m2 / (dist_sq + EPS);  /* :180  (m1*m2)/(dist_sq+EPS) */
    *fx = f * dx;                         /* :183 */
    *fy = f * dy;
}

/* nbody.rs:132-144 restricted to targets [i0,i1): forces only, no state update.
 * Used by the parity tests at sizes where a full CPU step is too slow, and by the cpu_baseline
 * timing leg (work is uniform per i, so an i-slice times the same loop). */
void orc_brute_forces(const orc_particle *p, int n, int i0, int i1, float *fx, float *fy)
{
    for (int i = i0; i < i1; i++) {
        float ax = 0.0f, ay = 0.0f;       /* :130 Force{0,0} */
        const orc_particle *a = &p[i];
        for (int j = 0; j < n; j++) {     /* :135 ascending j, sequential f32 sum */
            if (i == j) continue;         /* :136 skip by INDEX */
            const orc_particle *b = &p[j];
            float fx_add, fy_add;
            orc_force(a->px, a->py, a->m, b->px, b->py, b->m, &fx_add, &fy_add); /* :140 */
            ax += fx_add;                 /* :141 */
            ay += fy_add;                 /* :142 */
        }
        fx[i - i0] = ax;
        fy[i - i0] = ay;
    }
}

/* nbody.rs:153-160 kick-drift (semi-implicit Euler): v += (dt*F)/m ; p += dt*v_new */
static void integrate_one(orc_particle *q, float fx, float fy, float dt)
{
    q->vx += dt * fx / q->m;              /* :155  (dt*fx)/m */
    q->vy += dt * fy / q->m;              /* :156 */
    q->px += dt * q->vx;                  /* :158 uses the UPDATED v */
    q->py += dt * q->vy;                  /* :159 */
}

/* nbody.rs:106-162  nb_step_brute_force(dt): all forces from OLD positions, then update */
int orc_step_brute_force(orc_particle *p, int n, float dt)
{
    if (n <= 0) return ORC_OK;
    float *fx = (float *)malloc(sizeof(float) * (size_t)n * 2);
    if (!fx) return ORC_PANIC_ALLOC;
    float *fy = fx + n;
    orc_brute_forces(p, n, 0, n, fx, fy);
    for (int i = 0; i < n; i++) integrate_one(&p[i], fx[i], fy[i], dt);
    free(fx);
    return ORC_OK;
}

/* Multi-thread CPU baseline: the reference's brute force is single-threaded (nbody.rs:132-144);
 * this applies the reference's OWN static slab split (nbody.rs:426-428: range=N/T, last thread
 * takes the remainder) to the i-loop.  Results are bit-identical to the 1-thread path because
 * every force depends only on old positions. */
typedef struct {
    const orc_particle *p;
    int n, i0, i1;
    float *fx, *fy;
} brute_job;

static void *brute_worker(void *arg)
{
    brute_job *j = (brute_job *)arg;
    orc_brute_forces(j->p, j->n, j->i0, j->i1, j->fx + j->i0, j->fy + j->i0);
    return NULL;
}

/* forces for targets [0,ni) against all n sources with T threads; fx,fy sized ni */
int orc_brute_forces_mt(const orc_particle *p, int n, int ni, int nthreads, float *fx, float *fy)
{
    if (nthreads <= 0) return ORC_PANIC_NTHREADS;
    if (ni > n) ni = n;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    brute_job *jobs = (brute_job *)malloc(sizeof(brute_job) * (size_t)nthreads);
    if (!th || !jobs) { free(th); free(jobs); return ORC_PANIC_ALLOC; }
    int range = ni / nthreads;                                     /* :426 */
    for (int t = 0; t < nthreads; t++) {
        int lo = range * t;                                        /* :427 */
        int hi = (t == nthreads - 1) ? ni : range * (t + 1);       /* :428 */
        jobs[t] = (brute_job){p, n, lo, hi, fx, fy};
        pthread_create(&th[t], NULL, brute_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
    return ORC_OK;
}

int orc_step_brute_force_mt(orc_particle *p, int n, float dt, int nthreads)
{
    if (n <= 0) return ORC_OK;
    float *fx = (float *)malloc(sizeof(float) * (size_t)n * 2);
    if (!fx) return ORC_PANIC_ALLOC;
    float *fy = fx + n;
    int rc = orc_brute_forces_mt(p, n, n, nthreads, fx, fy);
    if (rc == ORC_OK)
        for (int i = 0; i < n; i++) integrate_one(&p[i], fx[i], fy[i], dt);
    free(fx);
    return rc;
}

/* fp64 arbiter: same law, same (unfactored) form, double arithmetic, for targets [i0,i1).
 * Not in the reference; used to show the GPU error is no worse than the f32 oracle's own. */
void orc_brute_forces_f64(const orc_particle *p, int n, int i0, int i1, double *fx, double *fy)
{
    for (int i = i0; i < i1; i++) {
        double ax = 0.0, ay = 0.0;
        for (int j = 0; j < n; j++) {
            if (i == j) continue;
            double dx = (double)p[j].px - (double)p[i].px;
            double dy = (double)p[j].py - (double)p[i].py;
            double f = (double)p[i].m * (double)p[j].m / (dx * dx + dy * dy + (double)EPS);
            ax += f * dx;
            ay += f * dy;
        }
        fx[i - i0] = ax;
        fy[i - i0] = ay;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Barnes-Hut: nbody.rs:186-480                                                                */

/* nbody.rs:206-214  struct Node; children: Option<Box<[Node;4]>> in order [UL,UR,LL,LR] */
typedef struct orc_node {
    float x1, y1, x2, y2;
    float px, py, m;
    struct orc_node *children; /* NULL or block of 4 */
    double em, emx, emy;       /* NOT in the reference: exact (fp64) mass and first moments, filled by exact_sums() for the
                                * arbiter below; never read by the restatement of the reference's own code */
} orc_node;

static void node_new(orc_node *nd, float x1, float y1, float x2, float y2) /* :216-222 */
{
    nd->x1 = x1; nd->y1 = y1; nd->x2 = x2; nd->y2 = y2;
    nd->px = 0.0f; nd->py = 0.0f; nd->m = 0.0f;
    nd->children = NULL;
}

static void node_free(orc_node *nd)
{
    if (nd->children) {
        for (int i = 0; i < 4; i++) node_free(&nd->children[i]);
        free(nd->children);
        nd->children = NULL;
    }
}

/* nbody.rs:303-320 */
static int node_add_mass(orc_node *nd, float px, float py, float m)
{
    if (!(m > 0.0f)) return ORC_PANIC_MASS;           /* :304 assert!(m > 0.0) */
    if (nd->m == 0.0f) {                               /* :305 empty: copy position EXACTLY */
        nd->px = px; nd->py = py; nd->m = m;
    } else {
        float inv_msum = 1.0f / (nd->m + m);           /* :315 */
        nd->px = (nd->px * nd->m + px * m) * inv_msum; /* :316 */
        nd->py = (nd->py * nd->m + py * m) * inv_msum; /* :317 */
        nd->m += m;                                    /* :318 */
    }
    return ORC_OK;
}

/* nbody.rs:322-331 ; returns index into [UL,UR,LL,LR] */
static int node_quadrant(const orc_node *nd, float x, float y)
{
    float cx = (nd->x1 + nd->x2) * 0.5f;
    float cy = (nd->y1 + nd->y2) * 0.5f;
    if (y < cy) return (x < cx) ? 2 /*LL*/ : 3 /*LR*/;
    return (x < cx) ? 0 /*UL*/ : 1 /*UR*/;
}

/* nbody.rs:286-301 */
static int node_create_children(orc_node *nd)
{
    float cx = (nd->x1 + nd->x2) * 0.5f;
    float cy = (nd->y1 + nd->y2) * 0.5f;
    if (!(cx > nd->x1 || cx < nd->x2 || cy > nd->y1 || cy < nd->y2)) return ORC_PANIC_SUBDIVIDE; /* :293 */
    orc_node *c = (orc_node *)malloc(sizeof(orc_node) * 4);
    if (!c) return ORC_PANIC_ALLOC;
    node_new(&c[0], nd->x1, cy, cx, nd->y2);     /* UL :296 */
    node_new(&c[1], cx, cy, nd->x2, nd->y2);     /* UR :297 */
    node_new(&c[2], nd->x1, nd->y1, cx, cy);     /* LL :298 */
    node_new(&c[3], cx, nd->y1, nd->x2, cy);     /* LR :299 */
    nd->children = c;
    return ORC_OK;
}

/* nbody.rs:226-284 */
static int node_insert(orc_node *nd, float px, float py, float m, unsigned depth)
{
    int rc;
    if (depth > 50) return ORC_PANIC_DEPTH;                        /* :230 */
    if (nd->children) {                                            /* :234 interior */
        if ((rc = node_add_mass(nd, px, py, m)) != ORC_OK) return rc;  /* :236 */
        int q = node_quadrant(nd, px, py);                         /* :237 */
        return node_insert(&nd->children[q], px, py, m, depth + 1);/* :240 */
    }
    int too_close = fabsf(nd->px - px) < EPS && fabsf(nd->py - py) < EPS; /* :249 */
    if (nd->m == 0.0f || too_close) {                              /* :250 */
        return node_add_mass(nd, px, py, m);                       /* :260 */
    }
    if (!(nd->px != px || nd->py != py)) return ORC_PANIC_SAME_POS;/* :267 */
    float px_o = nd->px, py_o = nd->py, m_o = nd->m;               /* :271-273 */
    nd->px = 0.0f; nd->py = 0.0f; nd->m = 0.0f;                    /* :274-276 */
    if ((rc = node_create_children(nd)) != ORC_OK) return rc;      /* :277 */
    if ((rc = node_insert(nd, px_o, py_o, m_o, depth + 1)) != ORC_OK) return rc; /* :278 */
    return node_insert(nd, px, py, m, depth + 1);                  /* :281 */
}

/* nbody.rs:333-377 ; hierarchical summation: an interior node returns sum of children 0..3 */
static void node_compute_force(const orc_node *nd, float px, float py, float m, float theta,
                               float *ofx, float *ofy)
{
    float fx = 0.0f, fy = 0.0f;                                    /* :336-337 */
    if (nd->children) {
        float s = nd->x2 - nd->x1;                                 /* :341 x-extent only */
        float dx = nd->px - px;                                    /* :342 */
        float dy = nd->py - py;                                    /* :343 */
        float d = sqrtf(dx * dx + dy * dy);                        /* :344 */
        if (s / d < theta) {                                       /* :345 */
            orc_force(px, py, m, nd->px, nd->py, nd->m, &fx, &fy); /* :348-350 assigns */
        } else {
            for (int i = 0; i < 4; i++) {                          /* :354 */
                float cx, cy;
                node_compute_force(&nd->children[i], px, py, m, theta, &cx, &cy);
                fx += cx;                                          /* :358 */
                fy += cy;                                          /* :359 */
            }
        }
    } else {
        if (nd->px == px && nd->py == py) { *ofx = 0.0f; *ofy = 0.0f; return; } /* :365 */
        if (nd->m == 0.0f) { *ofx = 0.0f; *ofy = 0.0f; return; }                /* :368 */
        orc_force(px, py, m, nd->px, nd->py, nd->m, &fx, &fy);     /* :371-373 */
    }
    *ofx = fx;
    *ofy = fy;
}

/* nbody.rs:388-415 : root AABB (not squared) + sequential insert in particle-index order */
static int build_tree(const orc_particle *p, int n, orc_node *root)
{
    float x1 = 3.40282347e+38f, y1 = 3.40282347e+38f;   /* f32::MAX :388-389 */
    float x2 = -3.40282347e+38f, y2 = -3.40282347e+38f; /* f32::MIN :390-391 */
    for (int i = 0; i < n; i++) {
        x1 = p[i].px < x1 ? p[i].px : x1;   /* :394 */
        y1 = p[i].py < y1 ? p[i].py : y1;
        x2 = p[i].px > x2 ? p[i].px : x2;
        y2 = p[i].py > y2 ? p[i].py : y2;
    }
    node_new(root, x1, y1, x2, y2);         /* :410 */
    for (int i = 0; i < n; i++) {           /* :413-415 */
        int rc = node_insert(root, p[i].px, p[i].py, p[i].m, 0);
        if (rc != ORC_OK) return rc;
    }
    return ORC_OK;
}

typedef struct {
    orc_particle *p;
    const orc_node *tree;
    int lo, hi;
    float theta, dt;
    float *fx, *fy; /* optional: record forces instead of updating */
} bh_job;

static void *bh_worker(void *arg)
{
    bh_job *j = (bh_job *)arg;
    for (int i = j->lo; i < j->hi; i++) {                                  /* :443 */
        orc_particle *q = &j->p[i];
        float fx, fy;
        node_compute_force(j->tree, q->px, q->py, q->m, j->theta, &fx, &fy); /* :447 */
        if (j->fx) { j->fx[i] = fx; j->fy[i] = fy; continue; }
        integrate_one(q, fx, fy, j->dt);                                   /* :453-458 */
        if (fabsf(VP_ORG_X - q->px) > VP_WDH * 0.55f ||                    /* :466 */
            fabsf(VP_ORG_Y - q->py) > VP_WDH * 0.55f) {                    /* :467 */
            q->vx = 0.0f;                                                  /* :469 */
            q->vy = 0.0f;
        }
    }
    return NULL;
}

static int bh_run(orc_particle *p, int n, float theta, float dt, int nthreads, float *fx, float *fy)
{
    if (nthreads <= 0) return ORC_PANIC_NTHREADS;
    orc_node root;
    int rc = build_tree(p, n, &root);
    if (rc != ORC_OK) { node_free(&root); return rc; }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    bh_job *jobs = (bh_job *)malloc(sizeof(bh_job) * (size_t)nthreads);
    if (!th || !jobs) { free(th); free(jobs); node_free(&root); return ORC_PANIC_ALLOC; }
    int range = n / nthreads;                                              /* :426 */
    for (int t = 0; t < nthreads; t++) {
        int lo = range * t;                                                /* :427 */
        int hi = (t == nthreads - 1) ? n : range * (t + 1);                /* :428 */
        jobs[t] = (bh_job){p, &root, lo, hi, theta, dt, fx, fy};
        pthread_create(&th[t], NULL, bh_worker, &jobs[t]);                 /* :440 */
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);          /* :476-478 */
    free(th);
    free(jobs);
    node_free(&root);
    return ORC_OK;
}

/* nbody.rs:186-480  nb_step_barnes_hut(theta, dt, nthreads) */
int orc_step_barnes_hut(orc_particle *p, int n, float theta, float dt, int nthreads)
{
    if (theta == 0.0f) return orc_step_brute_force(p, n, dt);  /* :197-200 exact compare */
    if (nthreads <= 0) {
        /* :424 `(0..nthreads).map(|i| {...})` is an empty iterator: no worker is spawned, the closure holding the
         * division by nthreads (:426) never runs, no particle is touched.  The tree IS built first (:380-417), so its
         * asserts can still fire. */
        orc_node root;
        int rc = build_tree(p, n, &root);
        node_free(&root);
        return rc;
    }
    return bh_run(p, n, theta, dt, nthreads, NULL, NULL);
}

/* forces only (no update) through the same tree + traversal; theta must be != 0 */
int orc_bh_forces(const orc_particle *p, int n, float theta, int nthreads, float *fx, float *fy)
{
    return bh_run((orc_particle *)p, n, theta, 0.0f, nthreads, fx, fy);
}

/* ------------------------------------------------------------------------------------------ */
/* fp64 ARBITER for Barnes-Hut (not in the reference; SURVEY.md 8(d) asks for an fp64 arbiter).   */
/* The reference's own tree (same insertion, same merges, same boxes) and its own opening law     */
/* s/d < theta and pair law, but with every interior node's mass and centre taken as the EXACT    */
/* sum / weighted mean of its leaves (fp64) instead of the f32 running fold of nbody.rs:303-320,  */
/* and all arithmetic of the traversal in fp64.  Whoever is closer to this -- the f32 restatement */
/* or a GPU build with exactly rounded centres -- is closer to what the algorithm means.          */
static void exact_sums(orc_node *nd)
{
    if (!nd->children) {
        nd->em = (double)nd->m; nd->emx = (double)nd->m * (double)nd->px; nd->emy = (double)nd->m * (double)nd->py;
        return;
    }
    nd->em = nd->emx = nd->emy = 0.0;
    for (int i = 0; i < 4; i++) {
        exact_sums(&nd->children[i]);
        nd->em += nd->children[i].em; nd->emx += nd->children[i].emx; nd->emy += nd->children[i].emy;
    }
}

static void force_f64(double px1, double py1, double m1, double px2, double py2, double m2, double *fx, double *fy)
{
    double dx = px2 - px1, dy = py2 - py1;
    double f = m1 * m2 / (dx * dx + dy * dy + (double)EPS);
    *fx = f * dx; *fy = f * dy;
}

static void node_force_exact(const orc_node *nd, float px, float py, float m, double theta, double *ofx, double *ofy)
{
    double fx = 0.0, fy = 0.0;
    if (nd->children) {
        double s = (double)(nd->x2 - nd->x1);                       /* the f32 box width, as :341 */
        double cx = nd->emx / nd->em, cy = nd->emy / nd->em;
        double dx = cx - (double)px, dy = cy - (double)py;
        double d = sqrt(dx * dx + dy * dy);
        if (s / d < theta) {
            force_f64(px, py, m, cx, cy, nd->em, &fx, &fy);
        } else {
            for (int i = 0; i < 4; i++) {
                double ax, ay;
                node_force_exact(&nd->children[i], px, py, m, theta, &ax, &ay);
                fx += ax; fy += ay;
            }
        }
    } else if (!(nd->px == px && nd->py == py) && nd->m != 0.0f) {  /* :365, :368 */
        force_f64(px, py, m, nd->px, nd->py, nd->m, &fx, &fy);
    }
    *ofx = fx; *ofy = fy;
}

typedef struct { const orc_particle *p; const orc_node *tree; int lo, hi; double theta; double *fx, *fy; } arb_job;

static void *arb_worker(void *arg)
{
    arb_job *j = (arb_job *)arg;
    for (int i = j->lo; i < j->hi; i++)
        node_force_exact(j->tree, j->p[i].px, j->p[i].py, j->p[i].m, j->theta, &j->fx[i], &j->fy[i]);
    return NULL;
}

int orc_bh_forces_exact(const orc_particle *p, int n, float theta, int nthreads, double *fx, double *fy)
{
    if (nthreads <= 0) return ORC_PANIC_NTHREADS;
    orc_node root;
    int rc = build_tree(p, n, &root);
    if (rc != ORC_OK) { node_free(&root); return rc; }
    if (n > 0) exact_sums(&root);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    arb_job *jobs = (arb_job *)malloc(sizeof(arb_job) * (size_t)nthreads);
    if (!th || !jobs) { free(th); free(jobs); node_free(&root); return ORC_PANIC_ALLOC; }
    int range = n / nthreads;
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (arb_job){p, &root, range * t, (t == nthreads - 1) ? n : range * (t + 1), (double)theta, fx, fy};
        pthread_create(&th[t], NULL, arb_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
    node_free(&root);
    return ORC_OK;
}

/* Tree statistics for tests: node count, leaf count, max depth */
static void tree_stats(const orc_node *nd, int depth, int *nodes, int *leaves, int *maxdepth)
{
    (*nodes)++;
    if (depth > *maxdepth) *maxdepth = depth;
    if (!nd->children) { if (nd->m != 0.0f) (*leaves)++; return; }
    for (int i = 0; i < 4; i++) tree_stats(&nd->children[i], depth + 1, nodes, leaves, maxdepth);
}

int orc_bh_tree_stats(const orc_particle *p, int n, int *nodes, int *leaves, int *maxdepth, float *root_m,
                      float *root_px, float *root_py)
{
    orc_node root;
    int rc = build_tree(p, n, &root);
    *nodes = *leaves = *maxdepth = 0;
    if (rc == ORC_OK) {
        tree_stats(&root, 0, nodes, leaves, maxdepth);
        *root_m = root.m; *root_px = root.px; *root_py = root.py;
    }
    node_free(&root);
    return rc;
}

/* Flatten the reference-faithful tree in pre-order (children order 0..3) so the product's own
 * host tree build (csrc/bh_tree.cpp) can be compared node for node.
 * out layout per node: x1,y1,x2,y2,px,py,m,has_children(0/1 as float) -> 8 floats. */
static void tree_dump(const orc_node *nd, float *out, int cap, int *count)
{
    if (*count < cap) {
        float *o = out + 8 * (size_t)(*count);
        o[0] = nd->x1; o[1] = nd->y1; o[2] = nd->x2; o[3] = nd->y2;
        o[4] = nd->px; o[5] = nd->py; o[6] = nd->m; o[7] = nd->children ? 1.0f : 0.0f;
    }
    (*count)++;
    if (nd->children)
        for (int i = 0; i < 4; i++) tree_dump(&nd->children[i], out, cap, count);
}

int orc_bh_tree_dump(const orc_particle *p, int n, float *out, int cap, int *count)
{
    orc_node root;
    int rc = build_tree(p, n, &root);
    *count = 0;
    if (rc == ORC_OK) tree_dump(&root, out, cap, count);
    node_free(&root);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Presets: nbody.rs:39-104.                                                                   */
/* The reference draws from rand 0.3.14 `thread_rng()` (Cargo.lock:123-129; OS-seeded, absent  */
/* from /root/reference) => preset OUTPUT is unpinnable.  What IS restated from rand 0.3's      */
/* published algorithm: f32 sample = top 24 bits of a u32 * 2^-24 in [0,1) (Rng::next_f32), and */
/* Range::new(lo,hi).ind_sample = lo + (hi-lo)*u.  The bit source here is splitmix64 (seedable, */
/* trivially identical in C++/numpy); one u64 per f32 sample, top 24 bits.                      */

static uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

float orc_next_f32(uint64_t *s) { return (float)(splitmix64(s) >> 40) * (1.0f / 16777216.0f); }

static float range_sample(float lo, float hi, uint64_t *s) { return lo + (hi - lo) * orc_next_f32(s); }

/* nbody.rs:66-71 */
static void uniform_sample_disk(float *x, float *y)
{
    float r = sqrtf(*x);
    float theta = 2.0f * 3.14159274f * (*y); /* 2.0 * consts::PI * y, f32 */
    *x = r * cosf(theta);
    *y = r * sinf(theta);
}

/* nbody.rs:39-64 ; returns particle count written (n<=0 -> 0). draw order x,y,vx,vy,m */
int orc_random_disk(orc_particle *p, int n, uint64_t *rng)
{
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        float x = range_sample(0.0f, 1.0f, rng);  /* :52 */
        float y = range_sample(0.0f, 1.0f, rng);  /* :53 */
        uniform_sample_disk(&x, &y);              /* :54 */
        x *= 23.0f;                               /* :55 */
        y *= 23.0f;
        p[cnt].px = x;
        p[cnt].py = y;
        p[cnt].vx = range_sample(-3.5f, 3.5f, rng); /* :60 */
        p[cnt].vy = range_sample(-3.5f, 3.5f, rng); /* :61 */
        p[cnt].m = range_sample(0.1f, 1.5f, rng);   /* :62 */
        cnt++;
    }
    return cnt;
}

/* nbody.rs:73-104 ; the sun is always pushed, then n-1 planets (i32 arithmetic: n<=1 -> sun only).
 * returns particle count written (= max(n,1)); caller must size p for that. */
int orc_stable_orbits(orc_particle *p, int n, float rmin, float rmax, uint64_t *rng)
{
    const float sun_mass = 1000.0f, planet_mass = 1.0f, g = 1.0f;
    float speed = sqrtf(g * sun_mass);            /* :88 */
    p[0] = (orc_particle){0.0f, 0.0f, 0.0f, 0.0f, sun_mass}; /* :93 */
    int cnt = 1;
    for (int i = 0; i < n - 1; i++) {             /* :95 */
        float r = (rmax - rmin) * range_sample(0.0f, 1.0f, rng) + rmin; /* :96 */
        float theta = 2.0f * 3.14159274f * range_sample(0.0f, 1.0f, rng); /* :97 */
        p[cnt].px = r * cosf(theta);              /* :98 */
        p[cnt].py = r * sinf(theta);
        p[cnt].vx = -speed * sinf(theta);         /* :100 */
        p[cnt].vy = speed * cosf(theta);          /* :101 */
        p[cnt].m = planet_mass;
        cnt++;
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* nb_draw: nbody.rs:482-617                                                                   */

/* Rust `as i32` / `as u32` from f32: truncate toward zero, saturating, NaN -> 0 */
static int32_t f32_as_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
static uint32_t f32_as_u32(float v)
{
    if (v != v || v <= 0.0f) return 0;
    if (v >= 4294967296.0f) return UINT32_MAX;
    return (uint32_t)v;
}

/* nbody.rs:585-593 */
uint32_t orc_rgb_to_abgr32(uint8_t r8, uint8_t g8, uint8_t b8, float factor)
{
    uint32_t r = f32_as_u32((float)r8 * factor);
    uint32_t g = f32_as_u32((float)g8 * factor);
    uint32_t b = f32_as_u32((float)b8 * factor);
    return ((r > 255 ? 255 : r) << 0) | ((b > 255 ? 255 : b) << 16) | ((g > 255 ? 255 : g) << 8);
}

/* nbody.rs:595-617 per-channel saturating add */
uint32_t orc_add_abgr32(uint32_t c1, uint32_t c2)
{
    uint32_t a1 = (c1 & 0xFF000000u) >> 24, b1 = (c1 & 0x00FF0000u) >> 16, g1 = (c1 & 0x0000FF00u) >> 8,
             r1 = (c1 & 0x000000FFu);
    uint32_t a2 = (c2 & 0xFF000000u) >> 24, b2 = (c2 & 0x00FF0000u) >> 16, g2 = (c2 & 0x0000FF00u) >> 8,
             r2 = (c2 & 0x000000FFu);
    uint32_t ar = a1 + a2 < 255 ? a1 + a2 : 255;
    uint32_t gr = g1 + g2 < 255 ? g1 + g2 : 255;
    uint32_t br = b1 + b2 < 255 ? b1 + b2 : 255;
    uint32_t rr = r1 + r2 < 255 ? r1 + r2 : 255;
    return (ar << 24) | (br << 16) | (gr << 8) | rr;
}

/* nbody.rs:482-583 (buffer=false path). fb is caller-owned w*h u32, cleared here. */
void orc_draw(const orc_particle *p, int n, int32_t w, int32_t h, uint32_t *fb)
{
    static const int dir[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1}}; /* :543-552 */
    memset(fb, 0, sizeof(uint32_t) * (size_t)(w * h));  /* :490 */
    float aspect = (float)h / (float)w;                 /* :494 */
    float x1 = VP_ORG_X - VP_WDH / 2.0f;                /* :497 */
    float y1 = (VP_ORG_Y - VP_WDH / 2.0f) * aspect;     /* :498 */
    float x2 = VP_ORG_X + VP_WDH / 2.0f;
    float y2 = (VP_ORG_Y + VP_WDH / 2.0f) * aspect;
    float vpw = x2 - x1, vph = y2 - y1;                 /* :503-504 */
    float scalex = (1.0f / vpw) * (float)w;             /* :505 */
    float scaley = (1.0f / vph) * (float)h;             /* :506 */
    uint32_t col_body = orc_rgb_to_abgr32(255, 215, 130, 0.3f);  /* :520 */
    uint32_t col_tail = orc_rgb_to_abgr32(255, 215, 130, 0.25f); /* :521 */
    for (int k = 0; k < n; k++) {
        float x = (p[k].px - x1) * scalex;              /* :525 */
        float y = (p[k].py - y1) * scaley;              /* :526 */
        for (int i = 0; i < 2; i++) {
            int32_t xo, yo;
            uint32_t col;
            if (i == 0) {
                xo = f32_as_i32(x); yo = f32_as_i32(y); col = col_body;  /* :536-538 */
            } else {
                float angle = atan2f(p[k].vy, p[k].vx);                  /* :541 */
                int32_t octant = f32_as_i32(8.0f * angle / (2.0f * 3.14159274f) + 8.0f) % 8; /* :542 */
                xo = f32_as_i32(x) - dir[octant][0];                     /* :553 */
                yo = f32_as_i32(y) - dir[octant][1];
                col = col_tail;
            }
            if (xo < 0 || xo >= w || yo < 0 || yo >= h) continue;        /* :559 */
            int32_t idx = xo + yo * w;                                   /* :562 */
            fb[idx] = orc_add_abgr32(fb[idx], col);                      /* :564-565 */
        }
    }
    /* :571-577 centre cross, NOT bounds-checked in the reference (w,h >= 3 assumed) */
    fb[w / 2 + 0 + (h / 2 + 0) * w] = 0x00FF00FFu;
    fb[w / 2 + 1 + (h / 2 + 0) * w] = 0x00FF00FFu;
    fb[w / 2 + 0 + (h / 2 + 1) * w] = 0x00FF00FFu;
    fb[w / 2 - 1 + (h / 2 + 0) * w] = 0x00FF00FFu;
    fb[w / 2 + 0 + (h / 2 - 1) * w] = 0x00FF00FFu;
}

int orc_sizeof_particle(void) { return (int)sizeof(orc_particle); }
