/* frame_loop.c -- a plain C caller of the six reference symbols, linked against libnbody_mi355x.so the way the
 * Haskell front-end's `foreign import ccall` links against the Rust staticlib (hs-src/RustNBodyExperiment.hs:101-106).
 * No header of the library is included on purpose: the prototypes below are the reference's FFI declarations
 * (CInt -> int32_t, CFloat -> float, Ptr Word32 -> uint32_t*).
 *
 *   gcc -O2 frame_loop.c -o frame_loop -L<repo>/rust-exp_amd/lib -lnbody_mi355x -Wl,-rpath,<repo>/rust-exp_amd/lib
 *   NB_SEED=1 ./frame_loop [frames]
 *
 * Runs the experiment's frame loop (init hs:42, step + draw hs:50-62, status hs:66-70) and prints a checksum of the
 * last framebuffer so that two runs with the same NB_SEED can be compared.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

void nb_draw(int32_t w, int32_t h, uint32_t *fb);
void nb_step_brute_force(float dt);
void nb_step_barnes_hut(float theta, float dt, int32_t nthreads);
void nb_random_disk(int32_t n);
void nb_stable_orbits(int32_t n, float rmin, float rmax);
int32_t nb_num_particles(void);

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 30;
    const int32_t w = 512, h = 512;
    uint32_t *fb = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)w * (size_t)h);   /* stands in for the mapped PBO */
    if (!fb) return 2;
    nb_stable_orbits(10000, 0.5f, 30.0f);                       /* hs:42 */
    double step_ms = 0.0, draw_ms = 0.0;
    for (int f = 0; f < frames; f++) {
        double t0 = now_ms();
        nb_step_barnes_hut(0.85f, 0.01f, 1);                    /* hs:55-57 (defaults hs:43-47) */
        double t1 = now_ms();
        nb_draw(w, h, fb);                                      /* hs:58-60 */
        double t2 = now_ms();
        step_ms += t1 - t0;
        draw_ms += t2 - t1;
    }
    nb_step_brute_force(0.01f);                                 /* the other step entry point */
    nb_random_disk(2000);                                       /* hs:85-87 re-init keys */
    nb_step_barnes_hut(0.0f, 0.01f, 1);                         /* theta 0 delegates to brute force (nbody.rs:197-200) */
    nb_draw(w, h, fb);
    uint64_t sum = 1469598103934665603ull;                      /* FNV-1a over the framebuffer */
    size_t lit = 0;
    for (size_t i = 0; i < (size_t)w * (size_t)h; i++) {
        sum = (sum ^ fb[i]) * 1099511628211ull;
        lit += fb[i] != 0;
    }
    printf("bodies %d frames %d step %.3f ms draw %.3f ms lit %zu fnv %016llx\n", (int)nb_num_particles(), frames,
           step_ms / frames, draw_ms / frames, lit, (unsigned long long)sum);
    free(fb);
    return 0;
}
