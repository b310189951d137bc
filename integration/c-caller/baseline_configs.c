/* baseline_configs.c -- a plain C host behind the level-2 C ABI (include/nbody_mi355x.h) running BASELINE.json's
 * configurations from the library's own workload generators (nbx_plummer_sphere / nbx_two_galaxies: SURVEY.md 8(d)),
 * i.e. what a Rust or Haskell host would do to reproduce the benchmark inputs without Python.
 *
 *   gcc -O2 -I<repo>/include baseline_configs.c -o baseline_configs -L<repo>/rust-exp_amd/lib -lnbody_mi355x \
 *       -Wl,-rpath,<repo>/rust-exp_amd/lib
 *   ./baseline_configs generate          # host only: prints an FNV-1a checksum of every generated state
 *   ./baseline_configs run [steps]       # needs an MI355X: steps every configuration, prints interactions/s
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "nbody_mi355x.h"

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static uint64_t fnv_state(nbx_engine *e)
{
    const int32_t n = nbx_num_particles(e);
    float *buf = (float *)malloc(sizeof(float) * 7 * (size_t)(n > 0 ? n : 1));
    uint64_t sum = 1469598103934665603ull;
    if (!buf) return 0;
    if (nbx_get_particles3(e, n, buf, buf + n, buf + 2 * (size_t)n, buf + 3 * (size_t)n, buf + 4 * (size_t)n, buf + 5 * (size_t)n,
                           buf + 6 * (size_t)n) < 0) {
        free(buf);
        return 0;
    }
    const uint32_t *w = (const uint32_t *)buf;
    for (size_t i = 0; i < 7 * (size_t)n; i++) sum = (sum ^ w[i]) * 1099511628211ull;
    free(buf);
    return sum;
}

struct config {
    const char *name;
    int32_t n;
    int kind;        /* 0 stable_orbits (seeded preset), 1 plummer, 2 two galaxies */
    int dim;
    float theta;     /* 0 = brute force */
    int source_bits;
};

static const struct config CONFIGS[] = {
    {"#1 stable_orbits 1024 brute force", 1024, 0, 2, 0.0f, 32},
    {"#2 plummer 65536 brute force", 65536, 1, 3, 0.0f, 32},
    {"#3 plummer 262144 brute force", 262144, 1, 3, 0.0f, 32},
    {"#4 plummer 1048576 barnes-hut 0.5", 1048576, 1, 2, 0.5f, 32},
    {"#5 two galaxies 524288 fp16 sources", 524288, 2, 2, 0.0f, 16},
};

static int generate(nbx_engine *e, const struct config *c)
{
    if (c->kind == 0) {
        if (nbx_seed(e, 1) != NBX_OK) return -1;
        return nbx_stable_orbits(e, c->n, 0.5f, 30.0f);
    }
    if (c->kind == 1) return nbx_plummer_sphere(e, c->n, NBX_SEED_PLUMMER, c->dim);
    return nbx_two_galaxies(e, c->n, NBX_SEED_TWO_GALAXIES);
}

int main(int argc, char **argv)
{
    const int run = argc > 1 && strcmp(argv[1], "run") == 0;
    const int steps = argc > 2 ? atoi(argv[2]) : 5;
    for (size_t k = 0; k < sizeof CONFIGS / sizeof CONFIGS[0]; k++) {
        const struct config *c = &CONFIGS[k];
        nbx_engine *e = NULL;
        if (nbx_create(&e, 0) != NBX_OK || generate(e, c) != NBX_OK) {
            fprintf(stderr, "%s: %s\n", c->name, nbx_last_error());
            return 1;
        }
        printf("%-40s bodies %8d fnv %016llx", c->name, (int)nbx_num_particles(e), (unsigned long long)fnv_state(e));
        if (run) {
            if (nbx_set_option(e, NBX_OPT_SOURCE_PRECISION, c->source_bits) != NBX_OK) return 1;
            int rc = c->theta == 0.0f ? nbx_step_brute_force(e, 0.01f) : nbx_step_barnes_hut(e, c->theta, 0.01f, 1); /* warm-up */
            if (rc == NBX_OK) rc = nbx_synchronize(e);
            const double t0 = now_s();
            for (int s = 0; s < steps && rc == NBX_OK; s++)
                rc = c->theta == 0.0f ? nbx_step_brute_force(e, 0.01f) : nbx_step_barnes_hut(e, c->theta, 0.01f, 1);
            if (rc == NBX_OK) rc = nbx_synchronize(e);
            const double dt = now_s() - t0;
            if (rc != NBX_OK) {
                fprintf(stderr, "\n%s: %s\n", c->name, nbx_last_error());
                return 1;
            }
            const double n = (double)c->n;
            if (c->theta == 0.0f)
                printf("  %8.3f ms/step  %.3e interactions/s", 1e3 * dt / steps, n * (n - 1.0) * steps / dt);
            else
                printf("  %8.3f ms/step  %.3e body-steps/s", 1e3 * dt / steps, n * steps / dt);
        }
        printf("\n");
        nbx_destroy(e);
    }
    return 0;
}
