/* bh_lines_model.c -- where the Barnes-Hut walk's memory traffic comes from (VERDICT r02 next #5c), modelled on the CPU from the
 * flattened tree itself: every leaf (= body, in Morton order) walks the tree with the fast walk's decision q < theta^2 d^2; the
 * bodies are cut into 8 contiguous eighths like the kernel's XCD-aware block order; per eighth a bitmap records which node
 * records -- and which 128-byte lines of the node array -- its walks touch.  Output: unique lines per eighth (x 128 B = the least
 * an XCD's L2 must fetch once), visits by tree depth, share of the top levels.
 *   gcc -O2 -fopenmp tools/bh_lines_model.c -o /tmp/bh_lines_model -lm ;  /tmp/bh_lines_model nodes.bin n_nodes theta
 * nodes.bin = n_nodes records of 32 bytes {float px, py, m, s; int skip, interior; float q; int pad} (nbx_bh_flat_dump). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float px, py, m, s; int32_t skip, interior; float q; int32_t pad; } node_t;

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const int n = atoi(argv[2]);
    const float theta = (float)atof(argv[3]);
    node_t *nd = (node_t *)malloc(sizeof(node_t) * (size_t)n);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(nd, sizeof(node_t), (size_t)n, f) != (size_t)n) return 3;
    fclose(f);
    /* depth of every node from the pre-order + skip pointers */
    uint8_t *depth = (uint8_t *)calloc((size_t)n, 1);
    {
        int *end = (int *)malloc(sizeof(int) * 64);
        int sp = 0;
        for (int i = 0; i < n; i++) {
            while (sp > 0 && end[sp - 1] == i) sp--;
            depth[i] = (uint8_t)sp;
            if (nd[i].interior) end[sp++] = nd[i].skip;
        }
        free(end);
    }
    int *leaf = (int *)malloc(sizeof(int) * (size_t)n);
    int nl = 0;
    for (int i = 0; i < n; i++)
        if (!nd[i].interior) leaf[nl++] = i;
    const size_t words = ((size_t)n + 63) / 64;
    uint64_t *seen = (uint64_t *)calloc(words * 8, sizeof(uint64_t));
    double visits_by_depth[64] = {0}, takes_by_depth[64] = {0};
    const float th2 = theta * theta;
    for (int r = 0; r < 8; r++) {
        const int a = (int)((long long)nl * r / 8), b = (int)((long long)nl * (r + 1) / 8);
        uint64_t *bm = seen + words * (size_t)r;
#pragma omp parallel
        {
            double vd[64] = {0}, td[64] = {0};
#pragma omp for schedule(dynamic, 256)
            for (int t = a; t < b; t++) {
                const float x = nd[leaf[t]].px, y = nd[leaf[t]].py;
                int i = 0;
                while (i < n) {
                    const node_t *q = &nd[i];
                    uint64_t bit = 1ull << (i & 63), *w = &bm[i >> 6];
                    if (!(__atomic_load_n(w, __ATOMIC_RELAXED) & bit)) __atomic_fetch_or(w, bit, __ATOMIC_RELAXED);
                    const float dx = q->px - x, dy = q->py - y, d2 = dy * dy + dx * dx;
                    const int take = q->q < th2 * d2;
                    vd[depth[i]] += 1.0;
                    td[depth[i]] += take;
                    i = take ? q->skip : i + 1;
                }
            }
#pragma omp critical
            for (int d = 0; d < 64; d++) { visits_by_depth[d] += vd[d]; takes_by_depth[d] += td[d]; }
        }
    }
    printf("{\"nodes\": %d, \"leaves\": %d, \"theta\": %g, \"node_bytes\": %.0f,\n \"per_eighth\": [", n, nl, theta, 32.0 * n);
    double sum_lines = 0, sum_nodes = 0;
    for (int r = 0; r < 8; r++) {
        const uint64_t *bm = seen + words * (size_t)r;
        long long nodes = 0, lines = 0;
        for (size_t w = 0; w < words; w++) {
            const uint64_t v = bm[w];
            nodes += __builtin_popcountll(v);
            for (int k = 0; k < 16; k++) lines += ((v >> (4 * k)) & 0xF) != 0;   /* 4 records of 32 B per 128-B line */
        }
        sum_lines += (double)lines; sum_nodes += (double)nodes;
        printf("%s{\"unique_nodes\": %lld, \"unique_128B_lines\": %lld}", r ? ", " : "", nodes, lines);
    }
    double tv = 0, top5 = 0, top8 = 0;
    for (int d = 0; d < 64; d++) { tv += visits_by_depth[d]; if (d <= 5) top5 += visits_by_depth[d]; if (d <= 8) top8 += visits_by_depth[d]; }
    printf("],\n \"sum_over_eighths_unique_nodes\": %.0f, \"bytes_if_every_eighth_fetched_its_lines_once\": %.0f, \"bytes_if_fetched_as_exact_records\": %.0f,\n"
           " \"visits_per_body\": %.1f, \"share_of_visits_at_depth_le_5\": %.4f, \"share_of_visits_at_depth_le_8\": %.4f,\n \"visits_by_depth\": [",
           sum_nodes, 128.0 * sum_lines, 32.0 * sum_nodes, tv / nl, top5 / tv, top8 / tv);
    for (int d = 0; d < 32; d++) printf("%s%.0f", d ? ", " : "", visits_by_depth[d]);
    long long top5_nodes = 0;
    for (int i = 0; i < n; i++) top5_nodes += depth[i] <= 5;
    printf("],\n \"nodes_at_depth_le_5\": %lld, \"bytes_at_depth_le_5\": %.0f}\n", top5_nodes, 32.0 * top5_nodes);
    return 0;
}
