/* bh_union_model.c -- what sharing one Barnes-Hut walk among W Morton-adjacent bodies costs (VERDICT r03 next #1b: "two bodies per
 * lane", 128 bodies per wave-walk), counted EXACTLY on the CPU from the flattened tree: for every run of W consecutive leaves the
 * child-group walk of bh_walk.hip is replayed with the set of bodies that are inside each subtree --
 *     turns         groups loaded by the wave (= dependent scalar loads, stack traffic, loop overhead)
 *     child visits  children the wave evaluates (= the per-child VALU block, executed for all lanes whoever is inside)
 *     lane visits   (body, child) pairs that are really needed (what W separate walks would evaluate)
 * and the depth of every turn (how much of the walk is in the top levels a workgroup could keep in LDS).
 *   gcc -O2 -fopenmp tools/bh_union_model.c -o /tmp/bh_union_model -lm ;  /tmp/bh_union_model nodes.bin n_nodes theta
 * nodes.bin = n_nodes records of 32 bytes {float px, py, m, s; int skip, interior; float q; int pad} (nbx_bh_flat_dump). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float px, py, m, s; int32_t skip, interior; float q; int32_t pad; } node_t;
typedef struct { double turns, child_visits, lane_visits, turns_by_depth[40], max_stack; } tally_t;

static const node_t *nd;
static int n_nodes;
static float th2;

/* the children of interior node p, visited by the bodies bx/by[0..cnt) (indices into the wave's bodies in `who`) */
static void walk(int p, int depth, const float *bx, const float *by, const int *who, int cnt, tally_t *t)
{
    t->turns += 1.0;
    t->turns_by_depth[depth < 39 ? depth : 39] += 1.0;
    int open_who[256];
    for (int c = p + 1; c < nd[p].skip; c = nd[c].skip) {
        const node_t *q = &nd[c];
        t->child_visits += 1.0;
        t->lane_visits += cnt;
        int no = 0;
        for (int k = 0; k < cnt; k++) {
            const float dx = q->px - bx[who[k]], dy = q->py - by[who[k]], d2 = dy * dy + dx * dx;
            if (!(q->q < th2 * d2)) open_who[no++] = who[k];
        }
        if (no && q->interior) walk(c, depth + 1, bx, by, open_who, no, t);
    }
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    n_nodes = atoi(argv[2]);
    const float theta = (float)atof(argv[3]);
    th2 = theta * theta;
    node_t *buf = (node_t *)malloc(sizeof(node_t) * (size_t)n_nodes);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(buf, sizeof(node_t), (size_t)n_nodes, f) != (size_t)n_nodes) return 3;
    fclose(f);
    nd = buf;
    int *leaf = (int *)malloc(sizeof(int) * (size_t)n_nodes);
    int nl = 0;
    for (int i = 0; i < n_nodes; i++)
        if (!nd[i].interior) leaf[nl++] = i;
    const int widths[] = {1, 16, 32, 64, 128, 256};
    printf("{\"nodes\": %d, \"leaves\": %d, \"theta\": %g, \"by_bodies_per_walk\": [\n", n_nodes, nl, theta);
    for (int wi = 0; wi < 6; wi++) {
        const int W = widths[wi];
        const int waves = (nl + W - 1) / W;
        tally_t tot;
        memset(&tot, 0, sizeof tot);
#pragma omp parallel
        {
            tally_t t;
            memset(&t, 0, sizeof t);
#pragma omp for schedule(dynamic, 64)
            for (int w = 0; w < waves; w++) {
                float bx[256], by[256];
                int who[256];
                const int a = w * W, cnt = (a + W <= nl ? W : nl - a);
                for (int k = 0; k < cnt; k++) { bx[k] = nd[leaf[a + k]].px; by[k] = nd[leaf[a + k]].py; who[k] = k; }
                /* the root is the only child of group 0 */
                t.turns += 1.0; t.turns_by_depth[0] += 1.0; t.child_visits += 1.0; t.lane_visits += cnt;
                int open_who[256], no = 0;
                for (int k = 0; k < cnt; k++) {
                    const float dx = nd[0].px - bx[k], dy = nd[0].py - by[k], d2 = dy * dy + dx * dx;
                    if (!(nd[0].q < th2 * d2)) open_who[no++] = k;
                }
                if (no && nd[0].interior) walk(0, 1, bx, by, open_who, no, &t);
            }
#pragma omp critical
            {
                tot.turns += t.turns; tot.child_visits += t.child_visits; tot.lane_visits += t.lane_visits;
                for (int d = 0; d < 40; d++) tot.turns_by_depth[d] += t.turns_by_depth[d];
            }
        }
        double top5 = 0, top8 = 0;
        for (int d = 0; d <= 5; d++) top5 += tot.turns_by_depth[d];
        for (int d = 0; d <= 8; d++) top8 += tot.turns_by_depth[d];
        printf("  {\"bodies_per_walk\": %d, \"walks\": %d, \"turns_per_walk\": %.1f, \"child_visits_per_walk\": %.1f, \"children_per_turn\": %.3f,\n"
               "   \"turns_per_body\": %.2f, \"child_visits_per_body\": %.2f, \"needed_child_visits_per_body\": %.1f, \"lanes_inside_per_child_visit\": %.3f,\n"
               "   \"share_of_turns_at_depth_le_5\": %.4f, \"share_of_turns_at_depth_le_8\": %.4f}%s\n",
               W, waves, tot.turns / waves, tot.child_visits / waves, tot.child_visits / tot.turns, tot.turns / nl, tot.child_visits / nl,
               tot.lane_visits / nl, tot.lane_visits / (tot.child_visits * W), top5 / tot.turns, top8 / tot.turns, wi < 5 ? "," : "");
    }
    printf("]}\n");
    return 0;
}
