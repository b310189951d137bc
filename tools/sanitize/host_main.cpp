// Host-side sanitizer harness: threaded quadtree build (+ pre-flatten), pipelined flatten, threaded draw, straight on
// nbx::QuadTree / nbx::draw_particles (no device).  Built and run by tests/test_host_sanitizers.py with
// -fsanitize=thread and -fsanitize=address,undefined.
#include <cstdio>
#include <random>
#include <thread>
#include <vector>
#include "host_ops.h"
int main()
{
    const int n = 120000;
    std::mt19937 rng(5);
    std::normal_distribution<float> g(0.f, 8.f);
    std::uniform_real_distribution<float> um(0.5f, 2.f);
    std::vector<float> px(n), py(n), vx(n), vy(n), m(n);
    for (int i = 0; i < n; i++) { px[i] = g(rng); py[i] = g(rng); vx[i] = g(rng); vy[i] = g(rng); m[i] = um(rng); }
    nbx::QuadTree t;
    for (int rep = 0; rep < 3; rep++) {
        int rc = t.build(px.data(), py.data(), m.data(), n, true);
        nbx::QuadTree::FlatPlan plan;
        size_t cnt = t.flatten_prepare(plan);
        std::vector<nbx::BhNode> out(cnt);
        size_t sent = 0;
        t.flatten_write(plan, out.data(), [&](size_t a, size_t b) { sent += b - a; });
        std::printf("rc=%d nodes=%zu sent=%zu skip0=%d\n", rc, cnt, sent, out[0].skip);
    }
    {   // two callers at once: both use the one worker pool
        auto caller = [&](int seed) {
            std::vector<float> qx(px), qy(py);
            for (int i = 0; i < n; i += 3) { qx[i] += 0.001f * seed; qy[i] -= 0.002f * seed; }
            nbx::QuadTree u;
            const int rc = u.build(qx.data(), qy.data(), m.data(), n, true);
            std::vector<nbx::BhNode> flat;
            u.flatten(flat);
            std::printf("caller %d rc=%d flat=%zu\n", seed, rc, flat.size());
        };
        std::thread a(caller, 1), b(caller, 2);
        a.join(); b.join();
    }
    std::vector<uint32_t> fb(512 * 512);
    nbx::draw_particles(px.data(), py.data(), vx.data(), vy.data(), n, 512, 512, fb.data());
    size_t lit = 0; for (auto v : fb) lit += v != 0;
    std::printf("lit=%zu\n", lit);
    return 0;
}
