// bh_gate.h -- device-side gate shared by the Barnes-Hut kernels of bh_eval.hip and bh_walk.hip (internal)
#pragma once
#include "kernels.h"

namespace nbx {

// Device-side gate of a step enqueued BEHIND a device tree build whose outcome the host has not read yet (engine.cpp,
// speculative step): counters = the build's {node count, left-behind bodies, queued folds}; the walk and the kick-drift run
// only if the build produced a usable tree, exactly the test device_tree_build_end makes on the host -- otherwise they leave the
// state untouched and the host redoes the step on the host tree.  counters == nullptr: no gate (n_nodes comes from the host).
// counters[kTreePoisonWord]: an EARLIER gated step was refused and the host has not redone it yet -- nothing may run on the
// state until it has.  The kick-drift of a refused step raises it (mark = true).
struct BuildGate {
    int* counters;
    int node_cap, crowd_limit, queue_limit;
    int* host_out;    // pinned: the kick-drift's first thread hands the counters to the host (no copy command behind the build)
};
__device__ __forceinline__ bool gate_open(const BuildGate g, int& n_nodes, const bool mark = false)
{
    if (!g.counters) return true;
    if (mark && g.host_out) {
        g.host_out[0] = g.counters[0]; g.host_out[1] = g.counters[1]; g.host_out[2] = g.counters[2];
        g.host_out[5] = g.counters[5];   // why the build refused, if it did (bh_build.hip kWhy..)
        g.host_out[6] = g.counters[6]; g.host_out[7] = g.counters[7];   // the chain replay's tallies (bh_build.hip 3b)
        __threadfence_system();
    }
    if (g.counters[kTreePoisonWord] != 0) return false;
    const int nn = g.counters[0];
    if (nn > g.node_cap || g.counters[1] > g.crowd_limit || g.counters[2] > g.queue_limit) {
        if (mark) g.counters[kTreePoisonWord] = 1;
        return false;
    }
    n_nodes = nn;
    return true;
}

}  // namespace nbx
