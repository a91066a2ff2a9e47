// emdr2_amd/csrc/ops_timing.hip -- see ops_timing.h
#include "../../include/emdr2_ops.h"
#include "ops_timing.h"
#include <vector>

namespace {
struct Rec { hipEvent_t a, b; int kind; double flops; };
struct State {
    bool enabled = false;
    std::vector<Rec> pool;      // events are created once and re-used across collects
    size_t used = 0;
} g;
const size_t MAX_RECS = 1u << 16;
}

int ops_time_begin(int kind, double flops, hipStream_t stream)
{
    if (!g.enabled || g.used >= MAX_RECS) return -1;
    if (g.used == g.pool.size()) {
        Rec r;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
        g.pool.push_back(r);
    }
    Rec &r = g.pool[g.used];
    r.kind = kind; r.flops = flops;
    if (hipEventRecord(r.a, stream) != hipSuccess) return -1;
    return (int)g.used++;
}

void ops_time_end(int slot, hipStream_t stream) { (void)hipEventRecord(g.pool[slot].b, stream); }

extern "C" int emdr2_ops_set_timing(int enabled)
{
    g.enabled = enabled != 0;
    g.used = 0;
    return 0;
}

extern "C" int emdr2_ops_timing_collect(double *ms, double *flops, int64_t *launches, int kinds)
{
    if (!ms || !flops || !launches || kinds < OPS_KINDS) return -1;
    for (int k = 0; k < kinds; ++k) { ms[k] = 0; flops[k] = 0; launches[k] = 0; }
    for (size_t i = 0; i < g.used; ++i) {
        Rec &r = g.pool[i];
        if (hipEventSynchronize(r.b) != hipSuccess) return -3;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return -3;
        ms[r.kind] += t; flops[r.kind] += r.flops; launches[r.kind] += 1;
    }
    g.used = 0;
    return 0;
}
