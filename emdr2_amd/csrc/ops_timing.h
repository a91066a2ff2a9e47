// emdr2_amd/csrc/ops_timing.h -- optional per-launch hipEvent timing of the transformer kernels (include/emdr2_ops.h: emdr2_ops_set_timing /
// emdr2_ops_timing_collect).  Events are recorded on the stream the kernel is launched on, so bench.py's roofline numbers for the GEMMs are
// measured live inside the timed region (torch.cuda.Event would only see torch's current stream).  Off by default: one predictable branch.
#ifndef EMDR2_OPS_TIMING_H
#define EMDR2_OPS_TIMING_H
#include <hip/hip_runtime.h>

enum { OPS_GEMM_NT = 0, OPS_GEMM_TN = 1, OPS_ATTN_FWD = 2, OPS_ATTN_BWD = 3, OPS_KINDS = 4 };

int ops_time_begin(int kind, double flops, hipStream_t stream);   // returns a slot (>= 0) when timing is on, -1 otherwise
void ops_time_end(int slot, hipStream_t stream);

struct OpsTimer {                                                   // scope guard around one launch
    int slot;
    hipStream_t stream;
    OpsTimer(int kind, double flops, hipStream_t s) : slot(ops_time_begin(kind, flops, s)), stream(s) {}
    ~OpsTimer() { if (slot >= 0) ops_time_end(slot, stream); }
};
#endif
