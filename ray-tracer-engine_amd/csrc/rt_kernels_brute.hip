// rt_kernels_brute.hip -- the frame kernel with the reference's loops as written (CULL = false), default tile,
// tables in global memory: what every culling kernel is compared with.
#include "rt_trace.inc"

RtTraceFn rt_trace_fn_brute8(int mode, int feat, int multi)
{
    return multi ? trace_fn_mode_feat<8, false, false, true>(mode, feat) : trace_fn_mode_feat<8, false, false, false>(mode, feat);
}
