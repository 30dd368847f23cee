// rt_kernels_lds.hip -- the frame kernel with the whole sphere table staged in LDS per 256-thread workgroup
// (rt_launch_opts.table_lds; north_star's first design, measured slower: DESIGN.md section 3).
#include "rt_trace.inc"

RtTraceFn rt_trace_fn_lds(int cull, int mode, int feat, int multi)
{
    if (multi) return cull ? trace_fn_mode_feat<8, true, true, true>(mode, feat) : trace_fn_mode_feat<8, false, true, true>(mode, feat);
    return cull ? trace_fn_mode_feat<8, true, true, false>(mode, feat) : trace_fn_mode_feat<8, false, true, false>(mode, feat);
}
