// rt_kernels_tiles.hip -- the frame kernel for the tile shapes 16x4, 32x2 and 64x1 (test dimensions: every shape
// renders the same bits; always with the sample loop, tables in global memory).
#include "rt_trace.inc"

template <int TW>
static RtTraceFn tiles_tw(int cull, int mode, int feat)
{
    return cull ? trace_fn_mode_feat<TW, true, false, true>(mode, feat) : trace_fn_mode_feat<TW, false, false, true>(mode, feat);
}

RtTraceFn rt_trace_fn_tiles(int tile_w, int cull, int mode, int feat)
{
    switch (tile_w) {
    case 16: return tiles_tw<16>(cull, mode, feat);
    case 32: return tiles_tw<32>(cull, mode, feat);
    case 64: return tiles_tw<64>(cull, mode, feat);
    default: return nullptr;
    }
}
