// rt_tables.h -- builders of the culling tables (rt_tables.hip); internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// Largest beam slope the eye cones are valid for (RtFrameConsts::cone_kcap): tiles whose own
// beam is wider (tiny resolutions) fall back to the 3-D blocks.
#define RT_CONE_KCAP 0.1
// Longest list (padded to 64) the one-workgroup device builder sorts in LDS (8 B per key).
#define RT_EYE_DEVICE_MAX 8192

void rt_build_sorted_blocks(const float4 *tab, int n, float4 *sorted, float4 *blocks, int *orig);
void rt_build_light_columns(const float4 *tab, int n, const float u[3], float4 *sorted, float4 *blocks);
void rt_build_eye_cones_host(const float4 *tab, int n, const float org[3], float4 *sorted, float4 *blocks, int *orig);

size_t rt_eye_cones_size(int n);   // float4 units: [n_pad entries][2 per block][n_pad ints]
// Build the table on the device, on `stream`, with one workgroup of `threads` (a multiple of 64, <= 1024)
// (tab: the list-order table in device memory).
hipError_t rt_eye_cones_launch(const float4 *tab, int n, const float org[3], float4 *out, int threads, hipStream_t stream);
// The same launch as a graph kernel node: function, geometry and dynamic LDS; the arguments are
// (const float4 *tab, int n, float ox, float oy, float oz, float4 *out).
void rt_eye_cones_kernel_config(int n, int threads, const void **func, dim3 *grid, dim3 *block, unsigned *lds_bytes);

// Order of the tiles of a launch, longest first (RtFrameConsts::tile_perm): a counting sort of `n` tiles by the wave
// durations of an earlier frame (cost[], shader clocks; 0 = never rendered) into perm[] = (tile_y << 16) | tile_x,
// one workgroup on `stream`. Always a permutation of the n tiles, whatever cost[] holds.
hipError_t rt_tile_order_launch(const unsigned *cost, unsigned *perm, int n, int tiles_x, hipStream_t stream);
