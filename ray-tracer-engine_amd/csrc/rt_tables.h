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
// Per sphere S of the table and one light: the entries a shadow ray from S's surface towards the light can hit
// (see RtFrameAux::cand_hdr). hdr: n records; ent: the concatenated lists.
#include <vector>
struct RtCandHdr;
void rt_build_occluder_lists(const float4 *tab, int n, const float lpos[3], std::vector<RtCandHdr> &hdr, std::vector<float4> &ent);
// Slope of the beam the frame kernel gives a group of shadow rays that start at `start` (binary64 restatement of the
// kernel's bound; NaN when it has none).
double rt_light_beam_slope(const double lpos[3], const double start[3]);
// ... and its supremum over every start in the ball B(c, r0) (the slope a group of pixels on that sphere uses); -1: none
double rt_sphere_beam_slope(const double lpos[3], const double c[3], double r0);
double rt_beam_sine_at_start(const double lpos[3], const double start[3], double *sigma, double *frob, double m9[9]);
// The lists built on the device (one wave per sphere): hdr: n records, ent: n * RT_CAND_CAP entries (a slot per sphere).
hipError_t rt_occluder_lists_launch(const float4 *tab, int n, const float lpos[3], RtCandHdr *hdr, float4 *ent, hipStream_t stream);
void rt_build_eye_cones_host(const float4 *tab, int n, const float org[3], float4 *sorted, float4 *blocks, int *orig);

size_t rt_eye_cones_size(int n);   // float4 units: [n_pad entries][2 per block][n_pad ints]
// Build the table on the device, on `stream`, with one workgroup of `threads` (a multiple of 64, <= 1024)
// (tab: the list-order table in device memory).
hipError_t rt_eye_cones_launch(const float4 *tab, int n, const float org[3], float4 *out, int threads, hipStream_t stream);
// The same launch as a graph kernel node: function, geometry and dynamic LDS; the arguments are
// (const float4 *tab, int n, float ox, float oy, float oz, float4 *out).
void rt_eye_cones_kernel_config(int n, int threads, const void **func, dim3 *grid, dim3 *block, unsigned *lds_bytes);

// Order of the tiles of a launch (RtFrameConsts::tile_perm): blocks of RT_TILE_ORDER_BLOCK x RT_TILE_ORDER_BLOCK tiles,
// the block with the longest tile first (cost[tile]: wave durations of the previous launch, shader clocks; 0 = never
// rendered), tiles row-major inside a block; perm[] = (tile_y << 16) | tile_x. Three small kernels on `stream` (the
// blocks' longest tiles into key[]; the blocks sorted, where each starts into start[]; a thread per tile writes perm[]).
// Always a permutation of the tiles, whatever cost[] holds. At most RT_TILE_ORDER_MAX_BLOCKS blocks; key[], start[]:
// one unsigned per block.
#define RT_TILE_ORDER_BLOCK 16
#define RT_TILE_ORDER_MAX_BLOCKS 4096
hipError_t rt_tile_order_launch(const unsigned *cost, unsigned *key, unsigned *start, unsigned *perm, int tiles_x, int tiles_y,
                                hipStream_t stream);
void rt_tile_order_kernel_configs(int tiles_x, int tiles_y, const void *func[3], dim3 grid[3], dim3 block[3]);
