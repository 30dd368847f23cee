// rt_internal.h -- helpers shared by the host translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rt_engine.h"
#include "rt_device.h"

void rt_set_error(const char *fmt, ...);
int rt_hip_fail(hipError_t e, const char *expr, const char *file, int line);

// For int-returning entry points: report, never exit.
#define RT_HIP(expr)                                                        \
    do {                                                                    \
        hipError_t rt_e_ = (expr);                                          \
        if (rt_e_ != hipSuccess) return rt_hip_fail(rt_e_, #expr, __FILE__, __LINE__); \
    } while (0)

struct rt_scene;

// which instantiation of the frame kernel renders a frame (rt_kernels.hip: TW, CULL, MODE, TABLDS, FEAT)
struct RtKernelChoice {
    int tile, cull, mode, table_lds, feat;
};

int rt_scene_set_spheres_async(rt_scene *s, const rt_sphere *host_spheres, int n, hipStream_t stream);
// per-light tables, raygen tables, RtFrameAux: everything but the eye cones (may wait for frames in flight)
int rt_scene_prepare_static(rt_scene *s, const rt_frame_desc *fd, hipStream_t stream);
// pure host computation of the by-value uniforms; `cones`: the eye-cone table the frame reads, or null
int rt_build_frame_consts(const rt_scene *s, const rt_frame_desc *fd, const float4 *cones, RtFrameConsts *fc);
int rt_frame_kernel_choice(const rt_scene *s, const rt_frame_desc *fd, RtKernelChoice *kc);
void rt_ray_origin(const rt_frame_desc *fd, float org[3]);
// frames in flight (rt_engine.cpp): host wait for all of them; bookkeeping after a launch
int rt_scene_quiesce(rt_scene *s);
int rt_scene_note_launch(rt_scene *s, hipStream_t stream, int cone_slot);
// what a graph node needs from the scene
const float4 *rt_scene_sphere_table(const rt_scene *s);
int rt_scene_sphere_count(const rt_scene *s);
unsigned long long rt_scene_epoch(const rt_scene *s);
bool rt_scene_wants_eye_cones(const rt_scene *s, const float org[3]);
int rt_scene_tile_order_mode(const rt_scene *s);   // rt_scene_set_tile_order
int rt_scene_build_eye_cones_host(rt_scene *s, const float org[3], float4 *buf, hipStream_t stream);

// launchers of rt_kernels.hip
extern "C" hipError_t rt_dev_prepare(void);
extern "C" hipError_t rt_dev_trace_config(const RtFrameConsts *fc, int tile_w, int cull, int mode, int table_in_lds, int feat,
                                          const void **func, dim3 *grid, dim3 *block, unsigned *lds_bytes);
extern "C" hipError_t rt_dev_launch_trace(const RtFrameConsts *fc, const float4 *spheres, int tile_w, int cull, int mode,
                                          int table_in_lds, int feat, hipStream_t stream);
