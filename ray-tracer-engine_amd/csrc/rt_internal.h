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
int rt_scene_set_spheres_async(rt_scene *s, const rt_sphere *host_spheres, int n, hipStream_t stream);
int rt_build_frame_consts(const rt_scene *s, const rt_frame_desc *fd, RtFrameConsts *fc);
int rt_scene_prepare_lights(rt_scene *s, hipStream_t stream);
int rt_scene_prepare_eye(rt_scene *s, const float org[3], hipStream_t stream);
void rt_ray_origin(const rt_frame_desc *fd, float org[3]);
