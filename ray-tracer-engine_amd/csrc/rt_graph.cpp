// rt_graph.cpp -- hipGraph-captured frame loop (BASELINE config C4: 4-spp
// progressive accumulation, one graph replay per frame). Replaces the
// reference's per-frame malloc -> H2D -> launch -> sync -> free sequence
// (/root/reference/kernel.cu:1762-1792) with a graph instantiated once.
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/rt_engine.h"
#include "rt_internal.h"

extern "C" hipError_t rt_dev_prepare(void);

struct rt_frame_graph {
    rt_scene *scene = nullptr;
    rt_frame_desc fd;
    int passes = 1;
    uint32_t *host_pixels = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

// Record the frame's work on `stream`: `passes` progressive sample passes (pass
// p takes sample p of `passes`, adds it into the float4 buffer, the last one
// resolves into the packed frame) and the optional copy to the present buffer.
static int record_frame(rt_frame_graph *g, hipStream_t stream)
{
    for (int p = 0; p < g->passes; ++p) {
        rt_frame_desc fd = g->fd;
        fd.opts.spp = 1;
        fd.opts.sample_base = p;
        fd.opts.sample_total = g->passes;
        fd.opts.accumulate = p > 0 ? 1 : 0;
        fd.opts.resolve = (p == g->passes - 1) ? 0 : -1;
        const int rc = rt_scene_render(g->scene, &fd, stream);
        if (rc != RT_OK) return rc;
    }
    if (g->host_pixels && g->fd.pixels) {
        int y0 = g->fd.opts.y0, y1 = g->fd.opts.y1;
        if (y0 == 0 && y1 == 0) y1 = g->fd.height;
        RT_HIP(hipMemcpyAsync(g->host_pixels, g->fd.pixels, sizeof(uint32_t) * (size_t)g->fd.width * (size_t)(y1 - y0),
                              hipMemcpyDeviceToHost, stream));
    }
    return RT_OK;
}

static int capture(rt_frame_graph *g, hipStream_t stream, hipGraph_t *out)
{
    {   // per-light tables are (re)built outside the capture; the captured launches only read them
        int prc = rt_scene_prepare_lights(g->scene, stream);
        if (prc != RT_OK) return prc;
        float org[3];
        rt_ray_origin(&g->fd, org);
        prc = rt_scene_prepare_eye(g->scene, org, stream);
        if (prc != RT_OK) return prc;
        RT_HIP(hipStreamSynchronize(stream));
    }
    RT_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    const int rc = record_frame(g, stream);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(stream, &graph);
    if (rc != RT_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
    }
    RT_HIP(e);
    *out = graph;
    return RT_OK;
}

extern "C" rt_frame_graph *rt_graph_capture(rt_scene *s, const rt_frame_desc *fd, int passes,
                                            uint32_t *host_pixels, void *stream)
{
    if (!s || !fd || passes < 1 || passes > RT_MAX_SPP || !stream) {
        rt_set_error("rt_graph_capture: invalid argument (a non-default stream is required)");
        return nullptr;
    }
    if (passes > 1 && !fd->opts.rgba) {
        rt_set_error("rt_graph_capture: progressive passes need opts.rgba (float4 accumulation buffer)");
        return nullptr;
    }
    if (rt_dev_prepare() != hipSuccess) {
        rt_set_error("rt_graph_capture: kernel image not loadable on this device");
        return nullptr;
    }
    rt_frame_graph *g = new rt_frame_graph();
    g->scene = s;
    g->fd = *fd;
    g->passes = passes;
    g->host_pixels = host_pixels;
    if (capture(g, (hipStream_t)stream, &g->graph) != RT_OK) {
        delete g;
        return nullptr;
    }
    if (hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0) != hipSuccess) {
        rt_set_error("rt_graph_capture: hipGraphInstantiate failed");
        (void)hipGraphDestroy(g->graph);
        delete g;
        return nullptr;
    }
    return g;
}

extern "C" int rt_graph_launch(rt_frame_graph *g, void *stream)
{
    if (!g || !g->exec) return RT_ERR_INVALID;
    RT_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return RT_OK;
}

// A camera move changes the by-value frame uniforms of every kernel node:
// re-record the frame and update the instantiated graph in place.
extern "C" int rt_graph_set_camera(rt_frame_graph *g, const rt_camera *cam)
{
    if (!g || !cam) return RT_ERR_INVALID;
    g->fd.cam = *cam;
    hipStream_t tmp = nullptr;
    RT_HIP(hipStreamCreateWithFlags(&tmp, hipStreamNonBlocking));
    hipGraph_t fresh = nullptr;
    const int rc = capture(g, tmp, &fresh);
    (void)hipStreamDestroy(tmp);
    if (rc != RT_OK) return rc;
    hipGraphExecUpdateResult res;
    hipGraphNode_t err_node = nullptr;
    if (hipGraphExecUpdate(g->exec, fresh, &err_node, &res) != hipSuccess) {
        (void)hipGetLastError();
        hipGraphExec_t exec = nullptr;
        RT_HIP(hipGraphInstantiate(&exec, fresh, nullptr, nullptr, 0));
        (void)hipGraphExecDestroy(g->exec);
        g->exec = exec;
    }
    (void)hipGraphDestroy(g->graph);
    g->graph = fresh;
    return RT_OK;
}

extern "C" void rt_graph_destroy(rt_frame_graph *g)
{
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}
