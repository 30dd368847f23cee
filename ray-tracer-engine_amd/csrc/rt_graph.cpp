// rt_graph.cpp -- the frame as a hipGraph (BASELINE config C4: 4-spp progressive
// accumulation, one graph replay per frame). Replaces the reference's per-frame
// malloc -> H2D -> launch -> sync -> free sequence (/root/reference/kernel.cu:1762-1792)
// with a graph built once:
//
//   [eye-cone build]  ->  [tile order: keys, sort, expand]  ->  [sample pass 0] -> ... -> [sample pass p-1]  ->  [copy to the present buffer]
//
// (the tile order -- blocks of 16 x 16 tiles, longest tile first, DESIGN.md section 4d -- is sorted at the head of every
// replay from the wave durations the passes of the previous replay recorded into the graph's own arrays)
//
// The nodes are added explicitly (no stream capture), so each kernel node's by-value frame
// uniforms can be replaced in the instantiated graph: a camera move -- the reference moves
// `cam` every frame (kernel.cu:1716-1764) -- is hipGraphExecKernelNodeSetParams on the pass
// nodes and on the build node, whose kernel then rebuilds the graph's OWN eye-cone table on
// the device at the next replay. No re-capture, no synchronisation, and the graph never reads
// a table that a direct render on the same scene has rebuilt for another camera.
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/rt_engine.h"
#include "rt_internal.h"
#include "rt_tables.h"

struct rt_frame_graph {
    rt_scene *scene = nullptr;
    rt_frame_desc fd;
    int passes = 1;         // kernel nodes that render samples: 1 (all samples in one launch) or one per sample
    int samples = 1;        // samples per pixel of the frame
    uint32_t *host_pixels = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // nodes whose parameters follow the camera
    hipGraphNode_t build_node = nullptr;
    hipGraphNode_t pass_node[RT_MAX_SPP] = {};
    bool pass_live[RT_MAX_SPP] = {};
    hipKernelNodeParams pass_params[RT_MAX_SPP];
    hipKernelNodeParams build_params;
    // argument storage the node parameters point at
    RtFrameConsts fc[RT_MAX_SPP];
    const float4 *spheres = nullptr;
    const float4 *build_tab = nullptr;
    int build_n = 0;
    float build_org[3] = {0, 0, 0};
    float4 *cones = nullptr;         // the graph's own eye-cone table
    size_t cones_cap = 0;            // float4 units
    bool cones_on_device = false;    // built by the graph's build node (else by the host at (re)build time)
    unsigned long long epoch = 0;    // rt_scene_epoch() the nodes were built against
    // launch order of the passes' tiles (rt_scene_set_tile_order): [n] durations, [n] order, [nb] block keys, [nb] block starts
    unsigned *order_buf = nullptr;
    size_t order_cap = 0;            // unsigneds
    unsigned *o_cost = nullptr, *o_perm = nullptr, *o_key = nullptr, *o_start = nullptr;
    bool order_on = false;
};

static void release_graph(rt_frame_graph *g)
{
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    g->exec = nullptr;
    g->graph = nullptr;
    g->build_node = nullptr;
    for (int p = 0; p < RT_MAX_SPP; ++p) g->pass_live[p] = false;
}

static rt_frame_desc pass_desc(const rt_frame_graph *g, int p)
{
    rt_frame_desc fd = g->fd;
    if (g->passes == 1) {   // every sample in one launch (the sums are formed in registers, in the same order)
        fd.opts.spp = g->samples;
        fd.opts.sample_base = 0;
        fd.opts.sample_total = g->samples;
        fd.opts.accumulate = 0;
        fd.opts.resolve = 0;
        return fd;
    }
    fd.opts.spp = 1;
    fd.opts.sample_base = p;
    fd.opts.sample_total = g->passes;
    fd.opts.accumulate = p > 0 ? 1 : 0;
    fd.opts.resolve = (p == g->passes - 1) ? 0 : -1;
    return fd;
}

// Fill the argument storage (frame uniforms of every pass, build origin) for the current camera.
static int fill_arguments(rt_frame_graph *g, bool with_cones)
{
    rt_ray_origin(&g->fd, g->build_org);
    for (int p = 0; p < g->passes; ++p) {
        const rt_frame_desc fd = pass_desc(g, p);
        const int rc = rt_build_frame_consts(g->scene, &fd, with_cones ? g->cones : nullptr, &g->fc[p]);
        if (rc != RT_OK) return rc;
        if (g->order_on) {
            g->fc[p].tile_perm = g->o_perm;
            g->fc[p].tile_cost = g->o_cost;
        }
    }
    return RT_OK;
}

// (Re)build graph and executable against the scene as it is now.
static int build_graph(rt_frame_graph *g, hipStream_t stream)
{
    rt_scene *s = g->scene;
    {   // an earlier replay may still be running: it reads the nodes and the cone table rebuilt here
        const int rc = rt_scene_quiesce(s);
        if (rc != RT_OK) return rc;
    }
    release_graph(g);
    {   // sample passes are one-sample launches of one frame: same tables for all of them
        rt_frame_desc fd0 = pass_desc(g, 0);
        const int rc = rt_scene_prepare_static(s, &fd0, stream);
        if (rc != RT_OK) return rc;
    }
    float org[3];
    rt_ray_origin(&g->fd, org);
    const int n = rt_scene_sphere_count(s);
    const bool want_cones = g->fd.opts.cull != 0 && rt_scene_wants_eye_cones(s, org);
    g->cones_on_device = want_cones && ((n + 63) & ~63) <= RT_EYE_DEVICE_MAX;
    if (want_cones) {
        const size_t total = rt_eye_cones_size(n);
        if (total > g->cones_cap) {
            if (g->cones) RT_HIP(hipFree(g->cones));
            g->cones = nullptr;
            g->cones_cap = 0;
            RT_HIP(hipMalloc((void **)&g->cones, sizeof(float4) * total));
            g->cones_cap = total;
        }
        if (!g->cones_on_device) {   // list too long for the device builder: host build, blocking upload
            const int rc = rt_scene_build_eye_cones_host(s, org, g->cones, stream);
            if (rc != RT_OK) return rc;
        }
    }
    // the passes' launch order: all passes render the same rows with the same tile shape
    int tiles_x = 0, tiles_y = 0;
    g->order_on = false;
    {
        const rt_frame_desc fd0 = pass_desc(g, 0);
        RtKernelChoice kc0;
        RtFrameConsts fc0;
        int rc0 = rt_frame_kernel_choice(s, &fd0, &kc0);
        if (rc0 == RT_OK) rc0 = rt_build_frame_consts(s, &fd0, nullptr, &fc0);
        if (rc0 != RT_OK) return rc0;
        tiles_x = (fc0.width + kc0.tile - 1) / kc0.tile;
        tiles_y = (fc0.local_rows + 64 / kc0.tile - 1) / (64 / kc0.tile);
        const long long nb = (long long)((tiles_x + RT_TILE_ORDER_BLOCK - 1) / RT_TILE_ORDER_BLOCK) * ((tiles_y + RT_TILE_ORDER_BLOCK - 1) / RT_TILE_ORDER_BLOCK);
        if (rt_scene_tile_order_mode(s) != 0 && !kc0.table_lds && fc0.local_rows > 0 && tiles_x <= 0xffff && tiles_y <= 0xffff &&
            nb <= RT_TILE_ORDER_MAX_BLOCKS) {
            const size_t n = (size_t)tiles_x * (size_t)tiles_y, need = 2 * n + 2 * (size_t)nb;
            if (need > g->order_cap) {
                if (g->order_buf) RT_HIP(hipFree(g->order_buf));
                g->order_buf = nullptr;
                g->order_cap = 0;
                RT_HIP(hipMalloc((void **)&g->order_buf, sizeof(unsigned) * need));
                g->order_cap = need;
            }
            g->o_cost = g->order_buf;
            g->o_perm = g->o_cost + n;
            g->o_key = g->o_perm + n;
            g->o_start = g->o_key + nb;
            RT_HIP(hipMemset(g->o_cost, 0, sizeof(unsigned) * n));   // no durations yet: the first replay sorts zeros (any order)
            g->order_on = true;
        }
    }
    int rc = fill_arguments(g, want_cones);
    if (rc != RT_OK) return rc;
    g->spheres = rt_scene_sphere_table(s);

    RT_HIP(hipGraphCreate(&g->graph, 0));
    hipGraphNode_t prev = nullptr;
    if (g->cones_on_device) {
        g->build_tab = g->spheres;
        g->build_n = n;
        memset(&g->build_params, 0, sizeof g->build_params);
        const void *func;
        unsigned lds;
        rt_eye_cones_kernel_config(n, 1024, &func, &g->build_params.gridDim, &g->build_params.blockDim, &lds);
        g->build_params.func = const_cast<void *>(func);
        g->build_params.sharedMemBytes = lds;
        void *args[] = {&g->build_tab, &g->build_n, &g->build_org[0], &g->build_org[1], &g->build_org[2], &g->cones};
        g->build_params.kernelParams = args;
        RT_HIP(hipGraphAddKernelNode(&g->build_node, g->graph, nullptr, 0, &g->build_params));
        g->build_params.kernelParams = nullptr;   // `args` is a local: re-pointed on every update
        prev = g->build_node;
    }
    if (g->order_on) {   // keys -> sort -> expand, from the durations of the previous replay
        const void *func[3];
        dim3 grid[3], block[3];
        rt_tile_order_kernel_configs(tiles_x, tiles_y, func, grid, block);
        int nbx = (tiles_x + RT_TILE_ORDER_BLOCK - 1) / RT_TILE_ORDER_BLOCK, nby = (tiles_y + RT_TILE_ORDER_BLOCK - 1) / RT_TILE_ORDER_BLOCK;
        int n = tiles_x * tiles_y;
        const unsigned *c_cost = g->o_cost, *c_key = g->o_key, *c_start = g->o_start;
        void *a0[] = {&c_cost, &g->o_key, &nbx, &tiles_x, &tiles_y};
        void *a1[] = {&c_key, &g->o_start, &nbx, &nby, &tiles_x, &tiles_y};
        void *a2[] = {&c_start, &g->o_perm, &n, &tiles_x, &nbx};
        void **args[3] = {a0, a1, a2};
        for (int k = 0; k < 3; ++k) {
            hipKernelNodeParams kp;
            memset(&kp, 0, sizeof kp);
            kp.func = const_cast<void *>(func[k]);
            kp.gridDim = grid[k];
            kp.blockDim = block[k];
            kp.kernelParams = args[k];
            hipGraphNode_t node;
            RT_HIP(hipGraphAddKernelNode(&node, g->graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
            prev = node;
        }
    }
    for (int p = 0; p < g->passes; ++p) {
        const rt_frame_desc fd = pass_desc(g, p);
        RtKernelChoice kc;
        rc = rt_frame_kernel_choice(s, &fd, &kc);
        if (rc != RT_OK) return rc;
        if (g->fc[p].local_rows == 0) continue;
        hipKernelNodeParams &kp = g->pass_params[p];
        memset(&kp, 0, sizeof kp);
        const void *func;
        unsigned lds;
        RT_HIP(rt_dev_trace_config(&g->fc[p], kc.tile, kc.cull, kc.mode, kc.table_lds, kc.feat, &func, &kp.gridDim, &kp.blockDim, &lds));
        kp.func = const_cast<void *>(func);
        kp.sharedMemBytes = lds;
        void *args[] = {&g->fc[p], &g->spheres};
        kp.kernelParams = args;
        RT_HIP(hipGraphAddKernelNode(&g->pass_node[p], g->graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
        kp.kernelParams = nullptr;
        g->pass_live[p] = true;
        prev = g->pass_node[p];
    }
    if (g->host_pixels && g->fd.pixels) {
        int y0 = g->fd.opts.y0, y1 = g->fd.opts.y1;
        if (y0 == 0 && y1 == 0) y1 = g->fd.height;
        hipGraphNode_t copy;
        RT_HIP(hipGraphAddMemcpyNode1D(&copy, g->graph, prev ? &prev : nullptr, prev ? 1 : 0, g->host_pixels, g->fd.pixels,
                                       sizeof(uint32_t) * (size_t)g->fd.width * (size_t)(y1 - y0), hipMemcpyDeviceToHost));
    }
    RT_HIP(hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0));
    g->epoch = rt_scene_epoch(s);
    return RT_OK;
}

extern "C" rt_frame_graph *rt_graph_capture(rt_scene *s, const rt_frame_desc *fd, int passes,
                                            uint32_t *host_pixels, void *stream)
{
    // passes > 0: that many samples per pixel, taken by ONE kernel node (the sample loop runs in the kernel: one pass
    // over the tile setup, no read-modify-write of the float4 frame between samples, no launch gaps -- C4 0.91 ms against
    // 1.00 as four nodes); passes < 0: |passes| progressive one-sample nodes, each adding into opts.rgba (a caller that
    // shows the frame between passes). The frames are the same bits either way.
    const bool progressive = passes < 0;
    if (progressive) passes = -passes;
    if (!s || !fd || passes < 1 || passes > RT_MAX_SPP) {
        rt_set_error("rt_graph_capture: invalid argument");
        return nullptr;
    }
    if (progressive && passes > 1 && !fd->opts.rgba) {
        rt_set_error("rt_graph_capture: progressive passes need opts.rgba (float4 accumulation buffer)");
        return nullptr;
    }
    if (fd->opts.stats) {
        rt_set_error("rt_graph_capture: the instrumented kernels are not recorded into graphs");
        return nullptr;
    }
    if (rt_dev_prepare() != hipSuccess) {
        rt_set_error("rt_graph_capture: kernel image not loadable on this device");
        return nullptr;
    }
    rt_frame_graph *g = new rt_frame_graph();
    g->scene = s;
    g->fd = *fd;
    g->samples = passes;
    g->passes = progressive ? passes : 1;
    g->host_pixels = host_pixels;
    if (build_graph(g, (hipStream_t)stream) != RT_OK) {
        rt_graph_destroy(g);
        return nullptr;
    }
    return g;
}

// Push the argument storage into the instantiated graph.
static int update_nodes(rt_frame_graph *g)
{
    if (g->build_node) {
        void *args[] = {&g->build_tab, &g->build_n, &g->build_org[0], &g->build_org[1], &g->build_org[2], &g->cones};
        g->build_params.kernelParams = args;
        const hipError_t e = hipGraphExecKernelNodeSetParams(g->exec, g->build_node, &g->build_params);
        g->build_params.kernelParams = nullptr;
        RT_HIP(e);
    }
    for (int p = 0; p < g->passes; ++p) {
        if (!g->pass_live[p]) continue;
        void *args[] = {&g->fc[p], &g->spheres};
        g->pass_params[p].kernelParams = args;
        const hipError_t e = hipGraphExecKernelNodeSetParams(g->exec, g->pass_node[p], &g->pass_params[p]);
        g->pass_params[p].kernelParams = nullptr;
        RT_HIP(e);
    }
    return RT_OK;
}

extern "C" int rt_graph_launch(rt_frame_graph *g, void *stream)
{
    if (!g || !g->scene) return RT_ERR_INVALID;
    if (!g->exec || g->epoch != rt_scene_epoch(g->scene)) {
        // the scene's tables were rewritten since the nodes were built: never replay against them
        const int rc = build_graph(g, (hipStream_t)stream);
        if (rc != RT_OK) return rc;
    }
    RT_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return rt_scene_note_launch(g->scene, (hipStream_t)stream, -1);
}

extern "C" int rt_graph_set_camera(rt_frame_graph *g, const rt_camera *cam)
{
    if (!g || !cam || !g->scene) return RT_ERR_INVALID;
    g->fd.cam = *cam;
    float org[3];
    rt_ray_origin(&g->fd, org);
    const bool want_cones = g->fd.opts.cull != 0 && rt_scene_wants_eye_cones(g->scene, org);
    const bool structure_same = g->exec && g->epoch == rt_scene_epoch(g->scene) && want_cones == (g->build_node != nullptr) &&
                                (want_cones ? g->cones_on_device : true);
    if (!structure_same) {   // host-built cones, a camera at a non-finite position, a changed scene: rebuild
        const int rc = rt_scene_quiesce(g->scene);
        if (rc != RT_OK) return rc;
        release_graph(g);
        return RT_OK;        // rebuilt by the next rt_graph_launch, on its stream
    }
    const int rc = fill_arguments(g, want_cones);
    if (rc != RT_OK) return rc;
    return update_nodes(g);
}

extern "C" void rt_graph_destroy(rt_frame_graph *g)
{
    if (!g) return;
    if (g->scene) (void)rt_scene_quiesce(g->scene);
    release_graph(g);
    if (g->cones) (void)hipFree(g->cones);
    if (g->order_buf) (void)hipFree(g->order_buf);
    delete g;
}
