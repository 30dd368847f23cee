// rt_multi.hip -- one frame on several GPUs of a node, from C++ (BASELINE config
// C5): ONE process, one rt_scene per device, the frame's rows dealt to the devices in 16-row
// blocks round-robin (balanced: sky rows are cheap, /root/reference/kernel.cu:1624-1625 uses the
// global y), every device's rows written as 3 bytes per pixel by the frame kernel itself
// (rt_launch_opts.packed24), ONE RCCL gather of those rows over xGMI to device 0, and one small
// kernel there that scatters the rows home and widens them to the 0x00RRGGBB words
// setPixelBuff() consumes. Pixels are independent (one store per thread, kernel.cu:1682/1688):
// nothing else is exchanged.
//
// RCCL is loaded at run time (dlopen) when a multi-device object is created, so the library has
// no link-time dependency on it. A second transport -- the root pulling each peer's rows with
// hipMemcpyPeerAsync, i.e. the SDMA engines over the same xGMI links -- exists for two reasons:
// it needs no collective library, and it accepts the SAME device several times, which is how
// the whole multi-device path (row arithmetic, double buffering, events, scatter kernel) is
// tested on a one-GPU box.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_engine.h"
#include "rt_internal.h"

#define RT_MULTI_BLOCK 16   // rows per block of the round-robin split (a multiple of every tile height)

namespace {

// rows of the frame owned by `rank` (in local-row order) = blocks rank, rank + n, rank + 2n, ...
int rows_of(int height, int rank, int n)
{
    int rows = 0;
    for (int k = rank; k * RT_MULTI_BLOCK < height; k += n)
        rows += (height - k * RT_MULTI_BLOCK < RT_MULTI_BLOCK) ? height - k * RT_MULTI_BLOCK : RT_MULTI_BLOCK;
    return rows;
}

// One thread per four pixels of the assembled frame: three dwords in (12 bytes = B,G,R of four
// pixels), one uint4 out. recv holds `n` slots of `slot_rows` rows of `width * 3` bytes each;
// frame row y lives in slot (y / 16) % n at local row (y / 16 / n) * 16 + y % 16.
// `height` rows are assembled (a whole frame, or one row band of it whose first row `frame` points at).
__global__ void rt_scatter_rows24(const unsigned *__restrict__ recv, uint4 *__restrict__ frame, int width, int height, int n,
                                  int slot_rows)
{
    const int quads = width >> 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)quads * height) return;
    const int y = (int)(t / quads), q = (int)(t % quads);
    const int blk = y / RT_MULTI_BLOCK, rank = blk % n, lrow = (blk / n) * RT_MULTI_BLOCK + y % RT_MULTI_BLOCK;
    const unsigned *src = recv + ((size_t)rank * slot_rows + lrow) * (size_t)(quads * 3) + (size_t)q * 3;
    const unsigned a = src[0], b = src[1], c = src[2];
    uint4 o;
    o.x = a & 0x00ffffffu;
    o.y = (a >> 24) | ((b & 0x0000ffffu) << 8);
    o.z = (b >> 16) | ((c & 0x000000ffu) << 16);
    o.w = c >> 8;
    frame[(size_t)y * quads + q] = o;
}

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load()
    {
        if (handle) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
        Gather = (decltype(Gather))dlsym(handle, "ncclGather");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Gather && GetErrorString;
    }
};
Rccl g_rccl;

struct Dev {
    int device = 0;
    rt_scene *scene = nullptr;
    hipStream_t stream = nullptr;          // this device's frames: kernel, then its part of the gather
    unsigned *send[2] = {nullptr, nullptr}; // peers: the rows rendered, 3 bytes per pixel, two frames deep
    ncclComm_t comm = nullptr;
    hipEvent_t rendered[2] = {nullptr, nullptr};
};

}  // namespace

struct rt_multi {
    int n = 0;
    int transport = RT_MULTI_RCCL;
    std::vector<Dev> dev;
    // root (dev[0]) side
    hipStream_t copy_stream[2] = {nullptr, nullptr};   // peer-copy transport: pulls run beside the root's kernel
    unsigned *recv[2] = {nullptr, nullptr};            // n slots of slot_rows rows, two frames deep
    uint32_t *frame[2] = {nullptr, nullptr};           // assembled frames when the caller passes no buffer
    hipEvent_t assembled[2] = {nullptr, nullptr};      // scatter of buffer set b done: the set may be re-used
    hipEvent_t pulled[2] = {nullptr, nullptr};
    bool set_used[2] = {false, false};
    int width = 0, height = 0, slot_rows = 0;
    unsigned long long frames = 0;
    uint32_t *last = nullptr;
    int last_set = -1;                                 // buffer set of the last enqueued frame / band
    unsigned long long gathers = 0;                    // ncclGather groups issued (rt_multi_gathers)
    std::string note;                                  // what create() decided and why (rt_multi_note)
};

#define RT_NCCL(expr)                                                                                   \
    do {                                                                                                \
        const ncclResult_t r_ = (expr);                                                                 \
        if (r_ != ncclSuccess) {                                                                        \
            rt_set_error("RCCL error %d (%s) at %s:%d '%s'", (int)r_, g_rccl.GetErrorString(r_), __FILE__, __LINE__, #expr); \
            return RT_ERR_HIP;                                                                          \
        }                                                                                               \
    } while (0)

static int release_buffers(rt_multi *m)
{
    for (int d = 0; d < m->n; ++d) {
        RT_HIP(hipSetDevice(m->dev[d].device));
        RT_HIP(hipStreamSynchronize(m->dev[d].stream));
        for (int b = 0; b < 2; ++b) {
            if (m->dev[d].send[b]) RT_HIP(hipFree(m->dev[d].send[b]));
            m->dev[d].send[b] = nullptr;
        }
    }
    RT_HIP(hipSetDevice(m->dev[0].device));
    for (int b = 0; b < 2; ++b) {
        if (m->copy_stream[b]) RT_HIP(hipStreamSynchronize(m->copy_stream[b]));
        if (m->recv[b]) RT_HIP(hipFree(m->recv[b]));
        if (m->frame[b]) RT_HIP(hipFree(m->frame[b]));
        m->recv[b] = nullptr;
        m->frame[b] = nullptr;
        m->set_used[b] = false;
    }
    m->width = m->height = m->slot_rows = 0;
    return RT_OK;
}

extern "C" void rt_multi_destroy(rt_multi *m)
{
    if (!m) return;
    if (m->n > 0) (void)release_buffers(m);
    for (Dev &d : m->dev) {
        (void)hipSetDevice(d.device);
        if (d.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d.comm);
        if (d.scene) rt_scene_destroy(d.scene);
        if (d.stream) (void)hipStreamDestroy(d.stream);
        for (hipEvent_t e : d.rendered)
            if (e) (void)hipEventDestroy(e);
    }
    if (!m->dev.empty()) (void)hipSetDevice(m->dev[0].device);
    for (int b = 0; b < 2; ++b) {
        if (m->copy_stream[b]) (void)hipStreamDestroy(m->copy_stream[b]);
        if (m->assembled[b]) (void)hipEventDestroy(m->assembled[b]);
        if (m->pulled[b]) (void)hipEventDestroy(m->pulled[b]);
    }
    delete m;
}

static int enable_peer(int root, int peer)
{
    if (peer == root) return RT_OK;
    int can = 0;
    RT_HIP(hipDeviceCanAccessPeer(&can, root, peer));
    if (can) {
        RT_HIP(hipSetDevice(root));
        const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) RT_HIP(e);
        (void)hipGetLastError();
    }
    return RT_OK;
}

// Load RCCL, create the communicators (one process, one per device) and run the frame's exchange once on a
// known pattern: every device sends 256 bytes of (0x40 + rank), in place on the root exactly as
// rt_multi_render does it, and the root's receive buffer is checked on the host. The transport is only
// taken when that gather delivered every rank's bytes to its slot.
static int rccl_bring_up(rt_multi *m, const int *devices)
{
    const int n = m->n;
    if (!g_rccl.load()) {
        rt_set_error("rt_multi_create: librccl.so is not loadable (%s)", dlerror());
        return RT_ERR_UNSUPPORTED;
    }
    std::vector<ncclComm_t> comms((size_t)n);
    RT_NCCL(g_rccl.CommInitAll(comms.data(), n, devices));
    for (int i = 0; i < n; ++i) m->dev[i].comm = comms[i];
    const size_t slot = 256;
    std::vector<unsigned char *> bufs((size_t)n, nullptr);
    int rc = RT_OK;
    auto body = [&]() -> int {
        for (int i = 0; i < n; ++i) {
            RT_HIP(hipSetDevice(devices[i]));
            RT_HIP(hipMalloc((void **)&bufs[(size_t)i], i == 0 ? slot * (size_t)n : slot));
            if (i == 0) RT_HIP(hipMemsetAsync(bufs[0], 0, slot * (size_t)n, m->dev[0].stream));
            RT_HIP(hipMemsetAsync(bufs[(size_t)i], 0x40 + i, slot, m->dev[i].stream));
        }
        RT_NCCL(g_rccl.GroupStart());
        for (int i = 0; i < n; ++i) {
            RT_HIP(hipSetDevice(devices[i]));
            RT_NCCL(g_rccl.Gather(bufs[(size_t)i], i == 0 ? (void *)bufs[0] : nullptr, slot, ncclUint8, 0, m->dev[i].comm, m->dev[i].stream));
        }
        RT_NCCL(g_rccl.GroupEnd());
        for (int i = 0; i < n; ++i) {
            RT_HIP(hipSetDevice(devices[i]));
            RT_HIP(hipStreamSynchronize(m->dev[i].stream));
        }
        std::vector<unsigned char> host(slot * (size_t)n);
        RT_HIP(hipSetDevice(devices[0]));
        RT_HIP(hipMemcpy(host.data(), bufs[0], host.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i)
            for (size_t k = 0; k < slot; ++k)
                if (host[slot * (size_t)i + k] != (unsigned char)(0x40 + i)) {
                    rt_set_error("rt_multi_create: RCCL gather self-test: byte %zu of rank %d's slot is 0x%02x, expected 0x%02x",
                                 k, i, host[slot * (size_t)i + k], 0x40 + i);
                    return RT_ERR_HIP;
                }
        return RT_OK;
    };
    rc = body();
    for (int i = 0; i < n; ++i)
        if (bufs[(size_t)i]) {
            (void)hipSetDevice(devices[i]);
            (void)hipFree(bufs[(size_t)i]);
        }
    (void)hipSetDevice(devices[0]);
    return rc;
}

static int create_impl(rt_multi *m, const int *devices, int n, int transport)
{
    int have = 0;
    RT_HIP(hipGetDeviceCount(&have));
    bool distinct = true;
    for (int i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= have) {
            rt_set_error("rt_multi_create: device %d of %d", devices[i], have);
            return RT_ERR_NO_DEVICE;
        }
        for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
    }
    const bool automatic = transport == RT_MULTI_AUTO;
    if (automatic) transport = distinct ? RT_MULTI_RCCL : RT_MULTI_PEER_COPY;
    if (transport == RT_MULTI_RCCL && !distinct) {
        rt_set_error("rt_multi_create: the RCCL transport needs distinct devices (use RT_MULTI_PEER_COPY)");
        return RT_ERR_INVALID;
    }
    m->n = n;
    m->transport = transport;
    m->dev.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        Dev &d = m->dev[i];
        d.device = devices[i];
        RT_HIP(hipSetDevice(d.device));
        d.scene = rt_scene_create();
        RT_HIP(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
        for (hipEvent_t &e : d.rendered) RT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (transport == RT_MULTI_PEER_COPY) {
            const int prc = enable_peer(devices[0], d.device);
            if (prc != RT_OK) return prc;
        }
    }
    RT_HIP(hipSetDevice(devices[0]));
    for (int b = 0; b < 2; ++b) {
        RT_HIP(hipStreamCreateWithFlags(&m->copy_stream[b], hipStreamNonBlocking));
        RT_HIP(hipEventCreateWithFlags(&m->assembled[b], hipEventDisableTiming));
        RT_HIP(hipEventCreateWithFlags(&m->pulled[b], hipEventDisableTiming));
    }
    if (transport == RT_MULTI_RCCL) {
        const int rc = rccl_bring_up(m, devices);
        if (rc != RT_OK) {
            if (!automatic) return rc;
            // RT_MULTI_AUTO: the collective library is missing or its gather did not deliver -- the peer-copy
            // transport moves the same rows over the same links
            m->note = std::string("RCCL transport unusable (") + rt_last_error() + "); using peer copies";
            for (Dev &d : m->dev) {
                if (d.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d.comm);
                d.comm = nullptr;
            }
            m->transport = RT_MULTI_PEER_COPY;
            for (int i = 1; i < n; ++i) {
                const int prc = enable_peer(devices[0], devices[i]);
                if (prc != RT_OK) return prc;
            }
            RT_HIP(hipSetDevice(devices[0]));
        } else {
            m->note = "RCCL transport: gather self-test passed";
        }
    } else {
        m->note = "peer-copy transport";
    }
    return RT_OK;
}

extern "C" int rt_multi_create_ex(const int *devices, int n, int transport, rt_multi **out)
{
    if (!devices || n < 1 || n > 64 || !out || transport < RT_MULTI_AUTO || transport > RT_MULTI_PEER_COPY) {
        rt_set_error("rt_multi_create: invalid argument");
        return RT_ERR_INVALID;
    }
    rt_multi *m = new rt_multi();
    const int rc = create_impl(m, devices, n, transport);
    if (rc != RT_OK) {
        rt_multi_destroy(m);
        return rc;
    }
    *out = m;
    return RT_OK;
}

extern "C" rt_multi *rt_multi_create(int n_gpus)
{
    if (n_gpus < 1 || n_gpus > 64) {
        rt_set_error("rt_multi_create: n_gpus %d", n_gpus);
        return nullptr;
    }
    std::vector<int> devs((size_t)n_gpus);
    for (int i = 0; i < n_gpus; ++i) devs[i] = i;
    rt_multi *m = nullptr;
    return rt_multi_create_ex(devs.data(), n_gpus, RT_MULTI_AUTO, &m) == RT_OK ? m : nullptr;
}

extern "C" int rt_multi_device_count(const rt_multi *m) { return m ? m->n : 0; }
extern "C" int rt_multi_transport(const rt_multi *m) { return m ? m->transport : -1; }
extern "C" rt_scene *rt_multi_scene(rt_multi *m, int i, int *device)
{
    if (!m || i < 0 || i >= m->n) return nullptr;
    if (device) *device = m->dev[i].device;
    return m->dev[i].scene;
}

// the same scene on every device
#define RT_EACH_DEVICE(call)                                          \
    do {                                                              \
        if (!m) return RT_ERR_INVALID;                                \
        for (Dev & d : m->dev) {                                      \
            RT_HIP(hipSetDevice(d.device));                           \
            const int rc_ = (call);                                   \
            if (rc_ != RT_OK) return rc_;                             \
        }                                                             \
        RT_HIP(hipSetDevice(m->dev[0].device));                       \
        return RT_OK;                                                 \
    } while (0)

extern "C" int rt_multi_set_spheres(rt_multi *m, const rt_sphere *s, int n) { RT_EACH_DEVICE(rt_scene_set_spheres(d.scene, s, n)); }
extern "C" int rt_multi_set_planes(rt_multi *m, const rt_plane *p, int n) { RT_EACH_DEVICE(rt_scene_set_planes(d.scene, p, n)); }
extern "C" int rt_multi_set_cubes(rt_multi *m, const rt_cube *c, int n) { RT_EACH_DEVICE(rt_scene_set_cubes(d.scene, c, n)); }
extern "C" int rt_multi_set_mesh(rt_multi *m, const rt_mesh *mesh) { RT_EACH_DEVICE(rt_scene_set_mesh(d.scene, mesh)); }
extern "C" int rt_multi_set_texture(rt_multi *m, const float *r, const float *g, const float *b, int w, int h)
{
    RT_EACH_DEVICE(rt_scene_set_texture(d.scene, r, g, b, w, h));
}
extern "C" int rt_multi_set_sky(rt_multi *m, const rt_sphere *box, const float *r, const float *g, const float *b, int w, int h)
{
    RT_EACH_DEVICE(rt_scene_set_sky(d.scene, box, r, g, b, w, h));
}
extern "C" int rt_multi_set_lights(rt_multi *m, const rt_light *l, int n) { RT_EACH_DEVICE(rt_scene_set_lights(d.scene, l, n)); }

static int ensure_buffers(rt_multi *m, int width, int height)
{
    if (m->width == width && m->height == height) return RT_OK;
    int rc = release_buffers(m);
    if (rc != RT_OK) return rc;
    const int blocks = (height + RT_MULTI_BLOCK - 1) / RT_MULTI_BLOCK;
    const int slot_rows = ((blocks + m->n - 1) / m->n) * RT_MULTI_BLOCK;   // the largest share, whole blocks
    const size_t slot_bytes = (size_t)slot_rows * (size_t)width * 3;
    RT_HIP(hipSetDevice(m->dev[0].device));
    for (int b = 0; b < 2; ++b) {
        RT_HIP(hipMalloc((void **)&m->recv[b], slot_bytes * (size_t)m->n));
        RT_HIP(hipMalloc((void **)&m->frame[b], sizeof(uint32_t) * (size_t)width * (size_t)height));
    }
    for (int d = 1; d < m->n; ++d) {
        RT_HIP(hipSetDevice(m->dev[d].device));
        for (int b = 0; b < 2; ++b) RT_HIP(hipMalloc((void **)&m->dev[d].send[b], slot_bytes));
    }
    RT_HIP(hipSetDevice(m->dev[0].device));
    m->width = width;
    m->height = height;
    m->slot_rows = slot_rows;
    return RT_OK;
}

// One frame, or one row band of it. fd: the whole frame's width, height, aspect, cam; opts.spp / cull / tile
// honoured; opts.y0 / y1 (both multiples of 16 rows, or 0 / 0 for the whole frame) select a row band, whose
// 16-row blocks are dealt to the devices from the band's first row; interleave and output pointers of fd are
// ignored. The assembled 0x00RRGGBB rows land at their place in `pixels_dev0` (device memory of the first
// device, the WHOLE frame's buffer) or, when that is null, in an internal frame buffer (rt_multi_frame).
// Asynchronous: returns when the work is enqueued; two frames (or bands) may be in flight.
extern "C" int rt_multi_render(rt_multi *m, const rt_frame_desc *fd, uint32_t *pixels_dev0)
{
    if (!m || !fd || fd->width <= 0 || fd->height <= 0) {
        rt_set_error("rt_multi_render: invalid argument");
        return RT_ERR_INVALID;
    }
    const int w = fd->width, h = fd->height, n = m->n;
    int y0 = fd->opts.y0, y1 = fd->opts.y1;
    if (y0 == 0 && y1 == 0) y1 = h;
    if (y0 < 0 || y1 > h || y0 >= y1 || y0 % RT_MULTI_BLOCK != 0 || (y1 % RT_MULTI_BLOCK != 0 && y1 != h)) {
        rt_set_error("rt_multi_render: bad row band [%d,%d) of %d rows (whole %d-row blocks)", y0, y1, h, RT_MULTI_BLOCK);
        return RT_ERR_INVALID;
    }
    const int hb = y1 - y0;   // rows of this band
    int rc;
    // one device and no collective asked for: it renders the rows where they are wanted. (With the RCCL transport
    // named explicitly a single device still runs the whole exchange -- 24-bit rows, the in-place one-rank
    // gather, the scatter kernel -- so that every call of the multi-device path executes on a one-GPU box.)
    if (n == 1 && (m->transport != RT_MULTI_RCCL || w % 4 != 0)) {
        rc = ensure_buffers(m, w, h);
        if (rc != RT_OK) return rc;
        const int b = (int)(m->frames & 1);
        RT_HIP(hipSetDevice(m->dev[0].device));
        if (m->set_used[b]) RT_HIP(hipStreamWaitEvent(m->dev[0].stream, m->assembled[b], 0));
        rt_frame_desc f = *fd;
        f.opts.y0 = y0;
        f.opts.y1 = y1;
        f.opts.interleave_count = f.opts.interleave_index = f.opts.interleave_rows = 0;
        f.opts.rgba = nullptr;
        f.opts.packed24 = nullptr;
        f.opts.stats = nullptr;
        uint32_t *out = pixels_dev0 ? pixels_dev0 : m->frame[b];
        f.pixels = out + (size_t)y0 * (size_t)w;
        rc = rt_scene_render(m->dev[0].scene, &f, m->dev[0].stream);
        if (rc != RT_OK) return rc;
        RT_HIP(hipEventRecord(m->assembled[b], m->dev[0].stream));
        m->set_used[b] = true;
        m->last = out;
        m->last_set = b;
        m->frames++;
        return RT_OK;
    }
    if (w % 4 != 0) {
        rt_set_error("rt_multi_render: the 24-bit rows need a frame width that is a multiple of 4 (got %d)", w);
        return RT_ERR_INVALID;
    }
    rc = ensure_buffers(m, w, h);
    if (rc != RT_OK) return rc;
    const int b = (int)(m->frames & 1);   // buffer set of this frame
    // slots of the receive buffer: the largest share of THIS band, whole blocks (a whole frame: m->slot_rows)
    const int blocks = (hb + RT_MULTI_BLOCK - 1) / RT_MULTI_BLOCK;
    const int slot_rows = ((blocks + n - 1) / n) * RT_MULTI_BLOCK;
    const size_t slot_bytes = (size_t)slot_rows * (size_t)w * 3;
    Dev &root = m->dev[0];
    // every device renders its rows (3 bytes per pixel) -- the root straight into its slot of the receive buffer
    for (int d = 0; d < n; ++d) {
        Dev &dv = m->dev[d];
        RT_HIP(hipSetDevice(dv.device));
        // buffer set b was last read by the scatter of frame (frames - 2): wait for it on the device
        if (m->set_used[b]) RT_HIP(hipStreamWaitEvent(dv.stream, m->assembled[b], 0));
        rt_frame_desc f = *fd;
        f.pixels = nullptr;
        f.opts.rgba = nullptr;
        f.opts.stats = nullptr;
        f.opts.y0 = y0;
        f.opts.y1 = y1;
        f.opts.interleave_count = n;
        f.opts.interleave_index = d;
        f.opts.interleave_rows = RT_MULTI_BLOCK;
        f.opts.packed24 = d == 0 ? (void *)m->recv[b] : (void *)dv.send[b];
        if (n == 1) f.opts.interleave_count = f.opts.interleave_index = f.opts.interleave_rows = 0;   // (a band as it is)
        if (rows_of(hb, d, n) > 0) {
            rc = rt_scene_render(dv.scene, &f, dv.stream);
            if (rc != RT_OK) return rc;
        }
        RT_HIP(hipEventRecord(dv.rendered[b], dv.stream));
    }
    // the frame's single exchange
    hipStream_t assemble_on = root.stream;
    if (m->transport == RT_MULTI_RCCL) {
        RT_NCCL(g_rccl.GroupStart());
        for (int d = 0; d < n; ++d) {
            Dev &dv = m->dev[d];
            RT_HIP(hipSetDevice(dv.device));
            // in place on the root: its send buffer is its own slot of the receive buffer
            const void *src = d == 0 ? (const void *)m->recv[b] : (const void *)dv.send[b];
            RT_NCCL(g_rccl.Gather(src, d == 0 ? (void *)m->recv[b] : nullptr, slot_bytes, ncclUint8, 0, dv.comm, dv.stream));
        }
        RT_NCCL(g_rccl.GroupEnd());
        m->gathers++;
    } else {
        // the root pulls: one asynchronous peer copy per peer, all on the set's copy stream (SDMA engines)
        RT_HIP(hipSetDevice(root.device));
        assemble_on = m->copy_stream[b];
        for (int d = 0; d < n; ++d) RT_HIP(hipStreamWaitEvent(assemble_on, m->dev[d].rendered[b], 0));
        for (int d = 1; d < n; ++d) {
            const size_t bytes = (size_t)rows_of(hb, d, n) * (size_t)w * 3;
            if (!bytes) continue;
            char *dst = (char *)m->recv[b] + slot_bytes * (size_t)d;
            if (m->dev[d].device == root.device)   // the same device twice (one-GPU rehearsal of the path)
                RT_HIP(hipMemcpyAsync(dst, m->dev[d].send[b], bytes, hipMemcpyDeviceToDevice, assemble_on));
            else
                RT_HIP(hipMemcpyPeerAsync(dst, root.device, m->dev[d].send[b], m->dev[d].device, bytes, assemble_on));
        }
    }
    // rows home, 24 -> 32 bits
    RT_HIP(hipSetDevice(root.device));
    uint32_t *out = pixels_dev0 ? pixels_dev0 : m->frame[b];
    const long long threads = (long long)(w / 4) * hb;
    hipLaunchKernelGGL(rt_scatter_rows24, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, assemble_on, m->recv[b],
                       reinterpret_cast<uint4 *>(out + (size_t)y0 * (size_t)w), w, hb, n, slot_rows);
    RT_HIP(hipGetLastError());
    RT_HIP(hipEventRecord(m->assembled[b], assemble_on));
    m->set_used[b] = true;
    m->last = out;
    m->last_set = b;
    m->frames++;
    return RT_OK;
}

// Make `stream` (a stream of the first device) wait, on the device, for the frame / band enqueued last:
// what follows on it -- a copy of the assembled rows to the host, a present -- then needs no host wait.
extern "C" int rt_multi_stream_wait(rt_multi *m, void *stream)
{
    if (!m || m->last_set < 0) {
        rt_set_error("rt_multi_stream_wait: nothing has been rendered");
        return RT_ERR_INVALID;
    }
    RT_HIP(hipSetDevice(m->dev[0].device));
    RT_HIP(hipStreamWaitEvent((hipStream_t)stream, m->assembled[m->last_set], 0));
    return RT_OK;
}

extern "C" const char *rt_multi_note(const rt_multi *m) { return m ? m->note.c_str() : ""; }
extern "C" unsigned long long rt_multi_gathers(const rt_multi *m) { return m ? m->gathers : 0ull; }

// The root side of the gather as one call, for hosts that run the exchange themselves (one process
// per GPU under torch.distributed / MPI: bench.py): `recv` holds n slots of slot_rows rows of
// width*3 bytes (slot r = what rank r rendered with interleave (n, r, 16) and opts.packed24);
// `frame` receives the width*height words. On `stream`, current device.
extern "C" int rt_assemble_rows24(const void *recv, uint32_t *frame, int width, int height, int n, int slot_rows, void *stream)
{
    if (!recv || !frame || width <= 0 || height <= 0 || width % 4 != 0 || n < 1 || slot_rows < 1) {
        rt_set_error("rt_assemble_rows24: invalid argument");
        return RT_ERR_INVALID;
    }
    const long long threads = (long long)(width / 4) * height;
    hipLaunchKernelGGL(rt_scatter_rows24, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned *>(recv), reinterpret_cast<uint4 *>(frame), width, height, n, slot_rows);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

// Wait for every frame enqueued so far (all devices).
extern "C" int rt_multi_sync(rt_multi *m)
{
    if (!m) return RT_ERR_INVALID;
    for (Dev &d : m->dev) {
        RT_HIP(hipSetDevice(d.device));
        RT_HIP(hipStreamSynchronize(d.stream));
    }
    RT_HIP(hipSetDevice(m->dev[0].device));
    for (int b = 0; b < 2; ++b) RT_HIP(hipStreamSynchronize(m->copy_stream[b]));
    return RT_OK;
}

extern "C" const uint32_t *rt_multi_frame(const rt_multi *m) { return m ? m->last : nullptr; }

// Synchronise and copy the last assembled frame to host memory (width * height words).
extern "C" int rt_multi_download(rt_multi *m, uint32_t *host)
{
    if (!m || !host || !m->last) return RT_ERR_INVALID;
    const int rc = rt_multi_sync(m);
    if (rc != RT_OK) return rc;
    RT_HIP(hipMemcpy(host, m->last, sizeof(uint32_t) * (size_t)m->width * (size_t)m->height, hipMemcpyDeviceToHost));
    return RT_OK;
}
