// rt_kernels.hip -- the frame kernel's default instantiations (8x8 tile, culling, tables in global memory), the
// diagnostic kernels, and the host-side launchers. The kernel itself is rt_trace.inc.
#include "rt_trace.inc"

namespace {
// ---------------------------------------------------------------------------
// diagnostics: scalar building blocks evaluated on the device (tests only)
// ---------------------------------------------------------------------------
__global__ void rt_dbg_math(int op, const float *a, const float *b, float *out, int n)
{
    __shared__ double atab[16];   // as the frame kernel: atan(k/8) read from LDS
    if (threadIdx.x < 16) atab[threadIdx.x] = kAtanEighth[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (op) {
    case 0: r = rtm::cosf_rt(a[i]); break;
    case 1: r = rtm::sinf_rt(a[i]); break;
    case 2: r = rtm::acosf_rt(a[i], atab); break;
    case 4: r = (float)((1.0 + rtm::div_by_3p1415((double)a[i])) * 0.5); break;   // kernel.cu:1402
    case 5: r = (float)rtm::div_by_3p1415((double)a[i]); break;                  // kernel.cu:1403
    default: r = rtm::atan2f_rt(a[i], b[i], atab); break;
    }
    out[i] = r;
}

__global__ void rt_dbg_intersect(const float4 *tab, const float *rays, int n, int *hit, float *t)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 o{rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]};
    const V3 d{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]};
    const RayK r = make_ray(o, d);
    const Quad q = quadratic(r, tab[i]);
    float tt;
    hit[i] = intersect_tail(r, q, tt) ? 1 : 0;
    t[i] = tt;
}


// castLightRay for n independent (start, normal) pairs against the whole table
// (brute force, no culling): the 10 sample directions and the returned brightness.
__global__ void rt_dbg_light(const RtFrameConsts fc, const float4 *tab, const float *starts,
                             const float *normals, int light_index, int n, float *dirs, float *bright,
                             float *adirs, int *aok)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const int k = live ? i : 0;
    const V3 start{starts[3 * k + 0], starts[3 * k + 1], starts[3 * k + 2]};
    const V3 normal{normals[3 * k + 0], normals[3 * k + 1], normals[3 * k + 2]};
    const AuxPtr ax = (AuxPtr)(uintptr_t)fc.aux;
    const bool force_slow = (fc.flags & RT_FLAG_FORCE_SLOW) != 0;
    const RtLightDev L = ax->lights[light_index];
    ShadowChain<false> chain;
    if (adirs) {   // the pre-pass's approximate directions, formed exactly as the frame kernel forms them
        V3 t{L.px - start.x, L.py - start.y, L.pz - start.z};
        const float inv = __builtin_amdgcn_rsqf(__builtin_fmaf(t.x, t.x, __builtin_fmaf(t.y, t.y, t.z * t.z)));
        t.x *= inv; t.y *= inv; t.z *= inv;
        const bool lane_ok = chain.setup_approx(L, start, t);
        for (int j = 0; j < RT_SHADOW_SAMPLES; ++j) {
            bool ok;
            const V3 d = chain.direction_approx(ax, L, j, ok);
            if (live) {
                adirs[30 * i + 3 * j + 0] = d.x;
                adirs[30 * i + 3 * j + 1] = d.y;
                adirs[30 * i + 3 * j + 2] = d.z;
                aok[10 * i + j] = (lane_ok && ok) ? 1 : 0;
            }
        }
    }
    chain.begin(V3{L.px, L.py, L.pz}, start);
    int unshadowed = 0;
    for (int j = 0; j < RT_SHADOW_SAMPLES; ++j) {
        const V3 d = chain.direction(ax, force_slow, L, start, j);
        if (live) {
            dirs[30 * i + 3 * j + 0] = d.x;
            dirs[30 * i + 3 * j + 1] = d.y;
            dirs[30 * i + 3 * j + 2] = d.z;
        }
        const RayK sr = make_ray(start, d);
        bool shadowed = !live;
        for (int e = 0; e < fc.n_spheres; ++e) {
            shadow_test(sr, tab[e], shadowed, force_slow);
            if (__all(shadowed)) break;
        }
        if (!shadowed) unshadowed += 1;
    }
    if (live) {
        float b = brightness_steps(unshadowed);
        const float a = dot3(normal, chain.toL);
        bright[i] = b * (a > 0.f ? a : 0.f);
    }
}

// The shortcuts of the culling kernels against the long forms (rt_debug_shortcuts, tests only).
__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float unit_float(unsigned h) { return (float)(h >> 8) * 0x1.0p-24f; }   // [0, 1)

__global__ void rt_dbg_shortcuts(int what, unsigned seed, long long n, unsigned long long *out)
{
    __shared__ double atab[16];
    if (threadIdx.x < 16) atab[threadIdx.x] = kAtanEighth[threadIdx.x];
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0, accepted = 0, wrong = 0;
    float emax_x = 0.f, emax_y = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned h0 = hash32((unsigned)i * 3u + seed), h1 = hash32((unsigned)i * 3u + 1u + seed * 7919u),
                       h2 = hash32((unsigned)i * 3u + 2u + seed * 104729u), h3 = hash32(h0 ^ (unsigned)(i >> 32) ^ 0x9e3779b9u);
        if (what == 0) {
            // vectors of every scale: unit-ish, tiny, huge, with zero / denormal components now and then
            const int e = (int)(h3 % 61u) - 30 + ((h3 >> 8) % 7u == 0 ? ((h3 >> 12) % 2u ? 60 : -60) : 0);
            const float sc = __builtin_ldexpf(1.f, e);
            V3 a{(unit_float(h0) * 2.f - 1.f) * sc, (unit_float(h1) * 2.f - 1.f) * sc, (unit_float(h2) * 2.f - 1.f) * sc};
            if ((h3 >> 20) % 13u == 0) a.x = 0.f;
            if ((h3 >> 24) % 17u == 0) a.y = -0.f;
            if ((h3 >> 28) % 5u == 0) a.z = a.z * 0x1.0p-100f;
            V3 b = a;
            const V3 ra = normalise_inplace(a);
            const V3 rb = normalise_t<true>(b);
            const bool same = __builtin_bit_cast(unsigned, a.x) == __builtin_bit_cast(unsigned, b.x) &&
                              __builtin_bit_cast(unsigned, a.y) == __builtin_bit_cast(unsigned, b.y) &&
                              __builtin_bit_cast(unsigned, a.z) == __builtin_bit_cast(unsigned, b.z) &&
                              __builtin_bit_cast(unsigned, ra.x) == __builtin_bit_cast(unsigned, rb.x) &&
                              __builtin_bit_cast(unsigned, ra.y) == __builtin_bit_cast(unsigned, rb.y) &&
                              __builtin_bit_cast(unsigned, ra.z) == __builtin_bit_cast(unsigned, rb.z);
            if (!same) bad += 1;
            // the same vector with a +-0 z component (castLightRay's rotation axis) through normalise_z0_t
            V3 c{a.x * sc, b.y * sc, (h3 & 1u) ? 0.f : -0.f}, d = c;   // (a, b are normalised by now: re-scaled)
            if ((h3 >> 3) % 11u == 0) c.x = d.x = 0.f;
            const V3 rc = normalise_inplace(c);
            const V3 rd = normalise_z0_t<1>(d);
            const bool same0 = __builtin_bit_cast(unsigned, c.x) == __builtin_bit_cast(unsigned, d.x) &&
                               __builtin_bit_cast(unsigned, c.y) == __builtin_bit_cast(unsigned, d.y) &&
                               __builtin_bit_cast(unsigned, c.z) == __builtin_bit_cast(unsigned, d.z) &&
                               __builtin_bit_cast(unsigned, rc.x) == __builtin_bit_cast(unsigned, rd.x) &&
                               __builtin_bit_cast(unsigned, rc.y) == __builtin_bit_cast(unsigned, rd.y) &&
                               __builtin_bit_cast(unsigned, rc.z) == __builtin_bit_cast(unsigned, rd.z);
            if (!same0) bad += 1;
        } else if (what == 2) {
            // unit normals: uniform on the sphere, plus clusters at the poles and the seams
            float z = unit_float(h0) * 2.f - 1.f, phi = unit_float(h1) * 6.2831853f;
            if ((h3 & 15u) == 0) z = 1.f - unit_float(h2) * 1.0e-6f;
            if ((h3 & 15u) == 1) z = -1.f + unit_float(h2) * 1.0e-6f;
            if ((h3 & 15u) == 2) phi = (float)((h2 >> 4) & 7u) * 0.78539816f + (unit_float(h2) - 0.5f) * 1.0e-6f;
            const float r = __builtin_sqrtf(__builtin_fmaxf(1.f - z * z, 0.f));
            V3 nrm{r * __builtin_cosf(phi), z, r * __builtin_sinf(phi)};
            normalise_inplace(nrm);
            const float tx = (float)((1.0 + rtm::div_by_3p1415((double)rtm::atan2f_rt(nrm.z, nrm.x, atab))) * 0.5);
            const float ty = (float)rtm::div_by_3p1415((double)rtm::acosf_rt(nrm.y, atab));
            float ux, uy;
            approx_sphere_uv(nrm, ux, uy);
            const float ex = __builtin_fabsf(ux - tx), ey = __builtin_fabsf(uy - ty);
            // a NaN approximation (0/0 at the poles, where the exact atan2 is defined) is simply never "sure"
            if (ex == ex) emax_x = __builtin_fmaxf(emax_x, ex);
            if (ey == ey) emax_y = __builtin_fmaxf(emax_y, ey);
            const int w = 512, hh = 512;
            const float mu = (float)w * (RT_UV_DELTA + 0x1.0p-22f) * 1.01f;
            const int cf = sure_texel(ux, uy, w, hh, mu, mu);
            if (cf >= 0) {
                accepted += 1;
                if (cf != f2i(ty * (float)hh) * w + f2i(tx * (float)w)) wrong += 1;
            }
            // the sky's own (binary32) expressions, kernel.cu:1157-1158, at 2048 x 1024 with its wider margin
            const int sw = 2048, sh = 1024;
            const int cs = sure_texel(ux, uy, sw, sh, (float)sw * (1.0e-6f + 0x1.0p-22f) * 1.01f, (float)sh * (1.0e-6f + 0x1.0p-22f) * 1.01f);
            if (cs >= 0) {
                const int ix = f2i((1.f + rtm::atan2f_rt(nrm.z, nrm.x, atab) / 3.1415f) * 0.5f * (float)sw);
                const int iy = f2i(rtm::acosf_rt(nrm.y, atab) / 3.1415f * (float)sh);
                if (cs != iy * sw + ix) wrong += 1;
            }
        }
    }
    if (what == 1) {   // every float of [2^-96, 2^40]: bit patterns 0x0F800000 .. 0x53800000
        const unsigned lo = 0x0F800000u, hi = 0x53800000u;
        for (unsigned long long b = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += (unsigned long long)stride) {
            const float x = __builtin_bit_cast(float, (unsigned)b);
            if (__builtin_bit_cast(unsigned, lean_sqrt(x)) != __builtin_bit_cast(unsigned, __builtin_sqrtf(x))) bad += 1;
        }
    }
    if (bad) atomicAdd(&out[0], bad);
    if (what == 2) {
        atomicMax(&out[0], (unsigned long long)__builtin_bit_cast(unsigned, emax_x));
        atomicMax(&out[1], (unsigned long long)__builtin_bit_cast(unsigned, emax_y));
        if (accepted) atomicAdd(&out[2], accepted);
        if (wrong) atomicAdd(&out[3], wrong);
    }
}

}  // namespace

RtTraceFn rt_trace_fn_cull8(int mode, int feat, int multi)
{
    return multi ? trace_fn_mode_feat<8, true, false, true>(mode, feat) : trace_fn_mode_feat<8, true, false, false>(mode, feat);
}


// ---------------------------------------------------------------------------
// host-side launchers (called from rt_engine.cpp / rt_graph.cpp)
// ---------------------------------------------------------------------------
// The instantiations that exist: trace_exists() in rt_trace.inc. Tile widths other than 8 and whole-table LDS
// staging are tuning / test dimensions: TABLDS exists for the default tile only (other tiles read the table from
// global memory whatever was asked), and MODE 3 (phase stamps) exists in RT_TUNING builds only.
// Everything but the default tile's culling kernels lives in the other translation units.
static RtTraceFn trace_fn(int tile_w, int cull, int mode, int table_in_lds, int feat, int multi)
{
    if (tile_w == 8) {
        if (table_in_lds) return rt_trace_fn_lds(cull, mode, feat, multi);
        return cull ? rt_trace_fn_cull8(mode, feat, multi) : rt_trace_fn_brute8(mode, feat, multi);
    }
    if (tile_w == 16 || tile_w == 32 || tile_w == 64) return rt_trace_fn_tiles(tile_w, cull, mode, feat);   // sample loop always
    return nullptr;
}

// Raise the dynamic-LDS limit of the instantiations that stage the whole table (a single
// workgroup may use the whole 160 KiB). Not a stream operation: it runs once, outside any
// stream capture or graph construction.
extern "C" hipError_t rt_dev_prepare(void)
{
    static unsigned long long done = 0;   // one bit per device: function attributes belong to a device's code object
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorNoDevice;
    if (dev < 64 && ((done >> dev) & 1ull)) return hipSuccess;
    for (int cull = 0; cull < 2; ++cull)
        for (int mode = 0; mode < 5; ++mode)
            for (int feat = 0; feat < 3; ++feat)
                for (int multi = 0; multi < 2; ++multi) {
                    const RtTraceFn fn = trace_fn(8, cull, mode, 1, feat, multi);
                    if (!fn) continue;
                    const hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (e != hipSuccess) return e;
                }
    if (dev < 64) done |= 1ull << dev;
    return hipSuccess;
}

// Everything a launch of the frame kernel needs besides its two arguments. table_in_lds is
// honoured for the default tile only. hipErrorNotSupported: no such instantiation.
extern "C" hipError_t rt_dev_trace_config(const RtFrameConsts *fc, int tile_w, int cull, int mode, int table_in_lds, int feat,
                                          const void **func, dim3 *grid, dim3 *block, unsigned *lds_bytes)
{
    if (tile_w != 8 || (feat == 2 && fc->spp > 1)) table_in_lds = 0;
    const RtTraceFn fn = trace_fn(tile_w, cull, mode, table_in_lds, feat, fc->spp > 1 ? 1 : 0);
    if (!fn) return hipErrorNotSupported;
    const int n_pad = (fc->n_spheres + 63) & ~63;
    const int wpw = table_in_lds ? RT_WAVES_PER_WG : 1;   // as WPW in the kernel
    *lds_bytes = (unsigned)((size_t)((table_in_lds ? n_pad : 0) + wpw * RT_LIST_CAP) * sizeof(float4) +
                            (size_t)wpw * RT_LIST_CAP * sizeof(int) +   // list positions (primary order)
                            (size_t)wpw * 16 * sizeof(float) +          // brightness table per wave
                            (size_t)wpw * 64 * sizeof(int) +            // marked blocks of a culling pass
                            (size_t)wpw * 16 * sizeof(double) +         // atan(k/8) per wave
                            (size_t)wpw * 192 * sizeof(float) +         // texel colours per pixel
                            (feat == 2 ? (size_t)wpw * (RT_BOX_CAP + 128) * sizeof(int) : 0));
    const int th = 64 / tile_w;
    const int wgx = (tile_w <= 16 && wpw >= 2) ? 2 : 1;
    const int wgy = wpw / wgx;
    *grid = dim3((fc->width + tile_w * wgx - 1) / (tile_w * wgx), (fc->local_rows + th * wgy - 1) / (th * wgy));
    *block = dim3(64 * wpw);
    *func = (const void *)fn;
    return rt_dev_prepare();
}

extern "C" hipError_t rt_dev_launch_trace(const RtFrameConsts *fc, const float4 *spheres, int tile_w, int cull, int mode,
                                          int table_in_lds, int feat, hipStream_t stream)
{
    const void *func = nullptr;
    dim3 grid, block;
    unsigned lds_bytes = 0;
    const hipError_t ce = rt_dev_trace_config(fc, tile_w, cull, mode, table_in_lds, feat, &func, &grid, &block, &lds_bytes);
    if (ce != hipSuccess) return ce;
    hipLaunchKernelGGL((RtTraceFn)func, grid, block, lds_bytes, stream, *fc, spheres);
    return hipGetLastError();
}

extern "C" hipError_t rt_dev_launch_dbg_shortcuts(int what, unsigned seed, long long n, unsigned long long *out, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_dbg_shortcuts, dim3(4096), dim3(256), 0, stream, what, seed, n, out);
    return hipGetLastError();
}

extern "C" hipError_t rt_dev_launch_dbg_math(int op, const float *a, const float *b, float *out, int n,
                                             hipStream_t stream)
{
    hipLaunchKernelGGL(rt_dbg_math, dim3((n + 255) / 256), dim3(256), 0, stream, op, a, b, out, n);
    return hipGetLastError();
}

extern "C" hipError_t rt_dev_launch_dbg_intersect(const float4 *tab, const float *rays, int n, int *hit,
                                                  float *t, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_dbg_intersect, dim3((n + 255) / 256), dim3(256), 0, stream, tab, rays, n, hit, t);
    return hipGetLastError();
}

extern "C" hipError_t rt_dev_launch_dbg_light(const RtFrameConsts *fc, const float4 *tab, const float *starts,
                                              const float *normals, int light_index, int n, float *dirs,
                                              float *bright, float *adirs, int *aok, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_dbg_light, dim3((n + 63) / 64), dim3(64), 0, stream, *fc, tab, starts, normals,
                       light_index, n, dirs, bright, adirs, aok);
    return hipGetLastError();
}
