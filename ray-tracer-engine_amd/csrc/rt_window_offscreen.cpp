// rt_window_offscreen.cpp -- headless stand-in for the reference's Win32 present
// path (/root/reference/window.cpp:86-132, window.h:7-16): the same functions,
// over an offscreen host buffer instead of a DIB section.
//
// The C++ symbols are WEAK so that an application which links its own
// window.cpp (the real Win32 one, or any other presenter) overrides them.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/rt_engine.h"
#include "../../include/rt_window.h"

void rt_set_error(const char *fmt, ...);

namespace {
struct Render_State {   // window.cpp:10-15
    void *buffmemory = nullptr;
    int width = 0, height = 0;
};
Render_State render;
}  // namespace

#define RT_WEAK __attribute__((weak))

namespace {
inline size_t frame_words() { return (size_t)render.width * (size_t)render.height; }
inline unsigned int *frame() { return static_cast<unsigned int *>(render.buffmemory); }
}  // namespace

// The three functions the frame driver calls (kernel.cu:1771, 1788).
RT_WEAK int getScreenHeight() { return render.height; }
RT_WEAK int getScreenWidth() { return render.width; }
// setPixelBuff consumes a HOST-readable frame of getScreenWidth()*getScreenHeight() packed
// words (window.cpp:130-132) and makes it the presented image.
RT_WEAK void setPixelBuff(unsigned int *pixels)
{
    if (frame() && pixels) std::copy_n(pixels, frame_words(), frame());
}

// The rest of window.h:7-16 is never called by the hot path. It is defined so that an
// application which includes window.h resolves every symbol and OBSERVES what the reference's
// helpers do (window.cpp:95-129): the background pattern word(x, y) = y*x/(x+1) in int
// arithmetic, a clear to one colour, a clamped single-pixel store, and getBuffSize() returning
// the size of the buffer POINTER member (sizeof(render.buffmemory), 8 on x64 -- not the frame's
// byte count).
RT_WEAK int make_inbound(int min, int max, int val) { return std::min(std::max(val, min), max); }
RT_WEAK void Clear_Screen(unsigned int color)
{
    if (frame()) std::fill_n(frame(), frame_words(), color);
}
RT_WEAK void Set_Background()
{
    if (!frame()) return;
    const size_t w = (size_t)render.width;
    for (size_t i = 0, n = frame_words(); i < n; ++i) {
        const int x = (int)(i % w), y = (int)(i / w);
        frame()[i] = (unsigned int)(y * x / (x + 1));
    }
}
RT_WEAK void drawPixel(int x, int y, int color)
{
    if (!frame() || render.width <= 0 || render.height <= 0) return;
    const int cx = make_inbound(0, render.width - 1, x), cy = make_inbound(0, render.height - 1, y);
    frame()[(size_t)cy * (size_t)render.width + (size_t)cx] = (unsigned int)color;
}
RT_WEAK int getBuffSize() { return (int)sizeof render.buffmemory; }   // window.cpp:118-120: the pointer's size
RT_WEAK void setScreen(int *) {}

// ---- control surface of the offscreen window (C ABI) ----
// WM_SIZE (window.cpp:29-46) re-allocates the present buffer; the reference
// halves the client size there, here the caller passes the render size itself.
extern "C" int rt_offscreen_resize(int width, int height)
{
    if (width <= 0 || height <= 0) {
        rt_set_error("rt_offscreen_resize: bad size %d x %d", width, height);
        return RT_ERR_INVALID;
    }
    free(render.buffmemory);
    render.buffmemory = calloc((size_t)width * (size_t)height, sizeof(unsigned int));
    if (!render.buffmemory) {
        render.width = render.height = 0;
        rt_set_error("rt_offscreen_resize: out of memory");
        return RT_ERR_INVALID;
    }
    render.width = width;
    render.height = height;
    return RT_OK;
}
extern "C" const uint32_t *rt_offscreen_pixels(void) { return (const uint32_t *)render.buffmemory; }
extern "C" int rt_offscreen_width(void) { return render.width; }
extern "C" int rt_offscreen_height(void) { return render.height; }

extern "C" int rt_offscreen_write_ppm(const char *path)
{
    if (!path || !render.buffmemory) return RT_ERR_INVALID;
    FILE *f = fopen(path, "wb");
    if (!f) {
        rt_set_error("rt_offscreen_write_ppm: cannot create '%s'", path);
        return RT_ERR_INVALID;
    }
    fprintf(f, "P6\n%d %d\n255\n", render.width, render.height);
    const uint32_t *p = (const uint32_t *)render.buffmemory;
    for (size_t i = 0, n = (size_t)render.width * render.height; i < n; ++i) {
        const unsigned char rgb[3] = {(unsigned char)((p[i] >> 16) & 0xff), (unsigned char)((p[i] >> 8) & 0xff),
                                      (unsigned char)(p[i] & 0xff)};
        fwrite(rgb, 1, 3, f);
    }
    fclose(f);
    return RT_OK;
}
