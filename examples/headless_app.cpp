// examples/headless_app.cpp -- the reference's application shell without Win32:
// what wWinMain does (/root/reference/window.cpp:57-84: create the window, call
// onStart() once, then update() + present in a loop), with the application's OWN
// window.h functions. They are strong symbols, so they replace the weak offscreen
// versions inside librt_engine.so -- the same way the reference's window.cpp would.
//
//   g++ -std=c++17 -Iinclude examples/headless_app.cpp -Lray-tracer-engine_amd/csrc -lrt_engine \
//       -Wl,-rpath,$PWD/ray-tracer-engine_amd/csrc -o headless_app
//   ./headless_app 1920 1080 256 20 out.ppm
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_engine.h"
#include "rt_kernel.h"
#include "rt_window.h"

// ---- the application's window.cpp (window.cpp:10-15, 86-132) ----
struct Render_State {
    std::vector<unsigned int> buffmemory;
    int width = 0, height = 0;
} render;
static long present_calls = 0;

int getScreenHeight() { return render.height; }
int getScreenWidth() { return render.width; }
void setPixelBuff(unsigned int *pixels)
{
    memcpy(render.buffmemory.data(), pixels, sizeof(unsigned int) * render.width * render.height);
    ++present_calls;
}
void drawPixel(int, int, int) {}
void Set_Background() {}
void Clear_Screen(unsigned int) {}
int make_inbound(int lo, int hi, int v) { return v > hi ? hi : (v < lo ? lo : v); }
int getBuffSize() { return (int)sizeof(void *); }
void setScreen(int *) {}

int main(int argc, char **argv)
{
    const int w = argc > 1 ? atoi(argv[1]) : 640, h = argc > 2 ? atoi(argv[2]) : 360;
    const int spheres = argc > 3 ? atoi(argv[3]) : 256, frames = argc > 4 ? atoi(argv[4]) : 10;
    const char *out = argc > 5 ? argv[5] : nullptr;
    render.width = w;                         // WM_SIZE, window.cpp:29-46
    render.height = h;
    render.buffmemory.assign((size_t)w * h, 0u);

    rt_config_set_sphere_count(spheres);
    onStart();                                // window.cpp:71
    update();                                 // warm-up frame (first-touch uploads)
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < frames; ++i) update();   // window.cpp:80
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
    printf("%dx%d, %d spheres: %.3f ms/frame end to end (%.1f fps), kernel %.3f ms, %ld presents\n", w, h, spheres, ms,
           1e3 / ms, rt_last_frame_ms(), present_calls);
    if (out) {
        FILE *f = fopen(out, "wb");
        if (!f) return 2;
        fprintf(f, "P6\n%d %d\n255\n", w, h);
        for (unsigned int p : render.buffmemory) {
            const unsigned char rgb[3] = {(unsigned char)(p >> 16), (unsigned char)(p >> 8), (unsigned char)p};
            fwrite(rgb, 1, 3, f);
        }
        fclose(f);
    }
    return present_calls == frames + 1 ? 0 : 1;
}
