/*
 * include/rt_memmanager.h -- C++ mirror of the reference's device-memory
 * classes so the texture/scene producer side stays source-compatible:
 *   memManager            /root/reference/memManager.h:12-18
 *   buffer, sprite        /root/reference/sprite.h:11-47, Sprite.cpp:13-65
 * Layouts equal the C PODs of rt_engine.h (rt_buffer, rt_sprite), so a
 * `sprite*` can be handed to rt_launch_raytrace() as an `rt_sprite*`.
 */
#pragma once
#include <cstddef>
#include <string>

#include "rt_engine.h"

#define checkHipErrors(val) rt_check((int)(val), #val, __FILE__, __LINE__)

class memManager {
public:
    void *operator new(size_t len) { return rt_managed_alloc(len); }   /* memManager.cpp:12-17 */
    void operator delete(void *ptr) { rt_managed_free(ptr); }          /* memManager.cpp:18-22 */
};

class buffer : public memManager {
public:
    float *data;
    int size;
    buffer(float *pixels, int length);   /* Sprite.cpp:13-16: managed alloc + memcpy */
};

class sprite : public memManager {
public:
    /* Sprite.cpp:28-52 decodes with OpenCV; here `file` is a binary PPM (P6).
     * The special names "synthetic:object" and "synthetic:sky" produce the
     * deterministic stand-ins for wood.jpg / sky_box.jpg (rt_synth_texture). */
    explicit sprite(std::string file);
    int getBytes();                      /* Sprite.cpp:53-56 */

    buffer *rBuff;
    buffer *gBuff;
    buffer *bBuff;

    int width;
    int height;
};
