/*
 * include/rt_window.h -- the "downward" surface the frame driver calls, with
 * the exact C++ signatures of /root/reference/window.h:7-16 (minus <windows.h>).
 * librt_engine.so ships weak offscreen definitions (csrc/rt_window_offscreen.cpp);
 * an application linking its own window.cpp overrides them.
 */
#pragma once

int getScreenWidth();                     /* window.cpp:89-91  */
int getScreenHeight();                    /* window.cpp:86-88  */

void drawPixel(int x, int y, int color);  /* window.cpp:95-101 */
void setPixelBuff(unsigned int *pixels);  /* window.cpp:130-132: consumes a HOST-readable u32[W*H] */
void Set_Background();
void Clear_Screen(unsigned int color);
int make_inbound(int min, int max, int val);
int getBuffSize();
void setScreen(int *pixels);
