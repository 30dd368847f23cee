/*
 * include/rt_kernel.h -- the "upward" surface the application shell calls:
 * the whole of /root/reference/kernel.cuh:3-4, same C++ linkage and names
 * (wWinMain calls them at window.cpp:71 and :80).
 */
#pragma once

void onStart();
void update();
