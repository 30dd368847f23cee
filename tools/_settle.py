"""Untimed pre-roll for the timing tools: run `step()` in windows of `window` calls (a host synchronise after each
window) until two consecutive windows agree within `tol` -- a GPU that idled while the host built tables or captured a
graph takes a few hundred launches to get back to its clock (first windows of a fresh process: 4x slower)."""
import time


def settle(step, sync, window=10, tol=0.02, max_windows=60):
    hist = []
    for _ in range(max_windows):
        t0 = time.perf_counter()
        for _ in range(window):
            step()
        sync()
        hist.append((time.perf_counter() - t0) / window)
        if len(hist) >= 3 and abs(hist[-1] - hist[-2]) <= tol * hist[-1] and abs(hist[-2] - hist[-3]) <= tol * hist[-2]:
            break
    return hist
