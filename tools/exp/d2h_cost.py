"""What a job's result download costs by the kind of host buffer it lands in: 6.25 MB (the tags of a 50 M-read set) device -> host into a numpy
array made for the call (first touch of its pages), into one used before, and into pinned memory; the same for the upload of a selection.
  python tools/exp/d2h_cost.py"""
import ctypes as C
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
N = 6_250_001


def check(rc):
    assert rc == 0, rc


def timed(fn, reps=20):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


def main():
    check(hip.hipSetDevice(0))
    d = C.c_void_p()
    check(hip.hipMalloc(C.byref(d), C.c_size_t(N)))
    check(hip.hipMemset(d, 1, C.c_size_t(N)))
    check(hip.hipDeviceSynchronize())
    D2H, H2D = 2, 1

    def fresh_d2h():
        a = np.empty(N, np.uint8)
        check(hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), d, C.c_size_t(N), D2H))

    def fresh_zeros_d2h():
        a = np.zeros(N, np.uint8)
        check(hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), d, C.c_size_t(N), D2H))

    keep = np.zeros(N, np.uint8)
    keep[:] = 1

    def used_d2h():
        check(hip.hipMemcpy(keep.ctypes.data_as(C.c_void_p), d, C.c_size_t(N), D2H))

    def used_h2d():
        check(hip.hipMemcpy(d, keep.ctypes.data_as(C.c_void_p), C.c_size_t(N), H2D))

    p = C.c_void_p()
    t0 = time.perf_counter()
    check(hip.hipHostMalloc(C.byref(p), C.c_size_t(N), 0))
    print(f"hipHostMalloc {N / 1e6:.2f} MB: {(time.perf_counter() - t0) * 1e3:.2f} ms")

    def pinned_d2h():
        check(hip.hipMemcpy(p, d, C.c_size_t(N), D2H))

    def pinned_h2d():
        check(hip.hipMemcpy(d, p, C.c_size_t(N), H2D))

    def pinned_d2h_then_copy():
        check(hip.hipMemcpy(p, d, C.c_size_t(N), D2H))
        a = np.empty(N, np.uint8)
        C.memmove(a.ctypes.data_as(C.c_void_p), p, N)

    def alloc_only():
        a = np.zeros(N, np.uint8)
        a[::4096] = 1

    for name, fn in (("np.empty per call, D2H", fresh_d2h), ("np.zeros per call, D2H", fresh_zeros_d2h), ("array used before, D2H", used_d2h),
                     ("pinned, D2H", pinned_d2h), ("pinned D2H + memmove into np.empty", pinned_d2h_then_copy), ("np.zeros + touch alone", alloc_only),
                     ("array used before, H2D", used_h2d), ("pinned, H2D", pinned_h2d)):
        med, lo, hi = timed(fn)
        print(f"{name:40s} median {med:7.3f} ms  (min {lo:.3f}, max {hi:.3f})")


if __name__ == "__main__":
    main()
