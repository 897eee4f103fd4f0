"""H2D bandwidth probe (ctypes on libamdhip64): which copy shape feeds 1080p frames fastest on this box."""
import ctypes as C
import sys
import time

hip = C.CDLL('libamdhip64.so')
vp = C.c_void_p


def chk(rc):
    if rc != 0:
        raise RuntimeError('hip error %d' % rc)


FB = 1920 * 1080
B = 32
d = vp()
chk(hip.hipMalloc(C.byref(d), C.c_size_t(FB * B + 4096 * 1080)))
pin = vp()
chk(hip.hipHostMalloc(C.byref(pin), C.c_size_t(FB * B), C.c_uint(1)))
C.memset(pin, 1, FB * B)
page = (C.c_uint8 * (FB * B))()
C.memset(page, 1, FB * B)
streams = []
for _ in range(2):
    s = vp()
    chk(hip.hipStreamCreateWithFlags(C.byref(s), C.c_uint(1)))
    streams.append(s)
hip.hipMemcpyAsync.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
hip.hipMemcpy2DAsync.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, vp]


def timeit(name, fn, reps=10):
    fn()
    chk(hip.hipDeviceSynchronize())
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    chk(hip.hipDeviceSynchronize())
    dt = (time.perf_counter() - t) / reps
    print('%-40s %7.2f GB/s  (%.0f frames/s)' % (name, FB * B / dt / 1e9, B / dt))


def one_big(src):
    return lambda: chk(hip.hipMemcpyAsync(d, src, FB * B, 1, streams[0]))


def per_frame(src, nstreams=1):
    base = C.cast(src, vp).value

    def f():
        for i in range(B):
            chk(hip.hipMemcpyAsync(vp(d.value + i * FB), vp(base + i * FB), FB, 1, streams[i % nstreams]))
    return f


def per_frame_2d(src):
    base = C.cast(src, vp).value

    def f():
        for i in range(B):
            chk(hip.hipMemcpy2DAsync(vp(d.value + i * 2048 * 1080), 2048, vp(base + i * FB), 1920, 1920, 1080, 1, streams[0]))
    return f


timeit('pinned, one 66 MB copy', one_big(pin))
timeit('pinned, 32 x 2 MB copies, 1 stream', per_frame(pin))
timeit('pinned, 32 x 2 MB copies, 2 streams', per_frame(pin, 2))
timeit('pinned, 32 x 2D copies', per_frame_2d(pin))
timeit('pageable, one 66 MB copy', one_big(page))
timeit('pageable, 32 x 2 MB copies', per_frame(page))
timeit('pageable, 32 x 2D copies', per_frame_2d(page))
