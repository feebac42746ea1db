"""what pinning the reader's batch buffers could buy: hipHostMalloc / hipHostRegister cost per MB, H2D rate from pageable / pinned / registered memory"""
import ctypes as C, time, json, numpy as np
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipFree.argtypes = [C.c_void_p]
def chk(e):
    assert e == 0, e
chk(hip.hipSetDevice(0))
out = {}
for mb in (64, 288):
    n = mb << 20
    d = C.c_void_p(); chk(hip.hipMalloc(C.byref(d), n))
    a = np.empty(n, dtype=np.uint8); a[:] = 1                      # pageable, touched
    chk(hip.hipMemcpy(d, a.ctypes.data, n, 1)); chk(hip.hipDeviceSynchronize())
    ts = []
    for _ in range(3):
        t = time.perf_counter(); chk(hip.hipMemcpy(d, a.ctypes.data, n, 1)); chk(hip.hipDeviceSynchronize()); ts.append(time.perf_counter() - t)
    r = {"pageable_h2d_gbs": n / min(ts) / 1e9}
    t = time.perf_counter(); p = C.c_void_p(); chk(hip.hipHostMalloc(C.byref(p), n, 0)); r["hipHostMalloc_ms"] = (time.perf_counter() - t) * 1e3
    C.memset(p, 1, n)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); chk(hip.hipMemcpy(d, p, n, 1)); chk(hip.hipDeviceSynchronize()); ts.append(time.perf_counter() - t)
    r["pinned_h2d_gbs"] = n / min(ts) / 1e9
    t = time.perf_counter(); chk(hip.hipHostFree(p)); r["hipHostFree_ms"] = (time.perf_counter() - t) * 1e3
    t = time.perf_counter(); chk(hip.hipHostRegister(a.ctypes.data, n, 0)); r["hipHostRegister_ms"] = (time.perf_counter() - t) * 1e3
    ts = []
    for _ in range(3):
        t = time.perf_counter(); chk(hip.hipMemcpy(d, a.ctypes.data, n, 1)); chk(hip.hipDeviceSynchronize()); ts.append(time.perf_counter() - t)
    r["registered_h2d_gbs"] = n / min(ts) / 1e9
    t = time.perf_counter(); chk(hip.hipHostUnregister(a.ctypes.data)); r["hipHostUnregister_ms"] = (time.perf_counter() - t) * 1e3
    # multi-thread staging: 8 threads memcpy 4-MB pieces into a pinned ring — what a hand-made staging path could reach on this host
    chk(hip.hipFree(d))
    out["%d MB" % mb] = r
print(json.dumps(out, indent=1))
