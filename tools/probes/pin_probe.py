"""Probe (run by hand on the GPU box): can caller memory be page-locked, and at what cost."""
import ctypes, time, resource


def main():
    import numpy as np
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    print("memlock", resource.getrlimit(resource.RLIMIT_MEMLOCK))
    hip.hipHostRegister.argtypes=[ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
    hip.hipGetErrorString.restype=ctypes.c_char_p
    for mb in (1, 64, 512):
        a = np.ones(mb<<20, dtype=np.uint8)
        t=time.perf_counter(); r = hip.hipHostRegister(a.ctypes.data, a.size, 0); dt=time.perf_counter()-t
        print(mb, "MiB register rc", r, hip.hipGetErrorString(r), f"{dt*1e3:.2f} ms")
        if r == 0: hip.hipHostUnregister(ctypes.c_void_p(a.ctypes.data))
    p = ctypes.c_void_p()
    t=time.perf_counter(); r = hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(512<<20), 0); dt=time.perf_counter()-t
    print("hipHostMalloc 512 MiB rc", r, f"{dt*1e3:.2f} ms")
    if r == 0:
        hip.hipHostFree(p)


if __name__ == "__main__":
    main()
