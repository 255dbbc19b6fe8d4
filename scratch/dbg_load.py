import sys, ctypes, os
sys.path.insert(0, '.')
order = sys.argv[1]
if order == "torch_first":
    import torch
    x = torch.zeros(4, device="cuda:0"); torch.cuda.synchronize()
    print("torch hip:", torch.version.hip)
hip = None
from brotli_amd import m4
L = m4.lib()
print("device_count:", L.m4g_device_count())
# call hipGetDeviceCount through whatever runtime our lib is bound to
rt = ctypes.CDLL("libamdhip64.so.7")
n = ctypes.c_int(-1)
rc = rt.hipGetDeviceCount(ctypes.byref(n)); print("hipGetDeviceCount rc", rc, "n", n.value)
v = ctypes.c_int(0); rt.hipRuntimeGetVersion(ctypes.byref(v)); print("runtime version", v.value)
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "libamdhip64" in l or "libhsa-runtime" in l})
print(libs)
try:
    r = m4.M4().find_matches(b"hello world hello world hello world")
    print(r.tolist())
except Exception as e:
    print("ERR", e)
if order != "torch_first":
    import torch
    x = torch.zeros(4, device="cuda:0"); torch.cuda.synchronize(); print("torch after ok", torch.version.hip)
