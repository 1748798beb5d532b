"""round 5: what the look for an un-announced image costs: rtk_dev_detect_image alone, the trace with and without the hint"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from rtk_amd import api, synth
tris = synth.scene_for_config(2)
ds = api.DeviceScene.build([dict(positions=tris)])
rays = synth.rays_pinhole(4096, 4096)
d_rays = api.to_device(rays)
n = len(rays)
out = torch.empty(n * 16, dtype=torch.uint8, device="cuda")
L = api.lib()
w, h = C.c_uint32(0), C.c_uint32(0)
def timed(fn, k=50):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / k * 1e3
print("detect alone  %.4f ms" % timed(lambda: L.rtk_dev_detect_image(ds.handle, C.c_void_p(d_rays.data_ptr()), C.c_size_t(n), C.byref(w), C.byref(h), api._stream_ptr())), w.value, h.value)
hint = api.make_opts(image=(4096, 4096))
print("trace hinted  %.4f ms" % timed(lambda: ds.trace_device(d_rays, n, out, hint)))
print("trace no hint %.4f ms" % timed(lambda: ds.trace_device(d_rays, n, out, None)))
print("trace hinted, host waits each step %.4f ms" % timed(lambda: (ds.trace_device(d_rays, n, out, hint), torch.cuda.synchronize())))
# as bench.py does it: a hinted run, then a few un-hinted steps timed one by one
for _ in range(40): ds.trace_device(d_rays, n, out, hint)
torch.cuda.synchronize()
ref = out.clone()
ds.trace_device(d_rays, n, out, None); torch.cuda.synchronize()
ts = []
for _ in range(10):
    t = time.perf_counter(); ds.trace_device(d_rays, n, out, None); ts.append((time.perf_counter() - t) * 1e3)
t = time.perf_counter(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print("per call host ms:", ["%.3f" % x for x in ts])
