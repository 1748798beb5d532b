import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from rtk_amd import api, synth
tris = synth.scene_for_config(2)
ds = api.DeviceScene.build([dict(positions=tris)])
n = 4096 * 4096
rays = api.to_device(synth.rays_pinhole(4096, 4096))
out = torch.empty(n * 16, dtype=torch.uint8, device="cuda")
opts = api.make_opts(image=(4096, 4096))
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
t0 = time.perf_counter()
for a, b in ev:
    a.record(); ds.trace_device(rays, n, out, opts); b.record()
torch.cuda.synchronize()
print("wall per step %.4f ms" % ((time.perf_counter() - t0) / 40 * 1e3))
print(" ".join("%.3f" % a.elapsed_time(b) for a, b in ev))
