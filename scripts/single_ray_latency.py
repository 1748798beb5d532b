"""Latency of the reference's per-ray entry points on the GPU path: rtk_trace_ray (rtk.h:129) in a loop, from one and
from several host threads, and rtk_trace_rays for small batches. Usage: python scripts/single_ray_latency.py"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from rtk_amd import api, synth  # noqa: E402

tris = synth.scene_for_config(1)
scene, keep = api.build_scene([dict(positions=tris)])
rays = synth.rays_config1(1048576)
api.trace_ray(scene, rays[0])
n = 3000
t0 = time.perf_counter()
hits = 0
for i in range(n):
    hits += api.trace_ray(scene, rays[i]) is not None
dt = time.perf_counter() - t0
print("rtk_trace_ray: %.1f us per call (%d calls, %d hits), one thread" % (dt / n * 1e6, n, hits), flush=True)


def worker(k, out):
    t0 = time.perf_counter()
    for i in range(k * 1000, k * 1000 + 1000):
        api.trace_ray(scene, rays[i])
    out[k] = time.perf_counter() - t0


out = {}
ts = [threading.Thread(target=worker, args=(k, out)) for k in range(4)]
t0 = time.perf_counter()
[t.start() for t in ts]
[t.join() for t in ts]
dt = time.perf_counter() - t0
print("rtk_trace_ray: 4 threads x 1000 calls in %.1f ms = %.1f us per call aggregate" % (dt * 1e3, dt / 4000 * 1e6), flush=True)
for m in (64, 1024, 16384, 65536, 1048576):
    api.trace_rays(scene, rays[:m])
    t0 = time.perf_counter()
    for _ in range(20):
        api.trace_rays(scene, rays[:m])
    dt = (time.perf_counter() - t0) / 20
    print("rtk_trace_rays(%d rays): %.1f us per call = %.2f Mrays/s (PCIe inclusive, full 68-byte hits)" % (m, dt * 1e6, m / dt / 1e6), flush=True)
api.free_scene(scene)
