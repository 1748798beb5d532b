# round 5: any-hit on an image through the packet kernels: the test, then the rate on the config-2 frame with max_t = 2.2
timeout -k 10 600 python -m pytest tests/test_gpu_trace.py -m gpu -q -x -k "any_hit_on_an_image or shadow or any" 2>&1 | tail -3 || exit 1
timeout -k 10 300 python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from rtk_amd import api, synth
ds = api.DeviceScene.build([dict(positions=synth.scene_for_config(2))])
frame = synth.rays_pinhole(4096, 4096)
frame["max_t"] = np.float32(2.2)
n = len(frame)
d_rays = api.to_device(frame)
d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, opts in (("image hint (packet kernels)", api.make_opts(image=(4096, 4096))), ("no hint, no look (one ray per lane)", api.make_opts(no_detect=True)), ("no hint (the batch is looked at)", None)):
    for _ in range(3): ds.trace_any_device(d_rays, n, d_occ, opts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ds.trace_any_device(d_rays, n, d_occ, opts)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("any-hit, config-2 frame, max_t 2.2, %s: %.0f Mrays/s (%.3f ms), occluded %.3f" % (name, n / dt / 1e6, dt * 1e3, float(d_occ.float().mean())))
PY
