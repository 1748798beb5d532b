"""Stage timings of the device build (RTK_AMD_BUILD_TIMING=1 prints one line per stage, with a device
synchronisation between stages) and the end-to-end figure without those synchronisations, for 1M and 10M
triangles held in HBM. Usage: python scripts/build_timing.py [sizes...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from rtk_amd import api, synth  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 10_000_000]
for n in sizes:
    tris = synth.triangle_soup(n, 0.02 if n <= 1_000_000 else 0.01, seed=1)
    d = torch.from_numpy(tris).cuda()
    torch.cuda.synchronize()
    ms = []
    for rep in range(5):
        ds = api.DeviceScene.build([dict(positions=d)])
        ms.append(ds.info()["build_ms"])
        nodes = ds.info()["num_nodes"]
        ds.free()
    print("n=%d device-resident build_ms %s nodes %d" % (n, ["%.3f" % m for m in ms], nodes), flush=True)
    os.environ["RTK_AMD_BUILD_TIMING"] = "1"
    ds = api.DeviceScene.build([dict(positions=d)])
    ds.free()
    del os.environ["RTK_AMD_BUILD_TIMING"]
    t0 = time.time()
    ds = api.DeviceScene.build([dict(positions=tris)])
    print("n=%d host-memory build_ms %.3f (wall %.3f)" % (n, ds.info()["build_ms"], (time.time() - t0) * 1e3), flush=True)
    ds.free()
