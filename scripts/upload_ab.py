import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rtk_amd import api, synth
tris = synth.triangle_soup(10_000_000, 0.01, 1)
for rep in range(3):
    t0 = time.time(); ds = api.DeviceScene.build([dict(positions=tris)]); torch.cuda.synchronize(); t1 = time.time()
    print(os.environ.get("RTK_AMD_UPLOAD", "staged"), "rep", rep, "build_ms", round(ds.info()["build_ms"], 1), "wall", round((t1 - t0) * 1e3, 1), flush=True)
    ds.free()
d = torch.from_numpy(tris).cuda(); torch.cuda.synchronize()
for rep in range(3):
    ds = api.DeviceScene.build([dict(positions=d)])
    print("device-resident rep", rep, "build_ms", round(ds.info()["build_ms"], 1), flush=True)
    ds.free()
