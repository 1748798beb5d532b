"""Why does the assembly packet kernel hand tiles back? Stack pushes past 16 entries (counting build of the C++ kernel) and
leaf sizes of the device-built config-2 scene."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtk_amd import api, synth
from oracle import pyoracle
cfg = synth.CONFIGS[2]
tris = synth.triangle_soup(cfg["num_tris"], cfg["spread"], cfg["scene_seed"])
ds = api.DeviceScene.build([dict(positions=tris)])
rays = synth.rays_pinhole(4096, 4096)
_, ctr = ds.trace_counted(rays, api.make_opts(image=(4096, 4096)))
print("counters", ctr)
blob = pyoracle.Blob(ds.export_blob())
st = blob.stats() if hasattr(blob, "stats") else None
print("blob stats", st)
