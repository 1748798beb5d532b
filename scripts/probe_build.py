import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtk_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
spread = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tris = synth.triangle_soup(n, spread, seed=seed)
ds = api.DeviceScene.build([dict(positions=tris)])
print("info", ds.info())
ok, c = ds.validate()
print("validate", ok, c)
