import sys, subprocess, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
variants = ["impl_f32", "impl_f64", "u16_f32", "u32_f32", "u16_f64", "u32_f64"]
if len(sys.argv) == 1:
    for v in variants:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True)
        tail = [l for l in (r.stdout + r.stderr).splitlines() if "RESULT" in l or "VIOLATION" in l or "error" in l.lower()]
        print(v, "rc", r.returncode, tail[:3], flush=True)
    sys.exit(0)
from rtk_amd import api, synth
v = sys.argv[1]
tris = synth.triangle_soup(100, 0.3, seed=5)
idxk, posk = v.split("_")
pos = tris.astype(np.float64 if posk == "f64" else np.float32)
m = dict(positions=pos)
if idxk != "impl":
    m["indices"] = np.arange(300).reshape(100, 3).astype(np.uint16 if idxk == "u16" else np.uint32)
ds = api.DeviceScene.build([m])
print("RESULT", v, ds.info())
