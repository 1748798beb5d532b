import numpy as np, sys
from rtk_amd import api, synth
N = (1 << 19) + 37
tris = synth.triangle_soup(200_000, 0.03, 7)
ds = api.DeviceScene.build([dict(positions=tris)])
rays = synth.rays_incoherent(N, seed=12)
ex = synth.rays_exotic(2048, seed=9, tris=tris.reshape(-1, 3, 3))
rays[::257][:len(ex)] = ex[:len(rays[::257])]
for rep in range(2):
    asm = ds.trace(rays, full=False)
    cpp = ds.trace(rays, opts=api.make_opts(no_asm=True), full=False)
    bad = np.nonzero(asm.view(np.uint32).reshape(-1, 4) != cpp.view(np.uint32).reshape(-1, 4))[0]
    bad = np.unique(bad)
    print("rep", rep, "mismatching rays", len(bad), "of which exotic slots", int((bad % 257 == 0).sum()))
    for i in bad[:16]:
        print(i, i % 257 == 0, "asm", asm[i], "cpp", cpp[i], "ray", rays[i])
