import numpy as np, sys
from rtk_amd import api, synth
N = (1 << 16) + 37
tris = synth.triangle_soup(200_000, 0.03, 7)
ds = api.DeviceScene.build([dict(positions=tris)])
rays = synth.rays_incoherent(N, seed=11)
asm = ds.trace(rays, full=False)
cpp = ds.trace(rays, opts=api.make_opts(no_asm=True), full=False)
bad = np.nonzero((asm["prim"] != cpp["prim"]) | (asm["t"].view(np.uint32) != cpp["t"].view(np.uint32)) | (asm["u"].view(np.uint32) != cpp["u"].view(np.uint32)) | (asm["v"].view(np.uint32) != cpp["v"].view(np.uint32)))[0]
print("rays", N, "mismatches", len(bad))
print("prim differs", int((asm["prim"] != cpp["prim"]).sum()), "t differs", int((asm["t"].view(np.uint32) != cpp["t"].view(np.uint32)).sum()),
      "u differs", int((asm["u"].view(np.uint32) != cpp["u"].view(np.uint32)).sum()), "v differs", int((asm["v"].view(np.uint32) != cpp["v"].view(np.uint32)).sum()))
for i in bad[:12]:
    print(i, "asm", asm[i], "cpp", cpp[i], "ray", rays[i])
asm_closer = (asm["t"][bad] < cpp["t"][bad]).sum()
print("asm closer", int(asm_closer), "cpp closer", int((asm["t"][bad] > cpp["t"][bad]).sum()))
a = ds.trace_any(synth.rays_shadow(N, seed=5)); c = ds.trace_any(synth.rays_shadow(N, seed=5), opts=api.make_opts(no_asm=True))
print("any-hit mismatches", int((a != c).sum()), "asm occluded", a.mean(), "cpp", c.mean())
