"""Inputs for the lab's packet simulation of config 5 (scripts/bvh_lab.cpp -pf): the 10M-triangle scene and runs of 4096 shadow
rays that are consecutive in a sorted order of the FULL 2^24-ray batch (a packet's coherence depends on the ray density of the
whole batch, so a sparse sample would not do). Keys: 'morton' = 3-D Morton code of the origin cell (8 bits per axis: what the
re-ordering pre-pass does today, only finer); 'exit' = 2-D Morton code of the point where the ray leaves the scene box (10 bits per
axis) over the depth of the origin along the ray (rays on one line towards the light land in one packet)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from rtk_amd import synth

out = sys.argv[1] if len(sys.argv) > 1 else "variants/lab"
n = 1 << 24
if not os.path.exists(os.path.join(out, "tris_10m.f32")):
    synth.triangle_soup(10_000_000, 0.01, 1).astype(np.float32).tofile(os.path.join(out, "tris_10m.f32"))
r = synth.rays_shadow(n)
o = r["origin"].astype(np.float64); d = r["direction"].astype(np.float64)

def part1by1(v):
    v = v.astype(np.uint64) & 0xffff
    v = (v | (v << 8)) & 0x00ff00ff
    v = (v | (v << 4)) & 0x0f0f0f0f
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v
def part1by2(v):
    v = v.astype(np.uint64) & 0x3ff
    v = (v | (v << 16)) & 0x030000ff
    v = (v | (v << 8)) & 0x0300f00f
    v = (v | (v << 4)) & 0x030c30c3
    v = (v | (v << 2)) & 0x09249249
    return v

keys = {}
q = np.clip((o * 256).astype(np.int64), 0, 255)
keys["morton"] = part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)
# exit from the unit box: smallest positive t over the three far planes
with np.errstate(divide="ignore", invalid="ignore"):
    tfar = np.where(d > 0, (1.0 - o) / d, np.where(d < 0, (0.0 - o) / d, np.inf))
axis = np.argmin(tfar, axis=1); te = tfar[np.arange(n), axis]
p = o + d * te[:, None]
a0 = np.where(axis == 0, 1, 0); a1 = np.where(axis == 2, 1, 2)
u = np.clip((p[np.arange(n), a0] * 1024).astype(np.int64), 0, 1023); v = np.clip((p[np.arange(n), a1] * 1024).astype(np.int64), 0, 1023)
face = axis * 2 + (d[np.arange(n), axis] > 0)
depth = np.clip((te / np.maximum(te.max(), 1e-9) * 255).astype(np.int64), 0, 255)
for bits in (10, 8, 7):
    sh = 10 - bits
    keys["exit%d" % bits] = (face.astype(np.uint64) << 40) | ((part1by1(u >> sh) | (part1by1(v >> sh) << 1)) << 8) | (255 - depth).astype(np.uint64)
rng = np.random.default_rng(1)
starts = rng.integers(0, n - 4096, size=16) // 64 * 64
for name, k in keys.items():
    order = np.argsort(k, kind="stable")
    sel = np.concatenate([order[s:s + 4096] for s in starts])
    r[sel].tofile(os.path.join(out, "rays_shadow_%s.bin" % name))
    print(name, "written", len(sel))
r[rng.integers(0, n, size=65536)].tofile(os.path.join(out, "rays_shadow_random.bin"))
