"""Deterministic synthetic scenes and ray batches of BASELINE.json's configs.

Counter-based generator: value k of stream `seed` is
splitmix64((seed << 40) + k); its top 24 bits times 2^-24 is a float32 in [0,1) that is
exact, and every later operation is one correctly rounded float32 add/mul, so numpy here
and any C/HIP restatement produce bit-identical inputs.
"""
import numpy as np

from .types import RAY_DTYPE, RTK_INF

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def u01(seed, first, count):
    """float32 U[0,1) values number first..first+count-1 of stream `seed`."""
    k = np.arange(first, first + count, dtype=np.uint64)
    h = splitmix64((np.uint64(seed) << np.uint64(40)) + k)
    return ((h >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


def triangle_soup(num_tris, spread, seed=1, chunk=1 << 20):
    """Unindexed float32 mesh [3*num_tris, 3]: centre ~U[0,1)^3, vertex = centre + spread*(U-0.5)."""
    out = np.empty((num_tris, 3, 3), dtype=np.float32)
    sp = np.float32(spread)
    for a in range(0, num_tris, chunk):
        n = min(chunk, num_tris - a)
        u = u01(seed, a * 12, n * 12).reshape(n, 12)
        centre = u[:, 0:3]
        off = (u[:, 3:12] - np.float32(0.5)) * sp
        out[a:a + n] = centre[:, None, :] + off.reshape(n, 3, 3)
    return out.reshape(num_tris * 3, 3)


def _rays(n):
    r = np.zeros(n, dtype=RAY_DTYPE)
    r["min_t"] = np.float32(0.0)
    r["max_t"] = RTK_INF
    return r


def rays_config1(n=65536, seed=2):
    """origin (U,U,-1), direction (0.3(U-.5), 0.3(U-.5), 1)."""
    u = u01(seed, 0, n * 4).reshape(n, 4)
    r = _rays(n)
    r["origin"][:, 0] = u[:, 0]
    r["origin"][:, 1] = u[:, 1]
    r["origin"][:, 2] = np.float32(-1.0)
    r["direction"][:, 0] = np.float32(0.3) * (u[:, 2] - np.float32(0.5))
    r["direction"][:, 1] = np.float32(0.3) * (u[:, 3] - np.float32(0.5))
    r["direction"][:, 2] = np.float32(1.0)
    return r


def rays_pinhole(width=4096, height=4096, first=0, count=None, jitter=(0.5, 0.5)):
    """Row-major pinhole frame from (0.5,0.5,-1.5), fov factor 0.7 (config 2)."""
    total = width * height
    if count is None:
        count = total - first
    i = np.arange(first, first + count, dtype=np.int64)
    x = (i % width).astype(np.float32)
    y = (i // width).astype(np.float32)
    r = _rays(count)
    r["origin"][:] = np.array([0.5, 0.5, -1.5], dtype=np.float32)
    jx, jy = np.float32(jitter[0]), np.float32(jitter[1])
    r["direction"][:, 0] = ((x + jx) / np.float32(width) - np.float32(0.5)) * np.float32(0.7)
    r["direction"][:, 1] = ((y + jy) / np.float32(height) - np.float32(0.5)) * np.float32(0.7)
    r["direction"][:, 2] = np.float32(1.0)
    return r


def frame_jitter(frame, seed=4):
    """Sub-pixel offset of frame `frame` of config 4; frame 0 is config 2 itself."""
    if frame == 0:
        return (0.5, 0.5)
    u = u01(seed, 2 * frame, 2)
    return (float(u[0]), float(u[1]))


def rays_incoherent(n, seed=3, first=0):
    """origin ~U[-0.5,1.5)^3, direction = target ~U[0,1)^3 minus origin (config 3)."""
    u = u01(seed, first * 6, n * 6).reshape(n, 6)
    r = _rays(n)
    o = u[:, 0:3] * np.float32(2.0) - np.float32(0.5)
    r["origin"] = o
    r["direction"] = u[:, 3:6] - o
    return r


def rays_shadow(n, seed=5, first=0, light=(0.5, 2.0, 0.5)):
    """origin ~U[0,1)^3, direction = light - origin, t in (1e-4, 1) (config 5)."""
    u = u01(seed, first * 3, n * 3).reshape(n, 3)
    r = _rays(n)
    r["origin"] = u
    r["direction"] = np.array(light, dtype=np.float32)[None, :] - u
    r["min_t"] = np.float32(1e-4)
    r["max_t"] = np.float32(1.0)
    return r


CONFIGS = {
    1: dict(num_tris=10_000, spread=0.05, scene_seed=1),
    2: dict(num_tris=1_000_000, spread=0.02, scene_seed=1),
    3: dict(num_tris=1_000_000, spread=0.02, scene_seed=1),
    4: dict(num_tris=1_000_000, spread=0.02, scene_seed=1),
    5: dict(num_tris=10_000_000, spread=0.01, scene_seed=1),
}


def scene_for_config(cfg):
    c = CONFIGS[cfg]
    return triangle_soup(c["num_tris"], c["spread"], c["scene_seed"])


def rays_exotic(n=2048, seed=9, tris=None):
    """Rays full of the values that break sloppy implementations: zero, negative-zero, denormal, tiny and huge direction
    components, equal-magnitude components (dominant-axis ties), origins far away, on vertex coordinates or in triangle
    planes, empty / reversed / negative parameter intervals, NaN min_t (a NaN or infinite max_t makes the real
    rtk.c run off its traversal stack -- rtk.c:477 asserts; its own "no limit" is FLT_MAX -- so those are not part of the
    comparable domain). Deterministic; used against the real rtk.c
    (oracle/gen_golden.py --only exotic) and against the GPU kernels."""
    u = u01(seed, 0, n * 12).reshape(n, 12)
    r = _rays(n)
    o = u[:, 0:3] * np.float32(1.6) - np.float32(0.3)
    d = u[:, 3:6] - np.float32(0.5)
    k = np.arange(n)
    specials = np.array([0.0, -0.0, 1e-42, -1e-42, 1e-30, 1e30, -1e30, 1.0, -1.0, 0.5], dtype=np.float32)
    for axis in range(3):
        sel = (u[:, 6 + axis] < np.float32(0.35))
        pick = (u[:, 9 + axis] * np.float32(len(specials))).astype(np.int64) % len(specials)
        d[sel, axis] = specials[pick[sel]]
    d[k % 29 == 0] = np.float32([1.0, 1.0, 1.0]) * (u[k % 29 == 0, 3:4] - np.float32(0.5))      # |dx| = |dy| = |dz|
    d[k % 31 == 0, 1] = d[k % 31 == 0, 0]                                                            # |dx| = |dy|
    d[k % 211 == 0] = 0.0                                                                             # no direction at all
    o[k % 37 == 0, 2] = np.float32(-2.5e7)
    o[k % 41 == 0, 0] = np.float32(1e30)
    if tris is not None:
        t3 = np.asarray(tris, np.float32).reshape(-1, 3, 3)
        sel = k % 17 == 0
        o[sel] = t3[(k[sel] * 7) % len(t3), 0]                                                        # exactly on a vertex
        sel = k % 19 == 0
        o[sel, 0] = t3[(k[sel] * 11) % len(t3), 1, 0]                                                 # one coordinate of a vertex
    r["origin"] = o
    r["direction"] = d
    r["min_t"] = 0.0
    r["max_t"] = np.float32(3.402823e38)
    r["min_t"][k % 13 == 0] = np.float32(-1.0)
    r["min_t"][k % 23 == 0] = np.float32(0.75)
    r["max_t"][k % 7 == 0] = np.float32(1.0)
    r["max_t"][k % 43 == 0] = np.float32(0.0)             # empty interval
    r["max_t"][k % 47 == 0] = np.float32(-1.0)            # reversed interval
    r["min_t"][k % 53 == 0] = np.float32("nan")
    return r


# ---- the same generators on a torch device (bit-identical values: integer arithmetic modulo 2^64, then exactly rounded
# float32 operations). bench.py uses them for the large batches, where numpy on the host costs tens of seconds.
def _t_splitmix64(x):
    import torch

    def lsr(v, k):      # logical shift right of int64
        return (v >> k) & ((1 << (64 - k)) - 1)
    c1, c2, c3 = (0x9E3779B97F4A7C15 - (1 << 64)), (0xBF58476D1CE4E5B9 - (1 << 64)), (0x94D049BB133111EB - (1 << 64))
    z = x + c1
    z = (z ^ lsr(z, 30)) * c2
    z = (z ^ lsr(z, 27)) * c3
    return z ^ lsr(z, 31)


def t_u01(seed, first, count, device):
    import torch
    k = torch.arange(first, first + count, dtype=torch.int64, device=device)
    h = _t_splitmix64((int(seed) << 40) + k)
    return ((h >> 40) & 0xFFFFFF).to(torch.float32) * float(2.0 ** -24)


def t_triangle_soup(num_tris, spread, seed=1, device="cuda", chunk=1 << 22):
    """triangle_soup on a torch device: float32 tensor [3 * num_tris, 3]."""
    import torch
    out = torch.empty((num_tris, 3, 3), dtype=torch.float32, device=device)
    sp = torch.tensor(spread, dtype=torch.float32, device=device)
    for a in range(0, num_tris, chunk):
        n = min(chunk, num_tris - a)
        u = t_u01(seed, a * 12, n * 12, device).reshape(n, 12)
        off = (u[:, 3:12] - 0.5) * sp
        out[a:a + n] = u[:, 0:3][:, None, :] + off.reshape(n, 3, 3)
    return out.reshape(num_tris * 3, 3)


def _t_rays(o, d, tmin, tmax):
    import torch
    n = o.shape[0]
    r = torch.empty((n, 8), dtype=torch.float32, device=o.device)
    r[:, 0:3] = o
    r[:, 3:6] = d
    r[:, 6] = tmin
    r[:, 7] = tmax
    return r


def t_rays_incoherent(n, seed=3, first=0, device="cuda"):
    import torch
    u = t_u01(seed, first * 6, n * 6, device).reshape(n, 6)
    o = u[:, 0:3] * 2.0 - 0.5
    return _t_rays(o, u[:, 3:6] - o, 0.0, float(RTK_INF))


def t_rays_shadow(n, seed=5, first=0, light=(0.5, 2.0, 0.5), device="cuda"):
    import torch
    u = t_u01(seed, first * 3, n * 3, device).reshape(n, 3)
    L = torch.tensor(light, dtype=torch.float32, device=device)
    return _t_rays(u, L[None, :] - u, float(np.float32(1e-4)), 1.0)
