"""GPU tests at BASELINE.json's full batch size (2^24 rays on the 1M-triangle scene), through
size-independent properties: every hit record is geometrically consistent with its triangle,
launch modes agree bit for bit, any-hit equals the closest-hit boolean, and a strided sample is
identical to the CPU oracle traversing the exported BVH."""
import numpy as np
import pytest

from rtk_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

N = 1 << 24


@pytest.fixture(scope="module")
def scene(api):
    tris = synth.scene_for_config(2)
    return tris, api.DeviceScene.build([dict(positions=tris)])


def _check_geometry(tris, rays, rec):
    hit = rec["prim"] != 0xFFFFFFFF
    tv = tris.reshape(-1, 3, 3)
    worst = 0.0
    for a in range(0, len(rec), 1 << 21):
        sl = slice(a, a + (1 << 21))
        m = hit[sl]
        r, h = rays[sl][m], rec[sl][m]
        v = tv[h["prim"]].astype(np.float64)
        u, w = h["u"].astype(np.float64)[:, None], h["v"].astype(np.float64)[:, None]
        on_tri = u * v[:, 0] + w * v[:, 1] + (1.0 - u - w) * v[:, 2]
        on_ray = r["origin"].astype(np.float64) + h["t"].astype(np.float64)[:, None] * r["direction"].astype(np.float64)
        worst = max(worst, float(np.abs(on_tri - on_ray).max()))
        assert (h["t"] > r["min_t"]).all() and (h["t"] < r["max_t"]).all()
        # barycentrics inside the triangle (the sign test of rtk.c:340-344 admits exact edges)
        assert (h["u"] >= 0).all() and (h["v"] >= 0).all() and (h["u"] + h["v"] <= 1.0 + 1e-6).all()
    assert worst < 1e-5, worst
    return int(hit.sum())


def test_coherent_full_batch(api, oracle, scene):
    tris, ds = scene
    rays = synth.rays_pinhole(4096, 4096)
    rec = ds.trace(rays, opts=api.make_opts(image=(4096, 4096)), full=False)
    nhit = _check_geometry(tris, rays, rec)
    assert abs(nhit / N - 0.875) < 0.01
    # launch modes agree on all 2^24 rays
    assert ds.trace(rays, full=False).tobytes() == rec.tobytes()
    assert ds.trace(rays, opts=api.make_opts(node_exit=1, refill_min=64), full=False).tobytes() == rec.tobytes()
    assert ds.trace(rays, opts=api.make_opts(image=(4096, 4096), no_packet=True), full=False).tobytes() == rec.tobytes()
    # any-hit over the same interval
    assert (ds.trace_any(rays) == (rec["prim"] != 0xFFFFFFFF)).all()
    # strided sample: identical to the oracle on the exported BVH
    blob = oracle.Blob(ds.export_blob())
    sel = np.arange(0, N, 64)
    oh, om = oracle.trace(blob, np.ascontiguousarray(rays[sel]))
    g = rec[sel]
    assert ((g["prim"] != 0xFFFFFFFF) == om).all()
    assert (g["prim"][om] == oh["triangle_index"][om]).all()
    assert (g["t"][om] == oh["t"][om]).all() and (g["u"][om] == oh["u"][om]).all() and (g["v"][om] == oh["v"][om]).all()


def test_incoherent_full_batch(api, scene):
    tris, ds = scene
    rays = synth.rays_incoherent(N)
    rec = ds.trace(rays, full=False)
    nhit = _check_geometry(tris, rays, rec)
    assert nhit / N > 0.99
    assert ds.trace(rays, opts=api.make_opts(static=True, node_exit=1), full=False).tobytes() == rec.tobytes()
