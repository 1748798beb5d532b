"""GPU tests of the LDS ray-pool kernel (rtk_trace_pool.hip, an option: RTK_TRACE_POOL): what it writes must be what rtk_trace_kernel writes, byte for
byte, for closest-hit and any-hit batches, in given and re-ordered ray order, at batch sizes that are not whole chunks,
with rays it must leave to rtk_trace_kernel (non-finite or zero components) in the batch, and against the CPU oracle on the
exported BVH."""
import numpy as np
import pytest

from rtk_amd import synth

pytestmark = pytest.mark.gpu

N = (1 << 19) + 37          # above the pool's minimum batch (2^18), not a multiple of 64


@pytest.fixture(autouse=True, params=["sticky", "queued"])
def pool_form(request, monkeypatch):
    """Both forms of the pool kernel: rays that stay in their lane while they are in node state (default), and every ray
    through the queues on every trip (RTK_AMD_POOL_STICKY=0)."""
    monkeypatch.setenv("RTK_AMD_POOL_STICKY", "1" if request.param == "sticky" else "0")
    return request.param


@pytest.fixture(scope="module")
def scene(api):
    tris = synth.triangle_soup(200_000, 0.03, 7)
    return tris, api.DeviceScene.build([dict(positions=tris)])


def test_pool_equals_bound_lanes_closest_hit(api, oracle, scene):
    tris, ds = scene
    rays = synth.rays_incoherent(N, seed=11)
    pool = ds.trace(rays, opts=api.make_opts(pool=True), full=False)
    lanes = ds.trace(rays, full=False)
    assert pool.tobytes() == lanes.tobytes()
    assert (pool["prim"] != 0xFFFFFFFF).mean() > 0.9
    # re-ordered by entry cell: the same records in the rays' own slots
    assert ds.trace(rays, opts=api.make_opts(sort_rays=True, pool=True), full=False).tobytes() == lanes.tobytes()
    # ... and the oracle on the exported BVH agrees on a sample
    blob = oracle.Blob(ds.export_blob())
    sel = np.arange(0, N, 97)
    oh, om = oracle.trace(blob, np.ascontiguousarray(rays[sel]))
    g = pool[sel]
    assert ((g["prim"] != 0xFFFFFFFF) == om).all()
    assert (g["prim"][om] == oh["triangle_index"][om]).all()
    assert (g["t"][om] == oh["t"][om]).all() and (g["u"][om] == oh["u"][om]).all() and (g["v"][om] == oh["v"][om]).all()


def test_pool_equals_bound_lanes_any_hit(api, scene):
    tris, ds = scene
    rays = synth.rays_shadow(N, seed=12)
    pool = ds.trace_any(rays, opts=api.make_opts(pool=True))
    lanes = ds.trace_any(rays)
    assert (pool == lanes).all()
    assert 0.05 < pool.mean() < 1.0
    assert (ds.trace_any(rays, opts=api.make_opts(sort_rays=True, pool=True)) == lanes).all()
    # any-hit over the same interval is the closest-hit boolean
    assert (pool == (ds.trace(rays, full=False)["prim"] != 0xFFFFFFFF)).all()


def test_pool_leaves_special_rays_to_the_exact_path(api, scene):
    """Rays full of zeros, denormals, infinities and NaN intervals between ordinary ones: the pool kernel hands them to
    rtk_trace_kernel's exact-node path, and every ray's record is what that kernel alone writes."""
    tris, ds = scene
    rays = synth.rays_incoherent(N, seed=13)
    ex = synth.rays_exotic(2048, seed=9, tris=tris)
    at = np.arange(2048) * 211 + 5
    rays[at] = ex
    pool = ds.trace(rays, opts=api.make_opts(pool=True), full=False)
    lanes = ds.trace(rays, full=False)
    assert pool.tobytes() == lanes.tobytes()
    assert (ds.trace_any(rays, opts=api.make_opts(pool=True)) == ds.trace_any(rays)).all()


def test_pool_on_a_deep_tree(api):
    """A scene whose LBVH is deep (triangles on a line with shrinking spacing): stacks outgrow the pool's 8 LDS entries and
    use the launch's spill area."""
    n = 60_000
    k = np.arange(n, dtype=np.float64)
    x = (1.0 - 0.9997 ** k).astype(np.float32)
    c = np.stack([x, np.full(n, 0.5, np.float32), np.full(n, 0.5, np.float32)], axis=1)
    off = (synth.u01(21, 0, n * 9).reshape(n, 3, 3) - np.float32(0.5)) * np.float32(0.02)
    tris = (c[:, None, :] + off).astype(np.float32).reshape(n * 3, 3)
    ds = api.DeviceScene.build([dict(positions=tris)])
    rays = synth.rays_incoherent(N, seed=14)
    pool = ds.trace(rays, opts=api.make_opts(pool=True), full=False)
    assert pool.tobytes() == ds.trace(rays, full=False).tobytes()
    assert (pool["prim"] != 0xFFFFFFFF).any()
