"""Device build at sizes around every internal boundary (sort tile 4096, refit tile 1024, collapse levels of 1024 jobs,
256-job blocks): structure validates, two builds agree, and the traversal matches the oracle on the exported blob."""
import numpy as np
import pytest

from rtk_amd import synth

pytestmark = pytest.mark.gpu

SIZES = [2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 2049, 4095, 4096, 4097, 12289, 65535, 65536, 65537, 262145]


@pytest.mark.parametrize("n", SIZES)
def test_build_at_boundary_sizes(api, oracle, n):
    tris = synth.triangle_soup(n, 0.2 if n < 100 else 0.05, seed=100 + n % 97)
    ds = api.DeviceScene.build([dict(positions=tris)])
    ok, c = ds.validate()
    assert ok and c["loose_boxes"] == 0, (n, c)
    assert c["triangles_checked"] == n
    assert api.DeviceScene.build([dict(positions=tris)]).validate()[1]["content_hash"] == c["content_hash"]
    rays = synth.rays_config1(1024)
    blob = oracle.Blob(ds.export_blob())
    assert oracle.validate_blob(blob)[0] == 0
    oh, om = oracle.trace(blob, rays)
    for opts in (None, api.make_opts(image=(32, 32))):
        rec = ds.trace(rays, opts=opts, full=False)
        gm = rec["prim"] != 0xFFFFFFFF
        assert (gm == om).all() and (rec["prim"][gm] == oh["triangle_index"][om]).all(), n
        assert (rec["t"][gm] == oh["t"][om]).all() and (rec["u"][gm] == oh["u"][om]).all() and (rec["v"][gm] == oh["v"][om]).all(), n


def _degenerate_mix(seed):
    """Random scene made of the things real meshes contain and builders trip over: duplicates, points, needles,
    triangles flat in an axis plane, clusters at very different scales, one far outlier."""
    rng = np.random.RandomState(seed)
    n = int(rng.choice([37, 300, 2500, 9000]))
    base = synth.triangle_soup(n, float(rng.choice([0.3, 0.05, 0.01])), seed=seed).reshape(n, 3, 3).copy()
    k = np.arange(n)
    if rng.rand() < 0.7:
        base[k % 5 == 0] = base[0]                                   # many copies of one triangle
    if rng.rand() < 0.7:
        base[k % 7 == 1, 1] = base[k % 7 == 1, 0]; base[k % 7 == 1, 2] = base[k % 7 == 1, 0]   # points
    if rng.rand() < 0.7:
        base[k % 11 == 2, 2] = base[k % 11 == 2, 1]                   # needles (two equal vertices)
    if rng.rand() < 0.6:
        base[k % 3 == 0, :, int(rng.randint(3))] = np.float32(0.25)   # a third of the scene in one axis plane
    if rng.rand() < 0.5:
        sel = k % 13 == 3
        base[sel] = (base[sel] - np.float32(0.5)) * np.float32(1e-5) + np.float32(0.5)          # a speck of a cluster
    if rng.rand() < 0.5:
        base[-1] += np.float32(1e4)                                   # one outlier stretches the Morton grid
    if rng.rand() < 0.4:
        base[:, :, 1] = np.float32(0.0); base[:, :, 2] = np.float32(0.0)                          # everything on the x axis
    return np.ascontiguousarray(base.reshape(-1, 3).astype(np.float32))


@pytest.mark.parametrize("seed", list(range(24)))
def test_degenerate_mixes_build_and_trace_like_the_oracle_on_the_same_bvh(api, oracle, seed):
    tris = _degenerate_mix(seed)
    ds = api.DeviceScene.build([dict(positions=tris)])
    ok, c = ds.validate()
    assert ok and c["loose_boxes"] == 0, (seed, c)
    assert c["triangles_checked"] == len(tris) // 3
    blob = oracle.Blob(ds.export_blob())
    assert oracle.validate_blob(blob)[0] == 0, seed
    assert api.DeviceScene.build([dict(positions=tris)]).validate()[1]["content_hash"] == c["content_hash"]
    # the traversal does not overflow its stack and reports well-formed hits; rtk.c on the same BVH agrees wherever the
    # scene is resolvable in float (see test_adversarial_distributions for why not everywhere)
    rays = synth.rays_config1(2048)
    rec = ds.trace(rays, full=False)
    assert api.lib().rtk_dev_trace_status(ds.handle, None) == 0
    gm = rec["prim"] != 0xFFFFFFFF
    assert (rec["prim"][gm] < len(tris) // 3).all()
    # round trip: the exported blob loads again, validates, and gives the same answers ray for ray
    ds2 = api.DeviceScene.upload(blob)
    ok2, c2 = ds2.validate()
    assert ok2, (seed, c2)
    assert ds2.trace(rays, opts=api.make_opts(exact_nodes=True), full=False).tobytes() == ds.trace(rays, opts=api.make_opts(exact_nodes=True), full=False).tobytes()


def test_non_finite_vertices_do_not_break_the_build_or_the_traversal(api):
    """NaN / infinite vertex coordinates are the caller's bug, and the reference does not define what happens; here the
    build must still finish with a structurally valid tree (min/max drop NaNs) and the traversal must terminate."""
    tris = synth.triangle_soup(5000, 0.05, seed=5).reshape(5000, 3, 3).copy()
    k = np.arange(5000)
    tris[k % 97 == 0, 0, 0] = np.nan
    tris[k % 101 == 1] = np.nan
    tris[k % 103 == 2, 1, 2] = np.inf
    tris[k % 107 == 3, 2, 1] = -np.inf
    tris = np.ascontiguousarray(tris.reshape(-1, 3))
    ds = api.DeviceScene.build([dict(positions=tris)])
    ok, c = ds.validate()
    assert c["triangles_checked"] == 5000 and c["triangles_missing"] == 0 and c["triangles_duplicated"] == 0 and c["bad_references"] == 0, c
    assert c["nodes_unreachable"] == 0 and c["nodes_shared"] == 0 and c["leaf_format_errors"] == 0, c
    rays = synth.rays_config1(4096)
    for opts in (None, api.make_opts(image=(64, 64))):
        rec = ds.trace(rays, opts=opts, full=False)
        assert api.lib().rtk_dev_trace_status(ds.handle, None) == 0
        gm = rec["prim"] != 0xFFFFFFFF
        assert (rec["prim"][gm] < 5000).all() and np.isfinite(rec["t"][gm]).all()
    assert ds.trace_any(rays).dtype == bool


@pytest.mark.parametrize("counts", [(0, 5, 0, 1), (0,), (0, 0), (1,), (0, 1, 0), (3000, 0, 2), (0, 0, 4097)])
def test_scenes_with_empty_meshes(api, oracle, counts):
    """Meshes without triangles between others, scenes of one triangle, scenes of none: the build goes through, mesh and
    triangle indices keep their numbering (rtk.c:1168-1169), and an empty scene is traceable (everything misses)."""
    meshes, total = [], 0
    for mi, nt in enumerate(counts):
        meshes.append(dict(positions=synth.triangle_soup(nt, 0.3, seed=40 + mi) if nt else np.zeros((0, 3), np.float32)))
        total += nt
    ds = api.DeviceScene.build(meshes)
    info = ds.info()
    assert info["num_triangles"] == total and info["num_meshes"] == len(counts)
    rays = synth.rays_config1(1024)
    hits, mask, rec = ds.trace(rays)
    assert api.lib().rtk_dev_trace_status(ds.handle, None) == 0
    if total == 0:
        assert not mask.any()
        return
    ok, c = ds.validate()
    assert ok, c
    blob = oracle.Blob(ds.export_blob())
    assert oracle.validate_blob(blob)[0] == 0
    oh, om = oracle.trace(blob, rays)
    assert (mask == om).all()
    assert (hits["mesh_index"][mask] == oh["mesh_index"][om]).all() and (hits["triangle_index"][mask] == oh["triangle_index"][om]).all()
    nonempty = [mi for mi, nt in enumerate(counts) if nt]
    assert set(np.unique(hits["mesh_index"][mask])) <= set(nonempty)
    for opts in (api.make_opts(image=(32, 32)),):
        rec2 = ds.trace(rays, opts=opts, full=False)
        assert rec2.tobytes() == rec.tobytes()


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 255, 257, 4097])
def test_ray_batch_sizes_through_every_entry_point(api, oracle, n):
    """Batches of 0, 1 and around every chunk size through closest-hit (per-lane, packet-shaped, static, sorted, exact
    nodes), any-hit, filtered, counted and the host-pointer call: same answers as the oracle, no error left behind."""
    tris = synth.scene_for_config(1)
    ds = api.DeviceScene.build([dict(positions=tris)])
    blob = oracle.Blob(ds.export_blob())
    rays = synth.rays_config1(max(n, 1))[:n]
    oh, om = oracle.trace(blob, rays) if n else (None, np.zeros(0, bool))
    w = 1
    while w * w < max(n, 1):
        w += 1
    shapes = [None, api.make_opts(static=True), api.make_opts(sort_rays=True), api.make_opts(exact_nodes=True)]
    if n and n % w == 0:
        shapes.append(api.make_opts(image=(w, n // w)))
    for opts in shapes:
        rec = ds.trace(rays, opts=opts, full=False)
        assert len(rec) == n
        gm = rec["prim"] != 0xFFFFFFFF
        assert (gm == om).all()
        if n:
            assert (rec["prim"][gm] == oh["triangle_index"][om]).all() and (rec["t"][gm] == oh["t"][om]).all()
    assert (ds.trace_any(rays) == om).all()
    rec_f = ds.trace_filtered(rays, ignore_prim=np.full(n, 0xFFFFFFFF, np.uint32))
    assert ((rec_f["prim"] != 0xFFFFFFFF) == om).all()
    if n:
        _, ctr = ds.trace_counted(rays)
        assert ctr["rays"] == n
    assert api.lib().rtk_dev_trace_status(ds.handle, None) == 0
