"""GPU tests of the device LBVH builder (rtk_dev_scene_build / rtk_build_scene and the task
API) -- structure through the exported blob, results against the golden fixtures from the
real reference and bit-for-bit against the CPU oracle traversing the exported blob."""
import ctypes as C

import numpy as np
import pytest

from rtk_amd import synth
from rtk_amd.types import RAY_DTYPE, Task
from tests.util import compare_hits, compare_hits_struct, load_golden

pytestmark = pytest.mark.gpu


def _as_blob(oracle, arr):
    b = oracle._aligned_bytes(arr.size)
    b[:] = arr
    return oracle.Blob(b)


def _same_as_oracle(oracle, blob, ds, rays, what):
    hits, mask, rec = ds.trace(rays)
    ohits, omask = oracle.trace(blob, rays)
    st = compare_hits(mask, hits["mesh_index"], hits["triangle_index"], hits["t"], hits["u"], hits["v"],
                      omask, ohits["mesh_index"], ohits["triangle_index"], ohits["t"], ohits["u"], ohits["v"], what)
    assert st["bit_exact"] == 1.0, what
    assert (hits["vertex"]["index"][mask] == ohits["vertex"]["index"][omask]).all()
    return hits, mask, rec


def test_config1_device_build(api, oracle, golden_dir):
    tris = synth.scene_for_config(1)
    ds = api.DeviceScene.build([dict(positions=tris)])
    info = ds.info()
    assert info["num_triangles"] == 10000 and info["num_nodes"] > 100 and 2 <= info["max_depth"] < 40
    blob = _as_blob(oracle, ds.export_blob())
    rc, counts = oracle.validate_blob(blob)
    assert rc == 0 and counts["tris"] == 10000 and counts["nodes"] == info["num_nodes"]
    rays = synth.rays_config1(65536)
    hits, mask, rec = _same_as_oracle(oracle, blob, ds, rays, "lbvh vs oracle on exported blob")
    compare_hits_struct(hits, mask, load_golden(golden_dir, "cfg1_full.npz"), "lbvh vs reference fixture")
    # the exported blob uploaded again gives the same BVH
    ds2 = api.DeviceScene.upload(blob)
    assert ds2.trace(rays, full=False).tobytes() == rec.tobytes()


def test_every_triangle_is_hit_by_its_own_ray(api):
    """Completeness of the build: a ray aimed at each triangle's centroid from just in front of it
    along its normal must report a hit no farther than that triangle."""
    tris = synth.scene_for_config(1).reshape(-1, 3, 3).astype(np.float64)
    ds = api.DeviceScene.build([dict(positions=synth.scene_for_config(1))])
    c = tris.mean(axis=1)
    nrm = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    rays = np.zeros(len(tris), RAY_DTYPE)
    rays["origin"] = (c + 1e-3 * nrm).astype(np.float32)
    rays["direction"] = (-nrm).astype(np.float32)
    rays["min_t"] = 0
    rays["max_t"] = 1.0
    rec = ds.trace(rays, full=False)
    assert (rec["prim"] != 0xFFFFFFFF).all()
    assert (rec["t"] <= 1.01e-3).all()
    own = rec["prim"] == np.arange(len(tris))
    assert own.mean() > 0.99        # the rest are overlapping neighbours that are even closer


def test_edge_scene_indexed_multi_mesh(api, oracle, golden_dir):
    """Mesh 0 as u16-indexed float64 positions, mesh 1 as u32-indexed float32: same hits as the fixture."""
    g = load_golden(golden_dir, "edge_cases.npz")
    rays = np.ascontiguousarray(g["rays"]).view(RAY_DTYPE).reshape(-1)
    t0 = g["tris"][g["mesh"] == 0].reshape(-1, 3)
    t1 = g["tris"][g["mesh"] == 1].reshape(-1, 3)
    v0, inv0 = np.unique(t0, axis=0, return_inverse=True)
    v1, inv1 = np.unique(t1, axis=0, return_inverse=True)
    meshes = [dict(positions=v0.astype(np.float64), indices=inv0.reshape(-1, 3).astype(np.uint16)),
              dict(positions=v1.astype(np.float32), indices=inv1.reshape(-1, 3).astype(np.uint32))]
    ds = api.DeviceScene.build(meshes)
    assert list(ds.mesh_base()) == [0, 8, 10]
    blob = _as_blob(oracle, ds.export_blob())
    assert oracle.validate_blob(blob)[0] == 0
    hits, mask, rec = _same_as_oracle(oracle, blob, ds, rays, "edge lbvh")
    compare_hits_struct(hits, mask, g, "edge lbvh vs reference fixture")
    # vertex indices reported in a hit are the caller's
    i = 0
    assert mask[i] and set(hits["vertex"]["index"][i]) == set(inv0.reshape(-1, 3)[hits["triangle_index"][i]])


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 17])
def test_tiny_scenes(api, oracle, n):
    tris = synth.triangle_soup(max(n, 1), 0.5, seed=9)[:3 * n]
    ds = api.DeviceScene.build([dict(positions=tris)])
    assert ds.info()["num_triangles"] == n
    rays = synth.rays_config1(2048)
    rec = ds.trace(rays, full=False)
    if n == 0:
        assert (rec["prim"] == 0xFFFFFFFF).all()
        return
    blobs = oracle.leaf_chain_blobs(tris.reshape(-1, 3, 3))
    ohits, omask = oracle.trace_chain(blobs, rays)
    assert ((rec["prim"] != 0xFFFFFFFF) == omask).all()
    assert (rec["prim"][omask] == ohits["triangle_index"][omask]).all()
    assert np.allclose(rec["t"][omask], ohits["t"][omask], rtol=1e-5, atol=0)


def test_rtk_build_scene_entry_point(api, oracle, golden_dir):
    """The reference's own entry points: rtk_build_scene -> rtk_trace_rays / rtk_trace_ray -> rtk_free_scene."""
    tris = synth.scene_for_config(1)
    scene, keep = api.build_scene([dict(positions=tris)])
    try:
        blob = _as_blob(oracle, api.scene_bytes(scene))
        assert oracle.validate_blob(blob)[0] == 0
        g = load_golden(golden_dir, "cfg1_full.npz")
        rays = synth.rays_config1(65536)[:4096]
        hits, mask = api.trace_rays(scene, rays)
        gs = {k: g[k][:4096] for k in ("hit_mask", "hit_mesh", "hit_tri", "hit_t", "hit_u", "hit_v")}
        compare_hits_struct(hits, mask, gs, "rtk_build_scene + rtk_trace_rays")
        h = api.trace_ray(scene, rays[0])
        assert (h is not None) == bool(g["hit_mask"][0])
    finally:
        api.free_scene(scene)


def test_task_api_shape(api, oracle):
    """rtk_start_build(desc, &task) / rtk_run_task / rtk_get_build_size / rtk_finish_build_to (rtk.h:119-123)."""
    from rtk_amd.types import MeshSet
    L = api.lib()
    ms = MeshSet([dict(positions=synth.scene_for_config(1))])
    first = Task()
    b = L.rtk_start_build(C.byref(ms.desc), C.byref(first))
    assert b and first.build == b and first.fn
    assert L.rtk_get_build_size(b) == 0                      # nothing has run yet
    queue = (Task * 128)()
    assert L.rtk_run_task(C.byref(first), queue, 128) == 0   # one task does the whole device build
    size = L.rtk_get_build_size(b)
    assert size > 10000 * 48
    small = np.zeros(1024, np.uint8)
    assert not L.rtk_finish_build_to(b, small.ctypes.data, small.size)   # too small: NULL, build stays alive
    buf = oracle._aligned_bytes(size)
    s = L.rtk_finish_build_to(b, buf.ctypes.data, size)
    assert s == buf.ctypes.data
    assert oracle.validate_blob(oracle.Blob(buf))[0] == 0
    rays = synth.rays_config1(1024)
    hits, mask = api.trace_rays(s, rays)
    ohits, omask = oracle.trace(oracle.Blob(buf), rays)
    assert (mask == omask).all() and (hits["triangle_index"][mask] == ohits["triangle_index"][omask]).all()
    L.rtk_amd_forget_scene(s)


@pytest.mark.slow
def test_config2_device_build_vs_golden_and_oracle(api, oracle, golden_dir):
    tris = synth.scene_for_config(2)
    ds = api.DeviceScene.build([dict(positions=tris)])
    assert ds.info()["num_triangles"] == 1_000_000
    blob = _as_blob(oracle, ds.export_blob())
    rc, counts = oracle.validate_blob(blob)
    assert rc == 0 and counts["tris"] == 1_000_000
    g2 = load_golden(golden_dir, "cfg2_sample.npz")
    r2 = np.concatenate([synth.rays_pinhole(first=int(i), count=1) for i in g2["ray_index"]])
    hits, mask, _ = ds.trace(r2)
    compare_hits_struct(hits, mask, g2, "lbvh cfg2 sample")
    g3 = load_golden(golden_dir, "cfg3_sample.npz")
    hits, mask, _ = ds.trace(synth.rays_incoherent(4096))
    compare_hits_struct(hits, mask, g3, "lbvh cfg3 sample")
    sel = np.arange(0, 4096 * 4096, 61)
    rays = np.concatenate([synth.rays_pinhole(first=int(a), count=1) for a in sel[:4096]] +
                          [synth.rays_incoherent(1 << 18)])
    _same_as_oracle(oracle, blob, ds, rays, "lbvh 1M vs oracle on exported blob")


@pytest.mark.slow
def test_config5_device_build_any_hit_vs_golden(api, golden_dir):
    """10M-triangle scene built on the GPU; the 1024 golden shadow rays (made by the real reference's
    leaf chain over all 10M triangles): occluded flag == reference hit boolean, closest hit == reference."""
    tris = synth.scene_for_config(5)
    g = load_golden(golden_dir, "cfg5_sample.npz")
    from tests.util import sha
    assert sha(tris) == str(g["scene_sha256"])
    ds = api.DeviceScene.build([dict(positions=tris)])
    info = ds.info()
    assert info["num_triangles"] == 10_000_000 and info["build_ms"] > 0
    rays = synth.rays_shadow(1024)
    assert sha(rays) == str(g["rays_sha256"])
    occ = ds.trace_any(rays)
    assert (occ == g["hit_mask"].astype(bool)).all()
    hits, mask, _ = ds.trace(rays)
    compare_hits_struct(hits, mask, g, "lbvh cfg5 sample")
    # a larger batch: any-hit flag equals the closest-hit boolean
    big = synth.rays_shadow(1 << 20)
    rec = ds.trace(big, full=False)
    assert (ds.trace_any(big) == (rec["prim"] != 0xFFFFFFFF)).all()


def test_device_resident_mesh_is_read_in_place(api, oracle):
    """Positions (and indices) that already live in HBM: same scene, same hits as host buffers."""
    import torch
    tris = synth.scene_for_config(1)
    ds_host = api.DeviceScene.build([dict(positions=tris)])
    ds_dev = api.DeviceScene.build([dict(positions=torch.from_numpy(tris).cuda())])
    rays = synth.rays_config1(8192)
    assert ds_host.trace(rays, full=False).tobytes() == ds_dev.trace(rays, full=False).tobytes()
    idx = np.arange(30000, dtype=np.int32).reshape(-1, 3)[::-1].copy()
    ds_idx = api.DeviceScene.build([dict(positions=torch.from_numpy(tris).cuda(), indices=torch.from_numpy(idx).cuda())])
    rec = ds_idx.trace(rays, full=False)
    base = ds_host.trace(rays, full=False)
    hit = base["prim"] != 0xFFFFFFFF
    assert ((rec["prim"] != 0xFFFFFFFF) == hit).all()
    assert (rec["prim"][hit] == 9999 - base["prim"][hit]).all() and (rec["t"][hit] == base["t"][hit]).all()


def _morton_keys(tris):
    """Morton keys exactly as rtk_build.hip's k_bounds/k_morton compute them (float32 ops): 63 bits, of which
    the builder keeps the top 48 for scenes below 2^24 triangles."""
    t = tris.reshape(-1, 3, 3)
    c2 = (t.min(axis=1) + t.max(axis=1)).astype(np.float32)
    lo, hi = c2.min(axis=0), c2.max(axis=0)
    ext = (hi - lo).astype(np.float32)
    x = np.where(ext > 0, (c2 - lo) / np.where(ext > 0, ext, 1), 0).astype(np.float32)
    x = np.clip(x, 0, 1)
    q = np.minimum((x * np.float32(2097152.0)).astype(np.uint64), 2097151)

    def spread(v):
        v = v & np.uint64(0x1fffff)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v
    k = (spread(q[:, 0]) << np.uint64(2)) | (spread(q[:, 1]) << np.uint64(1)) | spread(q[:, 2])
    if len(t) >= (1 << 24):
        return k
    # the packed sort word keeps ceil(log2 n) + 8 bits of the 63-bit code, in whole 8-bit radix passes, 24 ... 40 of them
    # (rtk_build.hip: key width from n)
    lg = int(np.ceil(np.log2(max(2, len(t)))))
    bits = min(40, max(24, ((lg + 8 + 7) // 8) * 8))
    return k >> np.uint64(63 - bits)


@pytest.mark.parametrize("n", [1000, 4096, 4097, 70001, 1_000_000])
def test_radix_sort_leaves_triangles_in_morton_order(api, n):
    """The LDS-staged LSD radix sort: device triangle order is a permutation, sorted by Morton key,
    and stable (equal keys keep input order)."""
    tris = synth.triangle_soup(n, 0.02, seed=11)
    ds = api.DeviceScene.build([dict(positions=tris)])
    order = ds.primitive_order()
    assert len(order) == n and (np.sort(order) == np.arange(n)).all()
    keys = _morton_keys(tris)[order]
    assert (keys[1:] >= keys[:-1]).all()
    same = keys[1:] == keys[:-1]
    assert (order[1:][same] > order[:-1][same]).all()


@pytest.mark.parametrize("n,spread", [(2, 0.5), (3, 0.5), (1023, 0.1), (1024, 0.1), (1025, 0.1), (10_000, 0.05), (1_000_000, 0.02)])
def test_device_bvh_is_structurally_valid_and_reproducible(api, n, spread):
    """The device validator (independent of any traversal): every child box is the exact union of what is below
    it, every triangle sits in exactly one leaf, every node is referenced once; two builds of the same input are
    byte-identical (content hash over nodes and triangle records; at small sizes also the exported blobs).
    Sizes straddle the refit tile (1024 sorted triangles per workgroup) so that both refit passes are exercised."""
    tris = synth.triangle_soup(n, spread, seed=17)
    ds = api.DeviceScene.build([dict(positions=tris)])
    ok, c = ds.validate()
    assert ok, c
    assert c["triangles_checked"] == n and c["nodes_checked"] == ds.info()["num_nodes"]
    assert c["loose_boxes"] == 0 and c["first_bad_index"] == 2 ** 64 - 1
    ds2 = api.DeviceScene.build([dict(positions=tris)])
    ok2, c2 = ds2.validate()
    assert ok2 and c2["content_hash"] == c["content_hash"] and c2["nodes_checked"] == c["nodes_checked"]
    if n <= 10_000:
        assert ds.export_blob().tobytes() == ds2.export_blob().tobytes()
    # a different input gives a different hash (the hash is not vacuous)
    other = tris.copy(); other[0, 0] += np.float32(1e-3)
    assert api.DeviceScene.build([dict(positions=other)]).validate()[1]["content_hash"] != c["content_hash"]


def test_validator_sees_a_wrong_box_and_a_lost_triangle(api, oracle):
    """Negative control: a blob whose root child box was shrunk, and one whose leaf lost a triangle, fail."""
    tris = synth.triangle_soup(300, 0.2, seed=4)
    blob = oracle.build_scene([dict(positions=tris)])
    ok, c = api.DeviceScene.upload(blob).validate()
    assert ok and c["triangles_checked"] == 300, c
    bad = blob.data.copy()
    bx = bad[128:128 + 32].view(np.float32)         # bounds_x[min][4], bounds_x[max][4] of the root
    bx[4] = bx[0] + (bx[4] - bx[0]) * 0.5           # halve the x extent of child 0
    ok, c = api.DeviceScene.upload(bad).validate()
    assert not ok and c["box_violations"] >= 1 and c["first_bad_index"] == 0


@pytest.mark.slow
def test_config5_device_bvh_is_structurally_valid_and_reproducible(api):
    tris = synth.scene_for_config(5)
    ds = api.DeviceScene.build([dict(positions=tris)])
    ok, c = ds.validate()
    assert ok and c["triangles_checked"] == 10_000_000 and c["loose_boxes"] == 0, c
    h = c["content_hash"]
    ds.free()
    ds2 = api.DeviceScene.build([dict(positions=tris)])
    assert ds2.validate()[1]["content_hash"] == h


@pytest.mark.slow
@pytest.mark.parametrize("n", [(1 << 24) - 1, 1 << 24, (1 << 24) + 300_001])
def test_around_2_pow_24_triangles(api, oracle, n):
    """From 2^24 triangles on the sort works on (key, index) pairs in eight passes with the three-launch histogram scan;
    below, on one packed word (40-bit code over a 24-bit index) in five passes: the last scene of the one kind, the first
    of the other, and 17M. Structure validates, two builds agree, and a sample of rays hits real triangles at the reported t."""
    tris = synth.triangle_soup(n, 0.008, seed=21)
    ds = api.DeviceScene.build([dict(positions=tris)])
    ok, c = ds.validate()
    assert ok and c["triangles_checked"] == n and c["loose_boxes"] == 0, c
    h = c["content_hash"]
    rays = synth.rays_config1(4096)
    rec = ds.trace(rays, full=False)
    rec_img = ds.trace(rays, opts=api.make_opts(image=(64, 64)), full=False)
    assert rec_img.tobytes() == rec.tobytes()
    gm = rec["prim"] != 0xFFFFFFFF
    assert gm.sum() > 3000 and (rec["prim"][gm] < n).all()
    # every reported hit is a real intersection of that triangle at that t (the oracle's leaf chain on the one triangle)
    t3 = tris.reshape(-1, 3, 3)
    for i in np.nonzero(gm)[0][:200]:
        one = oracle.leaf_chain_blobs(t3[int(rec["prim"][i]):int(rec["prim"][i]) + 1])
        h1, m1 = oracle.trace_chain(one, rays[i:i + 1])
        assert m1[0] and abs(float(h1["t"][0]) - float(rec["t"][i])) <= 1e-5 * abs(float(h1["t"][0])), int(i)
    ds.free()
    ds2 = api.DeviceScene.build([dict(positions=tris)])
    assert ds2.validate()[1]["content_hash"] == h


def test_default_index_type_means_u32(api):
    """rtk_mesh.index.type == RTK_TYPE_DEFAULT with an index buffer: 32-bit indices (rtk.c:1049-1059)."""
    from rtk_amd.types import Mesh, SceneDesc, RTK_TYPE_DEFAULT, RTK_TYPE_U32
    tris = synth.triangle_soup(700, 0.1, seed=8)
    verts, inv = np.unique(tris, axis=0, return_inverse=True)
    idx = np.ascontiguousarray(inv.reshape(-1, 3).astype(np.uint32))
    verts = np.ascontiguousarray(verts.astype(np.float32))
    rays = synth.rays_config1(4096)
    recs = []
    for ty in (RTK_TYPE_DEFAULT, RTK_TYPE_U32):
        m = Mesh()
        m.num_triangles = len(idx)
        m.position.data = verts.ctypes.data
        m.position.type = RTK_TYPE_DEFAULT          # float32, tightly packed
        m.index.data = idx.ctypes.data
        m.index.type = ty
        arr = (Mesh * 1)(m)
        desc = SceneDesc()
        desc.meshes = C.cast(arr, C.POINTER(Mesh))
        desc.num_meshes = 1
        h = api.lib().rtk_dev_scene_build(C.byref(desc))
        assert h, api.last_error()
        ds = api.DeviceScene(h, keepalive=(arr, verts, idx))
        assert ds.validate()[0]
        recs.append(ds.trace(rays, full=False).tobytes())
    assert recs[0] == recs[1]
    ref = api.DeviceScene.build([dict(positions=tris)])
    a = np.frombuffer(recs[0], dtype=ref.trace(rays, full=False).dtype)
    b = ref.trace(rays, full=False)
    assert (a["prim"] == b["prim"]).all() and (a["t"] == b["t"]).all()


def test_cpu_task_builder_blob_on_the_device(api, oracle, golden_dir):
    """The CPU task-graph builder (rtk_amd_set_builder) + the GPU tracer: rtk_build_scene -> rtk_trace_rays, the device
    validator on the uploaded blob, hits equal to the reference fixture and to the oracle on the same blob."""
    L = api.lib()
    assert L.rtk_amd_set_builder(1) == 0
    try:
        tris = synth.scene_for_config(1)
        scene, keep = api.build_scene([dict(positions=tris)])
    finally:
        L.rtk_amd_set_builder(0)
    try:
        blob = _as_blob(oracle, api.scene_bytes(scene))
        assert oracle.validate_blob(blob)[0] == 0
        ds = api.DeviceScene.upload(blob)
        ok, c = ds.validate()
        assert ok and c["triangles_checked"] == 10000 and c["loose_boxes"] == 0, c
        rays = synth.rays_config1(65536)
        hits, mask, _ = _same_as_oracle(oracle, blob, ds, rays, "cpu-built blob on the device")
        compare_hits_struct(hits, mask, load_golden(golden_dir, "cfg1_full.npz"), "cpu-built blob vs reference fixture")
        h2, m2 = api.trace_rays(scene, rays[:4096])
        assert (m2 == mask[:4096]).all() and (h2["triangle_index"][m2] == hits["triangle_index"][:4096][m2]).all()
    finally:
        api.free_scene(scene)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_mixed_meshes_device_build(api, oracle, seed):
    """The same random multi-mesh scenes (implicit / u16 / u32 indices, f32 / f64, strides, callbacks) through the DEVICE
    builder: structurally valid, and the GPU's hits carry the caller's (mesh, triangle) identity and vertex indices --
    equal to the reference-style leaf chain over the same triangles."""
    from tests.util import random_mixed_scene
    desc, keep, tris, mesh_index, tri_index, vidx = random_mixed_scene(seed)
    h = api.lib().rtk_dev_scene_build(C.byref(desc))
    assert h, api.last_error()
    ds = api.DeviceScene(h, keepalive=keep)
    ok, c = ds.validate()
    assert ok and c["triangles_checked"] == len(tris) and c["loose_boxes"] == 0, c
    rays = synth.rays_config1(8192, seed=seed + 20)
    hits, mask, rec = ds.trace(rays)
    ohits, omask = oracle.trace_chain(oracle.leaf_chain_blobs(tris, mesh_index, tri_index, vidx), rays)
    assert (mask == omask).all() and mask.sum() > 100
    for k in ("mesh_index", "triangle_index"):
        assert (hits[k][mask] == ohits[k][omask]).all()
    assert np.allclose(hits["t"][mask], ohits["t"][omask], rtol=1e-5, atol=0)
    assert (np.sort(hits["vertex"]["index"][mask], axis=1) == np.sort(ohits["vertex"]["index"][omask], axis=1)).all()
    base = ds.mesh_base()
    assert (rec["prim"][mask] == base[hits["mesh_index"][mask].astype(np.int64)] + hits["triangle_index"][mask]).all()


def _adversarial_scenes():
    rng = np.random.RandomState(5)
    n = 40000
    base = synth.triangle_soup(n, 0.05, seed=3).reshape(n, 3, 3)
    scenes = {}
    tiny = base * np.float32(1e-4)
    tiny[0] = np.array([[900, 900, 900], [1000, 900, 900], [900, 1000, 950]], np.float32)      # one far, large triangle
    scenes["tiny_cluster_plus_far_triangle"] = tiny
    scenes["all_identical"] = np.repeat(base[:1], 5000, axis=0)                                  # every Morton key equal
    line = base.copy(); line[:, :, 1] *= np.float32(1e-6); line[:, :, 2] *= np.float32(1e-6)      # a 1-D distribution along x
    scenes["line_along_x"] = line
    scenes["huge_coordinates"] = base * np.float32(1e6) + np.float32(3e7)
    two = base.copy(); two[n // 2:] += np.float32(50.0)                                          # two clusters far apart
    scenes["two_distant_clusters"] = two
    flat = base.copy(); flat[:, :, 2] = np.float32(0.5)                                           # all triangles in one plane: flat boxes
    scenes["coplanar"] = flat
    g = 141                                                                                      # a floor: 2*141^2 non-overlapping triangles in z = 0.25
    xs = (np.arange(g + 1, dtype=np.float32) / np.float32(g))
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    P = np.stack([X, Y, np.full_like(X, 0.25)], axis=-1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    scenes["flat_floor_grid"] = np.concatenate([np.stack([a, b, c], axis=2).reshape(-1, 3, 3), np.stack([a, c, d], axis=2).reshape(-1, 3, 3)])
    # degenerate triangles in a row on the x axis: every box is flat on two axes, so every surface area is exactly zero
    xs = np.arange(5000, dtype=np.float32) * np.float32(1e-3)
    row = np.zeros((5000, 3, 3), np.float32)
    row[:, 0, 0] = xs; row[:, 1, 0] = xs + np.float32(3e-4); row[:, 2, 0] = xs + np.float32(6e-4)
    scenes["zero_area_row"] = row
    # a geometric sequence on every axis: Morton splits peel small groups off, extents underflow towards the small end
    k = np.arange(3000)
    cc = np.stack([0.5 ** (k % 40 + 1) * (1 + (k // 40) * 1e-3), 0.5 ** ((k * 7) % 40 + 1), 0.5 ** ((k * 13) % 40 + 1)], 1).astype(np.float32)
    scenes["geometric_chain"] = np.stack([cc, cc + np.float32([1e-4, 0, 0]) * cc[:, :1], cc + np.float32([0, 1e-4, 0]) * cc[:, :1]], 1).astype(np.float32)
    return scenes


@pytest.mark.parametrize("name", ["tiny_cluster_plus_far_triangle", "all_identical", "line_along_x", "huge_coordinates",
                                  "two_distant_clusters", "coplanar", "flat_floor_grid", "zero_area_row", "geometric_chain"])
def test_adversarial_distributions(api, oracle, name):
    """Distributions that stress the Morton build (equal keys, truncated keys, degenerate extents, deep unbalanced
    trees): the device BVH stays structurally valid and deterministic. Where boxes are resolvable in float, everything
    coincides: device == rtk.c on the same BVH bit for bit (both node formats, both kernels) == brute force. Where they
    are not (a 1e-4 cluster seen from 1e3 away, a line 1e-7 thin, overlapping triangles in a plane of zero thickness),
    entry distances of boxes and hit distances of triangles are the same number up to rounding, so rtk.c's own culling
    (rtk.c:432, 458-470) depends on the visit order among equal keys and ANY tight BVH can drop hits a brute-force pass
    finds; there only what holds regardless is asserted (no traversal error, ids and t in range, barycentric u/v)."""
    tris = np.ascontiguousarray(_adversarial_scenes()[name].reshape(-1, 3))
    ds = api.DeviceScene.build([dict(positions=tris)])
    ok, c = ds.validate()
    assert ok and c["loose_boxes"] == 0, (name, c)
    assert api.DeviceScene.build([dict(positions=tris)]).validate()[1]["content_hash"] == c["content_hash"]
    t3 = tris.reshape(-1, 3, 3)
    lo, hi = t3.min(axis=(0, 1)).astype(np.float64), t3.max(axis=(0, 1)).astype(np.float64)
    ext = np.maximum(hi - lo, 0.25 * (hi - lo).max())          # rays come from a cube around the scene, never from inside a flat one
    n = 4096
    u = synth.u01(9, 0, n * 6).reshape(n, 6).astype(np.float64)
    rays = np.zeros(n, RAY_DTYPE)
    org = lo - 0.5 * ext + 2.0 * ext * u[:, 0:3]
    tgt = lo + ext * u[:, 3:6]
    # half of the rays aim at triangle centroids so that thin / tiny scenes are hit at all
    cent = t3.mean(axis=1).astype(np.float64)
    tgt[::2] = cent[(np.arange(n // 2) * 7919) % len(cent)]
    rays["origin"] = org.astype(np.float32)
    rays["direction"] = (tgt - org).astype(np.float32)
    rays["max_t"] = np.float32(4.0)
    blob = _as_blob(oracle, ds.export_blob())
    assert oracle.validate_blob(blob)[0] == 0
    resolvable = name in ("huge_coordinates", "two_distant_clusters", "all_identical", "flat_floor_grid")
    exact = ds.trace(rays, opts=api.make_opts(exact_nodes=True), full=False)
    rec = ds.trace(rays, full=False)
    em, qm = exact["prim"] != 0xFFFFFFFF, rec["prim"] != 0xFFFFFFFF
    assert em.sum() > 50 or name in ("zero_area_row", "geometric_chain"), (name, int(em.sum()))   # (nothing there to hit: lines and specks)
    ohits, omask = oracle.trace_chain(oracle.leaf_chain_blobs(t3), rays)          # brute force: rtk.c's triangle test on every triangle
    t_chain = np.where(omask, ohits["t"], np.float32(np.inf))
    if resolvable:
        # (a) rtk.c on the same BVH, bit for bit: exact and compressed nodes, per-lane and packet kernel
        oh, om = oracle.trace(blob, rays)
        assert (em == om).all() and (exact["prim"][em] == oh["triangle_index"][om]).all(), name
        assert (exact["t"][em] == oh["t"][om]).all() and (exact["u"][em] == oh["u"][om]).all() and (exact["v"][em] == oh["v"][om]).all(), name
        assert rec.tobytes() == exact.tobytes()
        if ds.info()["stack_entries"] <= 64:                          # else the per-lane kernel takes image-shaped batches too
            assert ds.trace(rays, opts=api.make_opts(image=(64, 64)), full=False).tobytes() == exact.tobytes()
    # (b) unresolvable scenes: hit/miss and the winner are rounding noise in rtk.c itself (and differ with the leaf grouping
    #     through the double-precision group rule, rtk.c:302-336), so only what must hold regardless is asserted: no
    #     traversal error, ids in range, t inside the ray's interval, u/v barycentric up to rounding
    assert api.lib().rtk_dev_trace_status(ds.handle, None) == 0
    for r_, m_ in ((exact, em), (rec, qm)):
        assert (r_["prim"][m_] < len(t3)).all(), name
        assert (r_["t"][m_] >= rays["min_t"][m_]).all() and (r_["t"][m_] <= rays["max_t"][m_]).all(), name
        assert (r_["u"][m_] >= -1e-5).all() and (r_["v"][m_] >= -1e-5).all() and (r_["u"][m_] + r_["v"][m_] <= 1 + 1e-5).all(), name
    if name in ("huge_coordinates", "two_distant_clusters", "flat_floor_grid"):
        assert (omask == qm).all() and (ohits["triangle_index"][omask] == rec["prim"][qm]).all()
        assert np.allclose(rec["t"][qm], ohits["t"][omask], rtol=1e-5, atol=0)
    elif name == "all_identical":
        assert (rec["prim"][qm] == 0).all()                                        # 5000 copies: the lowest id wins every tie


@pytest.mark.gpu
def test_large_scene_sort_paths_on_a_small_scene():
    """Scenes of 2^24 triangles and more sort (key, index) pairs in eight passes and scan the digit histogram with the
    three-launch scan; smaller ones never take those paths. RTK_AMD_SORT_PACKED=0 / RTK_AMD_SORT_FUSED_SCAN=0 force them
    (read once per process, hence the child process): the BVH must validate and trace like the oracle on its blob."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from rtk_amd import api, synth
from oracle import pyoracle
tris = synth.triangle_soup(50000, 0.05, seed=11)
ds = api.DeviceScene.build([dict(positions=tris)])
ok, c = ds.validate()
assert ok and c["loose_boxes"] == 0, c
rays = synth.rays_config1(4096)
rec = ds.trace(rays, full=False)
eh, em = pyoracle.trace(pyoracle.Blob(ds.export_blob()), rays)
gm = rec["prim"] != 0xFFFFFFFF
assert (gm == em).all() and (rec["prim"][gm] == eh["triangle_index"][em]).all() and (rec["t"][gm] == eh["t"][em]).all()
print("OK", int(gm.sum()), c["content_hash"])
''' % ROOT
    outs = []
    # (third run: a node-count estimate that is far too small, so that the collapse is repeated into the workspace)
    for env_extra in ({}, {"RTK_AMD_SORT_PACKED": "0", "RTK_AMD_SORT_FUSED_SCAN": "0"}, {"RTK_AMD_NODE_ESTIMATE_DIV": "16"}):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
        outs.append(r.stdout.split())
    assert int(outs[0][1]) > 500 and outs[0][1] == outs[1][1] == outs[2][1]        # same hits either way
    assert outs[0][2] == outs[2][2]                                                # and the very same tree after the repeat


@pytest.mark.parametrize("n", [1025, 10_000, 300_000])
def test_tile_local_collapse_at_small_sizes(api, oracle, n, monkeypatch):
    """The tile-local collapse (k_count_tile / k_collapse_tile: subtrees inside a 1024-triangle refit tile are turned into
    4-wide nodes by the tile's own workgroup, numbered in pre-order, the nodes above the tiles behind them) is the default from
    1.5M triangles on; RTK_AMD_TILE_COLLAPSE_MIN=0 forces it here. Structure validates (no cycles: children after their parents inside the tiles' run of numbers and inside the run above, exact
    boxes, every triangle once), two builds are byte-identical, the level-by-level build of the same input holds the same
    triangles, and tracing is bit-identical to the oracle on the exported blob."""
    tris = synth.triangle_soup(n, 0.05, seed=70)
    monkeypatch.setenv("RTK_AMD_TILE_COLLAPSE_MIN", "0")
    ds = api.DeviceScene.build([dict(positions=tris)])
    ok, c = ds.validate()
    assert ok and c["triangles_checked"] == n and c["loose_boxes"] == 0, c
    ds_b = api.DeviceScene.build([dict(positions=tris)])
    assert ds_b.validate()[1]["content_hash"] == c["content_hash"]
    assert ds_b.export_blob().tobytes() == ds.export_blob().tobytes()
    monkeypatch.setenv("RTK_AMD_TILE_COLLAPSE_MIN", str(1 << 40))
    ds_level = api.DeviceScene.build([dict(positions=tris)])
    ok2, c2 = ds_level.validate()
    assert ok2 and c2["content_hash"] != c["content_hash"]          # another (valid) tree: tile roots are never opened from above
    # a node estimate that is far too small: the tiles' nodes and the move of the nodes above them are done once more into an
    # exact allocation -- the very same tree
    monkeypatch.setenv("RTK_AMD_TILE_COLLAPSE_MIN", "0")
    monkeypatch.setenv("RTK_AMD_NODE_ESTIMATE_DIV", "16")
    ds_r = api.DeviceScene.build([dict(positions=tris)])
    ok3, c3 = ds_r.validate()
    assert ok3 and c3["content_hash"] == c["content_hash"], c3
    monkeypatch.delenv("RTK_AMD_NODE_ESTIMATE_DIV")
    # more nodes above the tiles than their stretch of the workspace holds (forced: two): the build goes the level-by-level way
    monkeypatch.setenv("RTK_AMD_TOP_CAP", "2")
    # (a valid tree of its own: the refit of a tile-mode build keeps subtrees that cross a tile border from becoming one leaf)
    ds_f = api.DeviceScene.build([dict(positions=tris)])
    ok4, c4 = ds_f.validate()
    assert ok4 and c4["triangles_checked"] == n and c4["loose_boxes"] == 0, c4
    assert api.DeviceScene.build([dict(positions=tris)]).validate()[1]["content_hash"] == c4["content_hash"]
    _same_as_oracle(oracle, _as_blob(oracle, ds_f.export_blob()), ds_f, synth.rays_config1(4096), "top of the tree level by level vs oracle on exported blob")
    monkeypatch.delenv("RTK_AMD_TOP_CAP")
    rays = synth.rays_config1(16384)
    blob = _as_blob(oracle, ds.export_blob())
    _same_as_oracle(oracle, blob, ds, rays, "tile-local collapse vs oracle on exported blob")
    rec_img = ds.trace(rays, opts=api.make_opts(image=(128, 128)), full=False)
    assert rec_img.tobytes() == ds.trace(rays, full=False).tobytes()
    # the two trees agree on every hit but near-ties (ids exact, t to 1e-5 through compare_hits' tolerance on another BVH)
    h1, m1, _ = ds.trace(rays)
    h2, m2, _ = ds_level.trace(rays)
    compare_hits(m1, h1["mesh_index"], h1["triangle_index"], h1["t"], h1["u"], h1["v"],
                 m2, h2["mesh_index"], h2["triangle_index"], h2["t"], h2["u"], h2["v"], "tile collapse vs level collapse")


def test_records_made_inside_the_refit_are_the_separate_pass_records(api, monkeypatch):
    """k_refit_tile makes the triangle records of its tile itself (the gather of k_emit_tris fused in; each thread keeps its own record for
    the leaf box). RTK_AMD_FUSED_EMIT=0 runs the separate pass: the same scene, byte for byte -- for a float mesh read in place, for
    indexed / double / multi-mesh scenes that go through staged records, below and above the tile collapse's threshold."""
    scenes = [[dict(positions=synth.triangle_soup(30_001, 0.05, seed=5))],
              [dict(positions=synth.triangle_soup(2_100_000, 0.02, seed=6))]]
    v = synth.triangle_soup(5_000, 0.05, seed=7).reshape(-1, 3)
    idx = np.arange(len(v), dtype=np.uint32).reshape(-1, 3)[::-1].copy()
    scenes.append([dict(positions=v, indices=idx), dict(positions=synth.triangle_soup(3_000, 0.05, seed=8).astype(np.float64))])
    for meshes in scenes:
        monkeypatch.delenv("RTK_AMD_FUSED_EMIT", raising=False)
        fused = api.DeviceScene.build(meshes)
        ok, c = fused.validate()
        assert ok, c
        monkeypatch.setenv("RTK_AMD_FUSED_EMIT", "0")
        apart = api.DeviceScene.build(meshes)
        assert apart.validate()[1]["content_hash"] == c["content_hash"]
        assert apart.export_blob().tobytes() == fused.export_blob().tobytes()
    monkeypatch.delenv("RTK_AMD_FUSED_EMIT", raising=False)


def test_clustered_scene_is_rebuilt_with_wide_keys(api, oracle, monkeypatch):
    """The Morton key width follows the number of triangles (32 bits at 200k), which is too narrow when they sit in 1 % of the
    scene box: dozens share a cell and are ordered by their numbers. k_refit_tile counts equal-code neighbours and the build is
    repeated at 40 bits. Same hits either way; fewer node and triangle visits per ray with the wide keys."""
    n = 200_000
    dense = (synth.triangle_soup(n, 0.03, seed=31).reshape(-1, 3) * np.float32(0.01) + np.float32(0.495)).astype(np.float32)
    far = np.array([[-1, -1, -1], [-1, -1, -0.99], [-1, -0.99, -1], [2, 2, 2], [2, 2, 2.01], [2, 2.01, 2]], np.float32)
    tris = np.ascontiguousarray(np.concatenate([dense, far]))
    rays = synth.rays_config1(20000, seed=7)
    rays["origin"] = rays["origin"] * np.float32(0.01) + np.float32(0.495)
    rays["origin"][:, 2] = np.float32(0.48)
    rays["direction"] *= np.float32([0.01, 0.01, 1.0])

    def build_and_count():
        ds = api.DeviceScene.build([dict(positions=tris)])
        ok, c = ds.validate()
        assert ok and c["triangles_checked"] == n + 2, c
        rec, ctr = ds.trace_counted(rays)
        ds.free()
        return rec, ctr

    monkeypatch.setenv("RTK_AMD_KEY_REBUILD", "0")
    rec_narrow, ctr_narrow = build_and_count()
    monkeypatch.delenv("RTK_AMD_KEY_REBUILD")
    rec_wide, ctr_wide = build_and_count()
    assert (rec_narrow["prim"] != 0xFFFFFFFF).mean() > 0.5
    assert rec_narrow.tobytes() == rec_wide.tobytes()
    # the wide keys give the tree its resolution back: clearly fewer triangle tests and node visits
    assert ctr_wide["triangles"] < 0.8 * ctr_narrow["triangles"], (ctr_wide, ctr_narrow)
    assert ctr_wide["nodes"] < ctr_narrow["nodes"], (ctr_wide, ctr_narrow)
