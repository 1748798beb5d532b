"""GPU parity tests: the HIP traversal path (through the C-ABI of librtk_amd.so) against
the golden fixtures made by the real reference and against the CPU oracle on the same
inputs. Bar: hit/miss + ids bit-exact; |dt| <= 1e-5 |t|; |du|,|dv| <= 1e-5 max(1,|.|).
"""
import numpy as np
import pytest

from rtk_amd import synth
from rtk_amd.types import RAY_DTYPE
from tests.util import compare_hits, compare_hits_struct, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene1(oracle, api):
    tris = synth.scene_for_config(1)
    blob = oracle.build_scene([dict(positions=tris)])
    return blob, api.DeviceScene.upload(blob)


@pytest.fixture(scope="module")
def scene2(oracle, api):
    tris = synth.scene_for_config(2)
    blob = oracle.build_scene([dict(positions=tris)])
    return blob, api.DeviceScene.upload(blob)


def test_native_library_is_loaded(api):
    import ctypes
    assert isinstance(api.lib(), ctypes.CDLL)
    assert api.lib().rtk_amd_device_count() >= 1


def test_config1_vs_golden_and_oracle(api, oracle, scene1, golden_dir):
    blob, ds = scene1
    info = ds.info()
    assert info["num_triangles"] == 10000
    rays = synth.rays_config1(65536)
    hits, mask, rec = ds.trace(rays)
    g = load_golden(golden_dir, "cfg1_full.npz")
    st = compare_hits_struct(hits, mask, g, "gpu/cfg1 vs reference fixture")
    assert st["hits"] == int(g["hit_mask"].sum())
    # same blob through the CPU oracle
    ohits, omask = oracle.trace(blob, rays)
    st = compare_hits(mask, hits["mesh_index"], hits["triangle_index"], hits["t"], hits["u"], hits["v"],
                      omask, ohits["mesh_index"], ohits["triangle_index"], ohits["t"], ohits["u"], ohits["v"], "gpu vs oracle")
    # same leaves, same groups of four -> the HIP kernel reproduces rtk.c's arithmetic bit for bit
    assert st["bit_exact"] == 1.0
    # expansion returns the caller's vertices
    assert (hits["vertex"]["position"][mask] == ohits["vertex"]["position"][omask]).all()
    assert (hits["vertex"]["index"][mask] == ohits["vertex"]["index"][omask]).all()
    # compact records: single mesh -> prim == triangle_index; misses carry max_t
    assert (rec["prim"][mask] == hits["triangle_index"][mask]).all()
    assert (rec["prim"][~mask] == 0xFFFFFFFF).all() and (rec["t"][~mask] == rays["max_t"][~mask]).all()


def test_edge_cases_vs_golden(api, oracle, golden_dir):
    g = load_golden(golden_dir, "edge_cases.npz")
    rays = np.ascontiguousarray(g["rays"]).view(RAY_DTYPE).reshape(-1)
    meshes = [dict(positions=g["tris"][g["mesh"] == m].reshape(-1, 3)) for m in np.unique(g["mesh"])]
    blob = oracle.build_scene(meshes)
    ds = api.DeviceScene.upload(blob)
    hits, mask, rec = ds.trace(rays)
    compare_hits_struct(hits, mask, g, "gpu/edge")
    assert list(ds.mesh_base()) == [0, 8, 10]
    # single-leaf blob as the reference saw it
    chain = oracle.leaf_chain_blobs(g["tris"], g["mesh"], g["tri_index"])
    ds2 = api.DeviceScene.upload(chain[0])
    hits2, mask2, _ = ds2.trace(rays)
    compare_hits_struct(hits2, mask2, g, "gpu/edge single leaf")


@pytest.mark.parametrize("mode", ["static", "refill8", "tiled", "tiled_per_lane", "node_exit16", "node_exit64_refill8", "node_exit64_static",
                                  "sort_rays", "sort_rays_static", "exact_nodes", "exact_nodes_static_exit1"])
def test_launch_modes_agree(api, scene1, mode):
    _, ds = scene1
    rays = synth.rays_config1(65536)
    base = ds.trace(rays, full=False)
    if mode == "static":
        opts = api.make_opts(static=True)
    elif mode == "refill8":
        opts = api.make_opts(refill_min=8, blocks_per_cu=1)
    elif mode == "tiled_per_lane":
        opts = api.make_opts(image=(256, 256), no_packet=True)
    elif mode == "sort_rays":
        opts = api.make_opts(sort_rays=True)
    elif mode == "sort_rays_static":
        opts = api.make_opts(sort_rays=True, static=True)
    elif mode == "exact_nodes":
        opts = api.make_opts(exact_nodes=True)
    elif mode == "exact_nodes_static_exit1":
        opts = api.make_opts(exact_nodes=True, static=True, node_exit=1)
    elif mode == "node_exit16":
        opts = api.make_opts(node_exit=16)
    elif mode == "node_exit64_refill8":
        opts = api.make_opts(node_exit=64, refill_min=8)
    elif mode == "node_exit64_static":
        opts = api.make_opts(node_exit=64, static=True)
    else:
        opts = api.make_opts(image=(256, 256))
    other = ds.trace(rays, opts=opts, full=False)
    assert base.tobytes() == other.tobytes()


def test_ragged_and_empty_batches(api, scene1):
    _, ds = scene1
    rays = synth.rays_config1(1000)
    full = ds.trace(rays, full=False)
    for n in (1, 63, 64, 65, 257, 999):
        part = ds.trace(rays[:n], full=False)
        assert part.tobytes() == full[:n].tobytes()
    import torch
    assert api.lib().rtk_dev_trace_rays(ds.handle, None, 0, None, None, None) == 0
    torch.cuda.synchronize()


def test_counted_build_matches_and_counts(api, oracle, scene1):
    blob, ds = scene1
    rays = synth.rays_config1(8192)
    rec = ds.trace(rays, full=False)
    rec2, ctr = ds.trace_counted(rays)
    assert rec.tobytes() == rec2.tobytes()
    assert ctr["rays"] == 8192 and ctr["hits"] == int((rec["prim"] != 0xFFFFFFFF).sum())
    assert ctr["nodes"] >= ctr["rays"] and ctr["triangles"] >= ctr["leaves"] > 0
    assert ctr["stack_spills"] == 0


def test_any_hit_equals_closest_hit_boolean(api, scene1):
    _, ds = scene1
    rays = synth.rays_config1(65536).copy()
    rays["min_t"] = 1e-3
    rays["max_t"] = 1.6
    rec = ds.trace(rays, full=False)
    occ = ds.trace_any(rays)
    assert (occ == (rec["prim"] != 0xFFFFFFFF)).all()
    assert 0 < occ.sum() < len(occ)
    assert (ds.trace_any(rays, opts=api.make_opts(exact_nodes=True)) == occ).all()


def test_single_ray_and_host_batch_entry_points(api, oracle, scene1, golden_dir):
    blob, _ = scene1
    g = load_golden(golden_dir, "cfg1_full.npz")
    rays = synth.rays_config1(65536)[:512]
    hits, mask = api.trace_rays(blob.ptr, rays)
    gs = {k: g[k][:512] for k in ("hit_mask", "hit_mesh", "hit_tri", "hit_t", "hit_u", "hit_v")}
    compare_hits_struct(hits, mask, gs, "rtk_trace_rays")
    for i in range(8):
        h = api.trace_ray(blob.ptr, rays[i])
        assert (h is not None) == bool(g["hit_mask"][i])
        if h is not None:
            assert h["triangle_index"] == g["hit_tri"][i] and abs(h["t"] - g["hit_t"][i]) <= 1e-5 * abs(g["hit_t"][i])
    api.lib().rtk_amd_forget_scene(blob.ptr)


def test_bad_blob_is_rejected(api):
    junk = np.zeros(4096, np.uint8)
    with pytest.raises(api.RtkError):
        api.DeviceScene.upload(junk)


@pytest.mark.slow
def test_config2_and_3_samples_vs_golden(api, scene2, golden_dir):
    _, ds = scene2
    assert ds.info()["num_triangles"] == 1_000_000
    g2 = load_golden(golden_dir, "cfg2_sample.npz")
    r2 = np.concatenate([synth.rays_pinhole(first=int(i), count=1) for i in g2["ray_index"]])
    hits, mask, _ = ds.trace(r2)
    compare_hits_struct(hits, mask, g2, "gpu/cfg2 sample")
    g3 = load_golden(golden_dir, "cfg3_sample.npz")
    hits, mask, _ = ds.trace(synth.rays_incoherent(4096))
    compare_hits_struct(hits, mask, g3, "gpu/cfg3 sample")


@pytest.mark.slow
def test_config2_prefix_vs_oracle_and_tiling(api, oracle, scene2):
    """2^20-ray prefix of the 4096^2 frame: GPU == oracle on every ray; tiled == untiled."""
    blob, ds = scene2
    n = 1 << 20
    rays = synth.rays_pinhole(first=0, count=n)
    hits, mask, rec = ds.trace(rays)
    ohits, omask = oracle.trace(blob, rays)
    st = compare_hits(mask, hits["mesh_index"], hits["triangle_index"], hits["t"], hits["u"], hits["v"],
                      omask, ohits["mesh_index"], ohits["triangle_index"], ohits["t"], ohits["u"], ohits["v"], "gpu vs oracle 1M")
    assert st["hits"] > n // 4 and st["bit_exact"] == 1.0
    tiled = ds.trace(rays, opts=api.make_opts(image=(4096, 256)), full=False)
    assert tiled.tobytes() == rec.tobytes()


def test_stack_spill_path(api, oracle):
    """Large overlapping triangles: every child box is hit at every level, so the per-lane stack
    outgrows its LDS entries and spills to the global buffer. Results must not change."""
    n = 6000
    u = synth.u01(31, 0, n * 9).reshape(n, 3, 3)
    tris = (u * np.float32(1.0)).reshape(-1, 3).astype(np.float32)          # each triangle spans the unit cube
    blob = oracle.build_scene([dict(positions=tris)])
    for ds in (api.DeviceScene.upload(blob), api.DeviceScene.build([dict(positions=tris)])):
        assert ds.info()["stack_entries"] > 16
        rays = synth.rays_config1(4096)
        rec, ctr = ds.trace_counted(rays)
        assert ctr["stack_spills"] > 0
        assert ds.trace(rays, full=False).tobytes() == rec.tobytes()
        # packet kernel on the same rays (as a 64x64 image) and its own spill path
        popts = api.make_opts(image=(64, 64))
        prec, pctr = ds.trace_counted(rays, popts)
        assert prec.tobytes() == rec.tobytes()
    # the upload keeps the oracle's BVH: bit-exact against the oracle itself
    ds = api.DeviceScene.upload(blob)
    hits, mask, rec = ds.trace(rays)
    oh, om = oracle.trace(blob, rays)
    assert (mask == om).all() and (hits["triangle_index"][mask] == oh["triangle_index"][om]).all()
    assert (hits["t"][mask] == oh["t"][om]).all()


def test_upload_of_maximum_size_leaves(api, oracle):
    """Blobs whose leaves hold 60-63 triangles (6-bit count, rtk.c:188) upload and trace correctly."""
    tris = synth.triangle_soup(63, 0.3, seed=41)
    blob = oracle.leaf_chain_blobs(tris.reshape(-1, 3, 3), chunk=63)[0]
    ds = api.DeviceScene.upload(blob)
    assert ds.info()["num_triangles"] == 63 and ds.info()["num_nodes"] == 1
    rays = synth.rays_config1(4096)
    hits, mask, _ = ds.trace(rays)
    oh, om = oracle.trace(blob, rays)
    assert (mask == om).all() and om.sum() > 100
    assert (hits["triangle_index"][mask] == oh["triangle_index"][om]).all()
    assert (hits["t"][mask] == oh["t"][om]).all() and (hits["u"][mask] == oh["u"][om]).all()


def test_special_rays_take_the_exact_min_max_path(api, oracle, scene1):
    """Rays with zero / negative-zero / tiny / huge direction components (0*inf = NaN in the slab test)
    force the SSE-ordered min/max path (rtk.c:464-465) in both kernels; mixed signs and dominant axes
    inside one wave force the per-lane variants of the packet kernel. Bit-exact against the oracle."""
    blob, ds = scene1
    rays = synth.rays_config1(4096).copy()
    n = len(rays)
    k = np.arange(n)
    d = rays["direction"]
    d[k % 7 == 0, 0] = 0.0
    d[k % 11 == 0, 1] = -0.0
    d[k % 13 == 0, 0] = 1e-42          # denormal: 1/d overflows to inf
    d[k % 17 == 0, 2] = 1e30
    d[k % 37 == 0, 1] = 1e-33          # 1/d = 1e33: finite, but beyond what the packet kernel's fast slab test accepts (2^100)
    d[k % 41 == 0, 0] = 3e31           # 1/d = 3e-32 < 2^-100: likewise
    d[k % 19 == 0] *= -1.0             # mixed signs in a wave
    d[k % 23 == 0] = d[k % 23 == 0][:, [2, 0, 1]]   # other dominant axes
    rays["origin"][k % 29 == 0, 0] = 0.5
    rays["origin"][k % 43 == 0, 2] = -2.0e7   # |origin| >= 2^23: an empty child slot's +1/-1 box could round shut for such a ray
    # origins exactly on vertex coordinates of the scene: bound - origin == 0 happens for real
    tris = synth.scene_for_config(1)
    rays["origin"][k % 31 == 0, 0] = tris[(k[k % 31 == 0] * 3) % len(tris), 0]
    rays["direction"] = d
    oh, om = oracle.trace(blob, rays)
    for opts in (None, api.make_opts(image=(64, 64)), api.make_opts(static=True)):
        hits, mask, _ = ds.trace(rays, opts=opts)
        assert (mask == om).all()
        assert (hits["triangle_index"][mask] == oh["triangle_index"][om]).all()
        assert (hits["t"][mask] == oh["t"][om]).all() and (hits["u"][mask] == oh["u"][om]).all() and (hits["v"][mask] == oh["v"][om]).all()
    assert om.sum() > 500


def test_exotic_rays_bit_exact_on_the_same_bvh(api, oracle, scene1):
    """synth.rays_exotic (zeros, negative zeros, denormals, huge and tied direction components, far / on-vertex origins,
    empty, reversed and NaN-min intervals; the oracle is pinned against the real rtk.c on exactly these rays in
    tests/test_oracle_golden.py): every kernel returns what the oracle returns on the same BVH, bit for bit -- uploaded
    SAH blob and device-built LBVH, per-lane and packet kernel, exact and compressed nodes, closest-hit and any-hit."""
    blob, ds_up = scene1
    tris = synth.scene_for_config(1)
    rays = synth.rays_exotic(2048, tris=tris)
    ds_dev = api.DeviceScene.build([dict(positions=tris)])
    for ds, b in ((ds_up, blob), (ds_dev, oracle.Blob(ds_dev.export_blob()))):
        oh, om = oracle.trace(b, rays)
        assert om.sum() > 300
        for opts in (None, api.make_opts(image=(32, 64)), api.make_opts(exact_nodes=True), api.make_opts(static=True)):
            rec = ds.trace(rays, opts=opts, full=False)
            gm = rec["prim"] != 0xFFFFFFFF
            assert (gm == om).all()
            hits, mask, _ = ds.trace(rays, opts=opts)
            assert (hits["triangle_index"][mask] == oh["triangle_index"][om]).all()
            for f in ("t", "u", "v"):
                assert (hits[f][mask].view(np.uint32) == oh[f][om].view(np.uint32)).all(), f
        assert (ds.trace_any(rays) == om).all()
    # outside the reference's domain (it runs off its stack, rtk.c:477): NaN and infinite max_t. Here: no hit, no hang
    odd = rays[:256].copy()
    odd["max_t"][0::2] = np.float32("nan")
    odd["max_t"][1::2] = np.float32("inf")
    for opts in (None, api.make_opts(image=(16, 16))):
        rec = ds_dev.trace(odd, opts=opts, full=False)
        assert api.lib().rtk_dev_trace_status(ds_dev.handle, None) == 0
        assert (rec["prim"][0::2] == 0xFFFFFFFF).all()                 # NaN limit: nothing compares less than it


def test_an_image_is_recognised_without_the_hint(api, scene1):
    """The reference's interface has no notion of an image (rtk.h:129): a batch that IS a row-major image is recognised by its
    regular step (rtk_dev_detect_image; rtk_dev_trace_rays does the same look when no hint comes) and gets the packet kernels; the
    records are the same bytes on every path. Batches that are not images, or not whole 64x64 blocks, go the way they always did."""
    ds = api.DeviceScene.build([dict(positions=synth.scene_for_config(1))])
    frame = synth.rays_pinhole(256, 128)
    assert ds.detect_image(frame) == (256, 128)
    assert ds.detect_image(synth.rays_pinhole(192, 320, jitter=synth.frame_jitter(3))) == (192, 320)
    assert ds.detect_image(synth.rays_pinhole(200, 96)) == (200, 96)            # recognised, but not whole blocks: traced per lane
    assert ds.detect_image(synth.rays_incoherent(32768)) == (0, 0)
    assert ds.detect_image(synth.rays_config1(32768)) == (0, 0)
    two = np.concatenate([synth.rays_pinhole(128, 64), synth.rays_pinhole(256, 96)])     # two images glued together are not one
    assert ds.detect_image(two) == (0, 0)
    hinted = ds.trace(frame, opts=api.make_opts(image=(256, 128)), full=False)
    for opts in (None, api.make_opts(), api.make_opts(no_detect=True)):
        assert ds.trace(frame, opts=opts, full=False).tobytes() == hinted.tobytes()
    # the look really switches kernels: with it the packet kernels' tile counter moves (the counting call needs the hint, so the
    # check is indirect: the per-lane kernel and the packet kernel agree, and a batch that is no image is untouched by the look)
    inc = synth.rays_incoherent(32768)
    assert ds.trace(inc, full=False).tobytes() == ds.trace(inc, opts=api.make_opts(no_detect=True), full=False).tobytes()


def test_any_hit_on_an_image_takes_the_packet_kernels(api, scene2):
    """rtk_dev_trace_rays_any with an image hint (whole 64 x 64 blocks): "is there a hit in (min_t, max_t)" is answered by the
    closest-hit packet kernels (records into a stream-ordered temporary, one pass to flags) -- the same flags as the per-lane
    any-hit kernel without the hint and as the closest-hit records, with bounded intervals that cut the
    scene (max_t inside it, min_t behind the first surface)."""
    _, ds = scene2
    frame = synth.rays_pinhole(512, 256)
    frame["max_t"] = np.float32(2.2)                                   # the camera is 1.5 in front of the unit cube: ends inside it
    frame["min_t"][::3] = np.float32(1.9)
    opts = api.make_opts(image=(512, 256))
    occ_packets = ds.trace_any(frame, opts=opts)
    occ_lanes = ds.trace_any(frame, opts=api.make_opts(no_detect=True))      # (without the hint the batch would be recognised as an image)
    assert (ds.trace_any(frame) == occ_packets).all()
    rec = ds.trace(frame, opts=opts, full=False)
    assert 0.2 < occ_packets.mean() < 0.95
    assert (occ_packets == occ_lanes).all() and (occ_packets == (rec["prim"] != 0xFFFFFFFF)).all()
    # rays full of special values (zero and denormal direction components, empty and reversed intervals, NaN min_t, ...) as an "image":
    # whatever the packet kernels do with them (most tiles are handed back), the flags are the per-lane kernel's
    exotic = np.tile(synth.rays_exotic(2048), 4)
    assert len(exotic) == 128 * 64
    e_opts = api.make_opts(image=(128, 64))
    assert (ds.trace_any(exotic, opts=e_opts) == ds.trace_any(exotic, opts=api.make_opts(no_detect=True))).all()
    assert (ds.trace_any(exotic, opts=e_opts) == (ds.trace(exotic, opts=e_opts, full=False)["prim"] != 0xFFFFFFFF)).all()
    ragged = synth.rays_pinhole(200, 96)                               # not whole blocks: per lane, same answers as closest hit
    assert (ds.trace_any(ragged, opts=api.make_opts(image=(200, 96))) == (ds.trace(ragged, full=False)["prim"] != 0xFFFFFFFF)).all()


def test_the_references_leaf_sizes_stay_on_the_hand_written_kernels(api, oracle, scene2):
    """The oracle's SAH builder makes leaves of 4 to 63 triangles like the reference's (rtk.c:6-7). Their full groups of four take the
    float edge functions in rtk_packet_beam2 and rtk_lane_hot_closest (the padded last group double precision), so such scenes --
    every imported blob, every rtk_cpu_build.cpp scene -- are not handed to the C++ kernels any more: no tile of a 1024 x 1024 frame
    comes back, and frame and incoherent rays are the oracle's answers bit for bit on the same leaves."""
    blob, ds = scene2
    assert ds.info()["num_triangles"] == 1_000_000
    frame = synth.rays_pinhole(1024, 1024)
    opts = api.make_opts(image=(1024, 1024))
    rec, pk = ds.trace_packet_counted(frame, opts)
    assert pk["tiles"] == 16384 and pk["tiles_handed_back"] <= 16, pk         # (the tiles across the image axes have rays of both signs; an exact zero in a full group would hand a pair back too)
    assert pk["triangle_group_tests"] > 4 * pk["pairs"]
    sel = np.arange(0, len(frame), 7)
    oh, om = oracle.trace(blob, np.ascontiguousarray(frame[sel]))
    g = rec[sel]
    assert ((g["prim"] != 0xFFFFFFFF) == om).all()
    assert (g["t"][om] == oh["t"][om]).all() and (g["u"][om] == oh["u"][om]).all() and (g["v"][om] == oh["v"][om]).all()
    inc = synth.rays_incoherent(1 << 18)
    r2 = ds.trace(inc, full=False)
    oh, om = oracle.trace(blob, inc)
    assert ((r2["prim"] != 0xFFFFFFFF) == om).all()
    assert (r2["t"][om] == oh["t"][om]).all() and (r2["u"][om] == oh["u"][om]).all() and (r2["v"][om] == oh["v"][om]).all()
    assert r2.tobytes() == ds.trace(inc, opts=api.make_opts(no_asm=True), full=False).tobytes()
