"""GPU tests of the remaining rows of the drop-in boundary: mesh ingestion
through callbacks and strided buffers (rtk.c:1028-1114), the log callback slot, the filter entry
point (rtk.h:117,130), error behaviour."""
import ctypes as C

import numpy as np
import pytest

from rtk_amd import synth
from rtk_amd.types import HIT_DTYPE, RAY_DTYPE, Mesh, SceneDesc, RTK_TYPE_F32, RTK_TYPE_U32

pytestmark = pytest.mark.gpu

POS_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Mesh), C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_size_t)
IDX_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Mesh), C.POINTER(C.c_uint32), C.c_size_t, C.c_size_t)
LOG_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_char_p)
FILTER_CB = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_void_p, C.c_void_p)


def _trace(api, ds, rays):
    return ds.trace(rays, full=True)


def test_callback_mesh_equals_buffer_mesh(api):
    """position_cb + index_cb (<=128 triangles per call, rtk.c:1141-1148) give the same scene as buffers."""
    tris = synth.triangle_soup(3000, 0.05, seed=21)
    verts, inv = np.unique(tris, axis=0, return_inverse=True)
    idx = inv.reshape(-1, 3).astype(np.uint32)
    calls = {"pos": 0, "idx": 0, "max_count": 0}

    def pos_cb(user, mesh, dst, indices, count):
        calls["pos"] += 1
        calls["max_count"] = max(calls["max_count"], count)
        ii = np.ctypeslib.as_array(indices, shape=(3 * count,))
        np.ctypeslib.as_array(dst, shape=(3 * count, 3))[:] = verts[ii]

    def idx_cb(user, mesh, dst, offset, count):
        calls["idx"] += 1
        np.ctypeslib.as_array(dst, shape=(3 * count,))[:] = idx[offset:offset + count].reshape(-1)

    pcb, icb = POS_CB(pos_cb), IDX_CB(idx_cb)
    m = Mesh()
    m.num_triangles = len(idx)
    m.position_cb = C.cast(pcb, C.c_void_p)
    m.index_cb = C.cast(icb, C.c_void_p)
    desc = SceneDesc()
    arr = (Mesh * 1)(m)
    desc.meshes = C.cast(arr, C.POINTER(Mesh))
    desc.num_meshes = 1
    h = api.lib().rtk_dev_scene_build(C.byref(desc))
    assert h, api.last_error()
    ds_cb = api.DeviceScene(h, keepalive=(pcb, icb, arr))
    assert calls["pos"] == calls["idx"] == (len(idx) + 127) // 128 and calls["max_count"] <= 128
    ds_buf = api.DeviceScene.build([dict(positions=verts.astype(np.float32), indices=idx)])
    rays = synth.rays_config1(8192)
    h1, m1, r1 = _trace(api, ds_cb, rays)
    h2, m2, r2 = _trace(api, ds_buf, rays)
    assert r1.tobytes() == r2.tobytes()
    assert (h1["vertex"]["index"][m1] == h2["vertex"]["index"][m2]).all()   # caller's vertex indices come back


def test_strided_interleaved_buffers(api):
    """Positions inside a 32-byte vertex struct, indices inside a 16-byte record (stride != 0)."""
    tris = synth.triangle_soup(2000, 0.05, seed=22)
    nv = len(tris)
    vbuf = np.zeros(nv, dtype=[("pad0", "<f4"), ("pos", "<f4", (3,)), ("uv", "<f4", (2,)), ("pad1", "<u4", (2,))])
    assert vbuf.itemsize == 32
    vbuf["pos"] = tris
    ibuf = np.zeros(nv // 3, dtype=[("i", "<u4", (3,)), ("material", "<u4")])
    ibuf["i"] = np.arange(nv, dtype=np.uint32).reshape(-1, 3)
    m = Mesh()
    m.num_triangles = nv // 3
    m.position.data = vbuf.ctypes.data + 4
    m.position.stride = 32
    m.position.type = RTK_TYPE_F32
    m.index.data = ibuf.ctypes.data
    m.index.stride = 16
    m.index.type = RTK_TYPE_U32
    arr = (Mesh * 1)(m)
    desc = SceneDesc()
    desc.meshes = C.cast(arr, C.POINTER(Mesh))
    desc.num_meshes = 1
    h = api.lib().rtk_dev_scene_build(C.byref(desc))
    assert h, api.last_error()
    ds = api.DeviceScene(h, keepalive=(vbuf, ibuf, arr))
    ref = api.DeviceScene.build([dict(positions=tris)])
    rays = synth.rays_config1(8192)
    assert ds.trace(rays, full=False).tobytes() == ref.trace(rays, full=False).tobytes()


def test_log_callback_slot_is_used(api):
    lines = []
    cb = LOG_CB(lambda user, build, s: lines.append(s))
    from rtk_amd.types import MeshSet
    ms = MeshSet([dict(positions=synth.triangle_soup(100, 0.1, seed=3))])
    ms.desc.log_fn = C.cast(cb, C.c_void_p)
    h = api.lib().rtk_dev_scene_build(C.byref(ms.desc))
    assert h and len(lines) >= 1
    api.lib().rtk_dev_scene_free(C.c_void_p(h))


def test_trace_ray_filter_returns_closest_accepted_hit(api, oracle):
    """rtk_trace_ray_filter: the filter sees candidates in increasing t; rejecting the first k returns hit k+1."""
    L = api.lib()
    L.rtk_trace_ray_filter.restype = C.c_bool
    L.rtk_trace_ray_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    # three parallel triangles behind each other
    tris = np.array([[[0, 0, z], [1, 0, z], [0, 1, z]] for z in (1.0, 2.0, 3.0)], np.float32).reshape(-1, 3)
    scene, keep = api.build_scene([dict(positions=tris)])
    try:
        ray = np.zeros(1, RAY_DTYPE)
        ray["origin"] = (0.25, 0.25, 0)
        ray["direction"] = (0, 0, 1)
        ray["max_t"] = 100.0
        seen = []

        def make(skip):
            def f(user, r, h):
                hit = np.ctypeslib.as_array((C.c_uint8 * 68).from_address(h)).view(HIT_DTYPE)[0]
                seen.append(int(hit["triangle_index"]))
                return len(seen) > skip
            return FILTER_CB(f)
        for skip, want in ((0, 0), (1, 1), (2, 2)):
            seen.clear()
            out = np.zeros(1, HIT_DTYPE)
            cb = make(skip)
            ok = L.rtk_trace_ray_filter(C.c_void_p(scene), ray.ctypes.data, out.ctypes.data, C.cast(cb, C.c_void_p), None)
            assert ok and out["triangle_index"][0] == want and seen == list(range(want + 1))
            assert out["t"][0] == float(want + 1)
        seen.clear()
        cb = make(3)
        out = np.zeros(1, HIT_DTYPE)
        assert not L.rtk_trace_ray_filter(C.c_void_p(scene), ray.ctypes.data, out.ctypes.data, C.cast(cb, C.c_void_p), None)
        # NULL filter == rtk_trace_ray
        assert L.rtk_trace_ray_filter(C.c_void_p(scene), ray.ctypes.data, out.ctypes.data, None, None) and out["triangle_index"][0] == 0
    finally:
        api.free_scene(scene)


def test_error_paths_are_loud(api):
    L = api.lib()
    assert not L.rtk_dev_scene_upload(None)
    assert "NULL" in api.last_error() or "scene" in api.last_error()
    assert not L.rtk_build_scene(None)
    assert L.rtk_dev_trace_rays(None, None, 10, None, None, None) != 0
    ds = api.DeviceScene.build([dict(positions=synth.triangle_soup(10, 0.1, seed=1))])
    assert L.rtk_dev_trace_rays(ds.handle, None, 10, None, None, None) != 0   # NULL rays with n > 0
    assert L.rtk_dev_scene_export(ds.handle, None, 0) is None
    small = np.zeros(16, np.uint8)
    assert L.rtk_dev_scene_export(ds.handle, small.ctypes.data, small.size) is None


def test_equal_t_candidates_are_all_offered(api, golden_dir):
    """The edge scene holds exact duplicates (primitives 0, 8 and 9 are the same triangle): a filter that rejects
    the first two must still be offered the third at the SAME t, in primitive-id order."""
    from tests.util import load_golden
    g = load_golden(golden_dir, "edge_cases.npz")
    tris = g["tris"].reshape(-1, 3, 3)
    dup = [i for i in range(len(tris)) if (tris[i] == tris[0]).all()]
    assert dup == [0, 8, 9]
    scene, keep = api.build_scene([dict(positions=g["tris"].reshape(-1, 3))])
    try:
        c = tris[0].mean(axis=0)
        ray = np.zeros(1, RAY_DTYPE)
        ray["origin"] = c + np.array([0.01, -0.02, 1.0], np.float32)
        ray["direction"] = -np.array([0.01, -0.02, 1.0], np.float32)
        ray["max_t"] = 10.0
        offered = []

        def reject_first(k):
            def f(i, hit):
                offered.append((int(hit["triangle_index"]), float(hit["t"])))
                return len(offered) > k
            return f
        for k in range(3):
            offered.clear()
            hits, mask = api.trace_rays_filter(scene, ray, reject_first(k))
            assert mask[0] and hits["triangle_index"][0] == dup[k]
            assert [o[0] for o in offered] == dup[:k + 1]
            assert len({o[1] for o in offered}) == 1                 # one and the same t
        # the single-ray entry point goes the same way
        L = api.lib()
        offered.clear()
        cb = FILTER_CB(lambda user, r, h: (offered.append(1) or len(offered) > 2))
        out = np.zeros(1, HIT_DTYPE)
        assert L.rtk_trace_ray_filter(C.c_void_p(scene), ray.ctypes.data, out.ctypes.data, C.cast(cb, C.c_void_p), None)
        assert out["triangle_index"][0] == 9 and len(offered) == 3
    finally:
        api.free_scene(scene)


def test_host_filter_batch_equals_oracle(api, oracle):
    """rtk_trace_rays_filter on a batch: same answer as the oracle driven with the same predicate."""
    tris = synth.scene_for_config(1)
    scene, keep = api.build_scene([dict(positions=tris)])
    try:
        rays = synth.rays_config1(2048)
        pred = lambda i, hit: (int(hit["triangle_index"]) * 2654435761 >> 7) % 3 != 0   # rejects a third of all primitives
        hits, mask = api.trace_rays_filter(scene, rays, pred)
        blob = oracle.Blob(np.ascontiguousarray(api.scene_bytes(scene)))
        oh, om = oracle.trace_filtered(blob, rays, callback=pred)
        assert (mask == om).all() and mask.sum() > 500
        assert (hits["triangle_index"][mask] == oh["triangle_index"][om]).all()
        assert (hits["t"][mask] == oh["t"][om]).all() and (hits["u"][mask] == oh["u"][om]).all()
        plain, pm = api.trace_rays(scene, rays)
        assert (hits["t"][mask] >= plain["t"][mask]).all() and (hits["triangle_index"][mask] != plain["triangle_index"][mask]).any()
    finally:
        api.free_scene(scene)


def test_device_filters_equal_oracle(api, oracle):
    """Built-in device filters (rtk_dev_filter): mesh mask, per-ray ignored primitive, (t, prim) continuation --
    closest-hit and any-hit -- against the oracle driven with the equivalent filter on the exported blob."""
    a = synth.triangle_soup(4000, 0.08, seed=31)
    b = synth.triangle_soup(3000, 0.08, seed=32)
    c = synth.triangle_soup(2000, 0.08, seed=33)
    ds = api.DeviceScene.build([dict(positions=a), dict(positions=b), dict(positions=c)])
    base = ds.mesh_base()
    blob = oracle.Blob(ds.export_blob())
    rays = synth.rays_config1(8192)
    plain = ds.trace(rays, full=False)
    hit = plain["prim"] != 0xFFFFFFFF

    def check(rec, oh, om, what):
        gm = rec["prim"] != 0xFFFFFFFF
        assert (gm == om).all(), what
        oprim = base[oh["mesh_index"][om].astype(np.int64)] + oh["triangle_index"][om]
        assert (rec["prim"][gm] == oprim).all(), what
        assert (rec["t"][gm] == oh["t"][om]).all() and (rec["u"][gm] == oh["u"][om]).all() and (rec["v"][gm] == oh["v"][om]).all(), what

    # mesh mask: only meshes 0 and 2 visible
    vis = [True, False, True]
    rec = ds.trace_filtered(rays, mesh_mask=vis)
    oh, om = oracle.trace_filtered(blob, rays, mesh_mask=vis)
    check(rec, oh, om, "mesh mask")
    mesh_of = np.searchsorted(base, rec["prim"][rec["prim"] != 0xFFFFFFFF], side="right") - 1
    assert (mesh_of != 1).all() and (rec["prim"] != plain["prim"]).any()
    assert (ds.trace_filtered(rays, mesh_mask=vis, any_hit=True) == om).all()
    # a mask that covers fewer meshes than the scene has: the uncovered mesh is invisible
    rec2 = ds.trace_filtered(rays, mesh_mask=[True, False])
    oh2, om2 = oracle.trace_filtered(blob, rays, mesh_mask=[True, False, False])
    check(rec2, oh2, om2, "short mesh mask")

    # ignore the primitive each ray hit first -> the second closest candidate
    ig = plain["prim"].copy()
    rec = ds.trace_filtered(rays, ignore_prim=ig)
    pm = np.where(hit, np.searchsorted(base, np.where(hit, plain["prim"], 0), side="right") - 1, 0xFFFFFFFF).astype(np.uint32)
    pt = np.where(hit, plain["prim"] - base[np.where(hit, pm, 0).astype(np.int64)], 0).astype(np.uint32)
    oh, om = oracle.trace_filtered(blob, rays, ignore=(pm, pt))
    check(rec, oh, om, "ignore primitive")
    assert ((rec["prim"] != plain["prim"]) | ~hit).all()

    # continuation: candidates after the first hit = the same second candidate (no two candidates tie here)
    rec3 = ds.trace_filtered(rays, after=plain)
    assert rec3.tobytes() == rec.tobytes()
    oh, om = oracle.trace_filtered(blob, rays, after=(plain["t"], pm, pt))
    check(rec3, oh, om, "after")
    # enumerating with `after` until exhaustion visits candidates in strictly increasing (t, prim) order
    # (a ray that has run out is dropped here: an `after` record with prim NONE means "no restriction")
    cur, alive, steps, total = plain.copy(), hit.copy(), 0, int(hit.sum())
    while alive.any() and steps < 64:
        nxt = ds.trace_filtered(rays, after=cur)
        more = alive & (nxt["prim"] != 0xFFFFFFFF)
        assert ((nxt["t"][more] > cur["t"][more]) | ((nxt["t"][more] == cur["t"][more]) & (nxt["prim"][more] > cur["prim"][more]))).all()
        cur = np.where(more, nxt, cur)
        alive, steps, total = more, steps + 1, total + int(more.sum())
    assert 3 < steps < 64 and total > 2 * int(hit.sum())


def test_one_scene_traced_from_many_threads(api, oracle):
    """rtk_trace_rays / rtk_trace_ray from 6 host threads on ONE cached scene, and rtk_dev_trace_rays on 4
    streams at once: every result equals the single-threaded one (per-(scene, stream) launch scratch)."""
    import threading
    import torch
    tris = synth.scene_for_config(1)
    scene, keep = api.build_scene([dict(positions=tris)])
    try:
        rays = synth.rays_config1(65536)
        want_hits, want_mask = api.trace_rays(scene, rays)
        results, errors = {}, []

        def worker(k):
            try:
                for rep in range(4):
                    sl = slice(k * 8192, (k + 1) * 8192 + rep * 1000)
                    h, m = api.trace_rays(scene, rays[sl])
                    if not ((m == want_mask[sl]).all() and (h["triangle_index"][m] == want_hits["triangle_index"][sl][m]).all()
                            and (h["t"][m] == want_hits["t"][sl][m]).all()):
                        errors.append(("batch", k, rep))
                for i in range(k * 50, k * 50 + 50):
                    one = api.trace_ray(scene, rays[i])
                    if (one is not None) != bool(want_mask[i]) or (one is not None and one["triangle_index"] != want_hits["triangle_index"][i]):
                        errors.append(("single", k, i))
                results[k] = True
            except Exception as e:      # noqa: BLE001
                errors.append(repr(e))
        ts = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors and len(results) == 6, errors[:4]
    finally:
        api.free_scene(scene)
    # device-pointer API on several streams at once
    ds = api.DeviceScene.build([dict(positions=tris)])
    big = np.concatenate([synth.rays_config1(65536, seed=s) for s in (2, 7, 8, 9)])
    want = ds.trace(big, full=False)
    d_rays = api.to_device(big)
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = [torch.zeros(65536 * 16, dtype=torch.uint8, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    for rep in range(3):
        for k, st in enumerate(streams):
            with torch.cuda.stream(st):
                ds.trace_device(d_rays[k * 65536 * 32:(k + 1) * 65536 * 32], 65536, outs[k])
    torch.cuda.synchronize()
    got = np.concatenate([o.cpu().numpy() for o in outs])
    assert got.tobytes() == want.tobytes()


def test_residency_cache_notices_a_new_blob_at_the_same_address(api, oracle):
    """rtk_finish_build_to writes into caller memory; a caller that reuses the buffer for another scene must get
    the new scene traced, not the cached device copy of the old one."""
    from rtk_amd.types import MeshSet
    L = api.lib()
    rays = synth.rays_config1(2048)
    buf = oracle._aligned_bytes(4 << 20)
    answers = []
    for seed in (41, 42):
        tris = synth.triangle_soup(3000, 0.1, seed=seed)
        ms = MeshSet([dict(positions=tris)])
        b = L.rtk_start_build(C.byref(ms.desc), None)
        assert b
        size = L.rtk_get_build_size(b)
        assert 0 < size <= buf.size
        s = L.rtk_finish_build_to(b, buf.ctypes.data, buf.size)
        assert s == buf.ctypes.data
        hits, mask = api.trace_rays(s, rays)
        oh, om = oracle.trace(oracle.Blob(buf[:size]), rays)
        assert (mask == om).all() and (hits["triangle_index"][mask] == oh["triangle_index"][om]).all()
        answers.append(hits["t"][mask].tobytes())
    assert answers[0] != answers[1]
    # a blob copied over the old one by plain memcpy (no library call at all) is noticed too
    tris = synth.triangle_soup(3000, 0.1, seed=43)
    blob = oracle.build_scene([dict(positions=tris)])
    buf[:blob.size] = blob.data
    hits, mask = api.trace_rays(buf.ctypes.data, rays)
    oh, om = oracle.trace(blob, rays)
    assert (mask == om).all() and (hits["triangle_index"][mask] == oh["triangle_index"][om]).all()
    L.rtk_amd_forget_scene(C.c_void_p(buf.ctypes.data))


def test_single_process_multi_gpu_context_with_virtual_shards(api, oracle):
    """rtk_mgpu_* (the C host's form of SURVEY.md 8e): the scene replicated on every slot, the batch cut into
    contiguous ranges, records gathered by per-slot copies. One physical GPU here, so the context is made of
    three slots on device 0 -- the sharding, the piecewise trace + copy pipeline and the gather are the real code."""
    import torch
    from rtk_amd.types import HIT_RECORD_DTYPE, MeshSet
    L = api.lib()
    tris = synth.scene_for_config(1)
    devs = (C.c_int * 3)(0, 0, 0)
    m = L.rtk_mgpu_create(devs, 3)
    assert m and L.rtk_mgpu_num_devices(m) == 3
    try:
        ms = MeshSet([dict(positions=tris)])
        assert L.rtk_mgpu_build(m, C.byref(ms.desc)) == 0, api.last_error()
        # every replica is the same tree
        hashes = set()
        for i in range(3):
            c = api.SceneCheck()
            assert L.rtk_dev_scene_validate(L.rtk_mgpu_scene(m, i), C.byref(c)) == 0
            hashes.add(c.content_hash)
        assert len(hashes) == 1
        ref = api.DeviceScene.build([dict(positions=tris)])
        # host rays in, host records out; a size that does not divide by 3 and spans several pieces
        n = (1 << 22) + 12345
        rays = np.concatenate([synth.rays_config1(1 << 20, seed=s) for s in (2, 3, 4, 5, 6)])[:n]
        want = ref.trace(rays, full=False)
        got = np.zeros(n, HIT_RECORD_DTYPE)
        assert L.rtk_mgpu_trace_rays(m, rays.ctypes.data, n, got.ctypes.data, None) == 0, api.last_error()
        assert got.tobytes() == want.tobytes()
        # device-resident shards gathered onto slot 1; each shard an image-shaped frame -> packet kernel in bands
        frames = [synth.rays_pinhole(2048, 2048, jitter=synth.frame_jitter(k)) for k in range(3)]
        d_rays = [api.to_device(f) for f in frames]
        d_rec = [torch.empty(len(f) * 16, dtype=torch.uint8, device="cuda") for f in frames]
        d_all = torch.zeros(3 * 2048 * 2048 * 16, dtype=torch.uint8, device="cuda")
        rp = (C.c_void_p * 3)(*[t.data_ptr() for t in d_rays])
        cp = (C.c_size_t * 3)(*[len(f) for f in frames])
        op = (C.c_void_p * 3)(*[t.data_ptr() for t in d_rec])
        opts = api.make_opts(image=(2048, 2048))
        torch.cuda.synchronize()
        assert L.rtk_mgpu_trace_rays_device(m, rp, cp, op, C.c_void_p(d_all.data_ptr()), 1, C.byref(opts)) == 0, api.last_error()
        allrec = d_all.cpu().numpy().view(HIT_RECORD_DTYPE)
        for k, f in enumerate(frames):
            w = ref.trace(f, opts=opts, full=False)
            assert allrec[k * len(f):(k + 1) * len(f)].tobytes() == w.tobytes()
            assert d_rec[k].cpu().numpy().tobytes() == w.tobytes()
        # the striped exchange (no root: slot j receives stripe j of every shard); shards of unequal size that span pieces
        counts = [len(frames[0]), len(frames[1]) - 4096 * 8 - 0, len(frames[2])]
        counts[1] = 2048 * 1000              # a shard that is still a whole number of tile rows
        cp2 = (C.c_size_t * 3)(*counts)
        seg = [[None] * 3 for _ in range(3)]
        f_, c_ = C.c_size_t(), C.c_size_t()
        need = [0, 0, 0]
        for j in range(3):
            for r in range(3):
                L.rtk_mgpu_striped_segment(cp2, 3, r, j, C.byref(f_), C.byref(c_))
                seg[j][r] = (f_.value, c_.value)
                need[j] = f_.value + c_.value
        assert sum(need) == sum(counts)
        d_str = [torch.full((need[j] * 16,), 0x5a, dtype=torch.uint8, device="cuda") for j in range(3)]
        sp = (C.c_void_p * 3)(*[t.data_ptr() for t in d_str])
        for t in d_rec:
            t.zero_()
        plain = api.make_opts()
        torch.cuda.synchronize()
        assert L.rtk_mgpu_trace_rays_device_striped(m, rp, cp2, op, sp, C.byref(plain)) == 0, api.last_error()
        from rtk_amd import shard as shard_py
        for r, f in enumerate(frames):
            w = ref.trace(f[:counts[r]], full=False)
            assert d_rec[r].cpu().numpy()[:counts[r] * 16].tobytes() == w.tobytes()
            for j in range(3):
                b, e = shard_py.stripe_bounds(counts[r], 3)[j]
                at, ln = seg[j][r]
                assert ln == e - b
                assert d_str[j].cpu().numpy()[at * 16:(at + ln) * 16].tobytes() == w[b:e].tobytes()
    finally:
        L.rtk_mgpu_destroy(m)


def test_striped_exchange_across_real_gpus(api):
    """rtk_mgpu_trace_rays_device_striped with every slot on a GPU of its own (needs >= 2 GPUs: skipped on the one-GPU box, run by
    whoever has a node): per-shard records and every stripe on every GPU equal the single-GPU records."""
    import torch
    from rtk_amd import shard as shard_py
    from rtk_amd.types import HIT_RECORD_DTYPE, MeshSet
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("one GPU: the virtual-shard test above covers the code, this one the peer copies")
    ndev = min(ndev, 4)
    L = api.lib()
    tris = synth.scene_for_config(1)
    devs = (C.c_int * ndev)(*range(ndev))
    m = L.rtk_mgpu_create(devs, ndev)
    assert m and L.rtk_mgpu_num_devices(m) == ndev
    try:
        ms = MeshSet([dict(positions=tris)])
        assert L.rtk_mgpu_build(m, C.byref(ms.desc)) == 0, api.last_error()
        ref = api.DeviceScene.build([dict(positions=tris)])
        counts = [(1 << 20) + 777 * k for k in range(ndev)]
        shards = [synth.rays_config1(counts[k], seed=10 + k) for k in range(ndev)]
        d_rays, d_rec, d_str, seg = [], [], [], [[None] * ndev for _ in range(ndev)]
        cp = (C.c_size_t * ndev)(*counts)
        f_, c_ = C.c_size_t(), C.c_size_t()
        for j in range(ndev):
            need = 0
            for r in range(ndev):
                L.rtk_mgpu_striped_segment(cp, ndev, r, j, C.byref(f_), C.byref(c_))
                seg[j][r] = (f_.value, c_.value)
                need = f_.value + c_.value
            dev = torch.device("cuda", j)
            d_rays.append(torch.from_numpy(shards[j].view(np.uint8).reshape(-1)).to(dev))
            d_rec.append(torch.zeros(counts[j] * 16, dtype=torch.uint8, device=dev))
            d_str.append(torch.full((need * 16,), 0x5a, dtype=torch.uint8, device=dev))
        for j in range(ndev):
            torch.cuda.synchronize(j)
        rp = (C.c_void_p * ndev)(*[t.data_ptr() for t in d_rays])
        op = (C.c_void_p * ndev)(*[t.data_ptr() for t in d_rec])
        sp = (C.c_void_p * ndev)(*[t.data_ptr() for t in d_str])
        plain = api.make_opts()
        assert L.rtk_mgpu_trace_rays_device_striped(m, rp, cp, op, sp, C.byref(plain)) == 0, api.last_error()
        for r in range(ndev):
            w = ref.trace(shards[r], full=False)
            assert d_rec[r].cpu().numpy().tobytes() == w.tobytes()
            for j in range(ndev):
                b, e = shard_py.stripe_bounds(counts[r], ndev)[j]
                at, ln = seg[j][r]
                assert ln == e - b
                assert d_str[j].cpu().numpy()[at * 16:(at + ln) * 16].tobytes() == w[b:e].tobytes()
    finally:
        L.rtk_mgpu_destroy(m)


def test_a_host_batch_that_is_an_image_goes_through_the_packet_kernels_in_bands(api):
    """rtk_trace_rays (host rays in, full rtk_hit out) cuts a batch into pieces for its staging buffers; a batch it recognises as a
    row-major image is cut into bands of whole 64-pixel rows, each announced to the launch as an image. 1024 x 704 pixels: eleven
    rows of blocks, so the last band is lower than the others. Every hit equals what the per-ray call (served on the host, an
    independent walk of the same blob) returns for that ray."""
    tris = synth.triangle_soup(200_000, 0.03, seed=5)
    scene, keep = api.build_scene([dict(positions=tris)])
    try:
        frame = synth.rays_pinhole(1024, 704)
        hits, mask = api.trace_rays(scene, frame)
        assert 0.5 < mask.mean() < 1.0
        for i in np.random.RandomState(1).randint(0, len(frame), 3000):
            one = api.trace_ray(scene, frame[i])
            assert (one is not None) == bool(mask[i]), i
            assert one is None or one.tobytes() == hits[i].tobytes(), i
        # a batch that is no image: pieces of 32 k rays as before
        inc = synth.rays_incoherent(100_000)
        h2, m2 = api.trace_rays(scene, inc)
        for i in range(0, 100_000, 97):
            one = api.trace_ray(scene, inc[i])
            assert (one is not None) == bool(m2[i]) and (one is None or one.tobytes() == h2[i].tobytes()), i
    finally:
        api.free_scene(scene)


def test_filter_rejection_chains_longer_than_one_launch_collects(api, oracle):
    """A stack of 150 parallel triangles: a single ray collects 64 candidates per launch, a 16k-ray batch 4 per
    launch; rejecting the first 100 candidates of every ray needs several rounds and still returns candidate 101,
    offered in order, exactly once each; same answer as the oracle with the same predicate."""
    zs = 1.0 + 0.01 * np.arange(150, dtype=np.float32)
    tris = np.array([[[-1, -1, z], [3, -1, z], [-1, 3, z]] for z in zs], np.float32).reshape(-1, 3)
    scene, keep = api.build_scene([dict(positions=tris)])
    try:
        ray = np.zeros(1, RAY_DTYPE)
        ray["origin"] = (0.25, 0.25, 0)
        ray["direction"] = (0, 0, 1)
        ray["max_t"] = 100.0
        offered = []

        def pred(i, hit):
            offered.append(int(hit["triangle_index"]))
            return int(hit["triangle_index"]) >= 100
        hits, mask = api.trace_rays_filter(scene, ray, pred)
        assert mask[0] and hits["triangle_index"][0] == 100 and offered == list(range(101))
        assert hits["t"][0] == zs[100]
        # a batch: every ray rejects up to its own threshold
        n = 16384
        rays = np.zeros(n, RAY_DTYPE)
        u = synth.u01(3, 0, 2 * n).reshape(n, 2)
        rays["origin"][:, 0] = u[:, 0] * np.float32(0.5); rays["origin"][:, 1] = u[:, 1] * np.float32(0.5)
        rays["direction"][:, 2] = 1
        rays["max_t"] = 100.0
        thr = (np.arange(n) * 7) % 160                      # some thresholds lie beyond the last triangle: those rays miss
        hits, mask = api.trace_rays_filter(scene, rays, lambda i, hit: int(hit["triangle_index"]) >= thr[i])
        assert (mask == (thr < 150)).all()
        assert (hits["triangle_index"][mask] == thr[mask]).all()
        blob = oracle.Blob(np.ascontiguousarray(api.scene_bytes(scene)))
        oh, om = oracle.trace_filtered(blob, rays[:512], callback=lambda i, hit: int(hit["triangle_index"]) >= thr[i])
        assert (om == mask[:512]).all() and (oh["triangle_index"][om] == hits["triangle_index"][:512][om]).all()
        assert (oh["t"][om] == hits["t"][:512][om]).all()
    finally:
        api.free_scene(scene)


def test_builds_and_traces_from_several_threads_at_once(api, oracle):
    """Four host threads each build their own scenes (device builder, shared workspace behind a mutex), trace them through
    the host-pointer and the device-pointer calls (pipelined pieces included) and free them, all at the same time: every
    thread gets what a single-threaded run gets."""
    import threading
    sizes = [3000, 20000, 70000, 1500]
    rays = synth.rays_config1(140000)
    expect = {}
    for k, n in enumerate(sizes):
        tris = synth.triangle_soup(n, 0.05, seed=60 + k)
        ds = api.DeviceScene.build([dict(positions=tris)])
        expect[k] = (ds.validate()[1]["content_hash"], ds.trace(rays, full=False).tobytes())
        ds.free()
    errors = []

    def worker(k):
        try:
            tris = synth.triangle_soup(sizes[k], 0.05, seed=60 + k)
            for rep in range(3):
                ds = api.DeviceScene.build([dict(positions=tris)])
                ok, c = ds.validate()
                if not ok or c["content_hash"] != expect[k][0]:
                    errors.append(("hash", k, rep))
                if ds.trace(rays, full=False).tobytes() != expect[k][1]:
                    errors.append(("device trace", k, rep))
                ds.free()
                scene, keep = api.build_scene([dict(positions=tris)])
                hits, mask = api.trace_rays(scene, rays)          # > 64k rays: two staging sets, two streams
                rec = np.frombuffer(expect[k][1], dtype=[("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<u4")])
                if not ((mask == (rec["prim"] != 0xFFFFFFFF)).all() and (hits["t"][mask] == rec["t"][mask]).all()
                        and (hits["triangle_index"][mask] == rec["prim"][mask]).all()):
                    errors.append(("host trace", k, rep))
                api.free_scene(scene)
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[:6]


def test_in_place_edit_of_a_cached_blob_is_found_out(api, oracle):
    """A blob edited in place behind the library's back (same size, same header, same root: one moved triangle): the
    residency cache re-checks one 4 KB stripe per lookup in rotation, so the stale device copy lives for at most
    size / 4 KB lookups; rtk_amd_forget_scene is the immediate way (include/rtk_amd.h)."""
    L = api.lib()
    tris = synth.triangle_soup(3000, 0.1, seed=77)
    blob = oracle.build_scene([dict(positions=tris)])
    buf = oracle._aligned_bytes(blob.size)
    buf[:] = blob.data
    rays = synth.rays_config1(2048)
    hits, mask = api.trace_rays(buf.ctypes.data, rays)
    k = int(np.nonzero(mask)[0][0])
    T = int(hits["triangle_index"][k])

    def move_triangle(dz):
        moved = 0
        for v in tris.reshape(-1, 3, 3)[T]:
            pat = np.asarray(v, np.float32).tobytes()
            raw = buf.tobytes()
            at = raw.find(pat)
            while at >= 0:
                if at % 16 == 0:
                    z = np.frombuffer(raw, np.float32, 1, at + 8)[0]
                    buf[at + 8:at + 12] = np.frombuffer(np.float32(z + dz).tobytes(), np.uint8)
                    moved += 1
                at = raw.find(pat, at + 1)
        return moved

    assert move_triangle(50.0) >= 3
    stripes = (blob.size + 4095) // 4096
    seen = None
    for call in range(stripes + 2):
        h2, m2 = api.trace_rays(buf.ctypes.data, rays)
        if not (m2[k] and h2["triangle_index"][k] == T):
            seen = call
            break
    assert seen is not None, "the edit was never noticed in %d lookups" % (stripes + 2)
    oh, om = oracle.trace(oracle.Blob(buf), rays)
    assert (m2 == om).all() and (h2["triangle_index"][m2] == oh["triangle_index"][om]).all()
    # the immediate way: edit back, forget, trace
    buf[:] = blob.data
    L.rtk_amd_forget_scene(C.c_void_p(buf.ctypes.data))
    h3, m3 = api.trace_rays(buf.ctypes.data, rays)
    assert m3[k] and h3["triangle_index"][k] == T
    L.rtk_amd_forget_scene(C.c_void_p(buf.ctypes.data))


def test_cpu_builder_into_a_reused_buffer_drops_the_old_device_copy(api, oracle):
    """rtk_finish_build_to with the task-graph CPU builder writes a new blob where an old one was traced from: the old
    device copy must go (only the device builder's branch re-adopted the scene)."""
    from rtk_amd.types import MeshSet
    L = api.lib()
    rays = synth.rays_config1(1024)
    buf = oracle._aligned_bytes(8 << 20)
    try:
        for seed in (51, 52):
            L.rtk_amd_set_builder(1)
            tris = synth.triangle_soup(2000, 0.1, seed=seed)
            ms = MeshSet([dict(positions=tris)])
            b = L.rtk_start_build(C.byref(ms.desc), None)
            assert b, api.last_error()
            size = L.rtk_get_build_size(b)
            assert 0 < size <= buf.size
            s = L.rtk_finish_build_to(b, buf.ctypes.data, buf.size)
            assert s == buf.ctypes.data
            hits, mask = api.trace_rays(s, rays)
            oh, om = oracle.trace(oracle.Blob(buf[:size]), rays)
            assert (mask == om).all() and (hits["triangle_index"][mask] == oh["triangle_index"][om]).all()
    finally:
        L.rtk_amd_set_builder(0)
        L.rtk_amd_forget_scene(C.c_void_p(buf.ctypes.data))


def test_bad_type_codes_are_refused(api):
    """rtk.c:1080-1113 asserts on an unknown position type; here both builders refuse the mesh."""
    from rtk_amd.types import MeshSet, RTK_TYPE_U16
    L = api.lib()
    tris = synth.triangle_soup(500, 0.1, seed=3)
    for builder in (0, 1):
        ms = MeshSet([dict(positions=tris)])
        ms._arr[0].position.type = RTK_TYPE_U16            # an index type where a position type belongs
        L.rtk_amd_set_builder(builder)
        try:
            b = L.rtk_start_build(C.byref(ms.desc), None)
            if b and builder == 0:
                # the device builder runs inside the one task of its graph: the failure shows there or at the finish
                s = L.rtk_finish_build(b)
                assert not s
            else:
                assert not b
            assert "type" in api.last_error()
        finally:
            L.rtk_amd_set_builder(0)
    assert not L.rtk_dev_scene_build(C.byref(ms.desc)) and "type" in api.last_error()


def test_planes_beyond_float_range_keep_the_scene_on_exact_nodes(api, oracle):
    """A blob whose boxes reach +-3e38 (extent not finite in float) cannot be compressed to the 8-bit grid: the scene must
    fall back to its exact nodes instead of storing wrapped planes that cull real hits."""
    tris = synth.triangle_soup(400, 0.2, seed=9)
    blob = oracle.build_scene([dict(positions=tris)])
    data = blob.data.copy()
    # stretch the root's child boxes: every finite plane of the root node becomes +-3e38 (a valid, huge box)
    root = np.frombuffer(data, np.float32, 24, 128).copy().reshape(3, 2, 4)
    occupied = root[:, 0, :] <= root[:, 1, :]
    root[:, 0, :][occupied] = np.float32(-3.0e38)
    root[:, 1, :][occupied] = np.float32(3.0e38)
    data[128:128 + 96] = np.frombuffer(root.tobytes(), np.uint8)
    ds = api.DeviceScene.upload(data)
    rays = synth.rays_config1(4096)
    rec_default = ds.trace(rays, full=False)
    rec_exact = ds.trace(rays, opts=api.make_opts(exact_nodes=True), full=False)
    assert rec_default.tobytes() == rec_exact.tobytes()
    oh, om = oracle.trace(oracle.Blob(data), rays)
    gm = rec_default["prim"] != 0xFFFFFFFF
    assert (gm == om).all() and om.sum() > 100


_FAIL_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from rtk_amd import api
from rtk_amd.types import RAY_DTYPE
tris = np.array([[0, 0, 1], [1, 0, 1], [0, 1, 1]], np.float32)
scene, keep = api.build_scene([dict(positions=tris)])
ray = np.zeros(1, RAY_DTYPE); ray["origin"] = (0.25, 0.25, 0); ray["direction"] = (0, 0, 1); ray["max_t"] = 100.0
assert api.trace_ray(scene, ray[0]) is not None
api.lib().rtk_amd_test_fail_next_calls(int(sys.argv[1]))
first = api.trace_ray(scene, ray[0])
print("first", first is None, flush=True)
second = api.trace_ray(scene, ray[0])
print("second", second is None, flush=True)
third = api.trace_ray(scene, ray[0])
print("third", third is None, flush=True)
"""


def _run_failing_per_ray_calls(n_fail, soft):
    import os
    import subprocess
    import sys
    env = dict(os.environ, RTK_AMD_TEST_HOOKS="1")       # the fault-injection hook is a no-op in a process started without this
    env.pop("RTK_AMD_SOFT_ERRORS", None)
    if soft:
        env["RTK_AMD_SOFT_ERRORS"] = "1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "-c", _FAIL_SNIPPET % root, str(n_fail)], env=env, capture_output=True, text=True, timeout=300)


def test_a_failed_per_ray_call_is_reported_not_passed_off_as_a_miss(api):
    """rtk_trace_ray has no error channel and `false` means "miss": a failure prints a diagnostic; one that passes returns
    false once, two in a row stop the process (unless the host opted into soft failures)."""
    # one injected failure: reported on stderr, false once, then the real hit again
    p = _run_failing_per_ray_calls(1, soft=False)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "first True" in p.stdout and "second False" in p.stdout and "third False" in p.stdout
    assert "rtk_trace_ray: FAILED, not a miss (transient)" in p.stderr and "injected failure" in p.stderr
    # two in a row: the second one is fatal
    p = _run_failing_per_ray_calls(2, soft=False)
    assert p.returncode != 0 and "first True" in p.stdout and "second" not in p.stdout
    assert "second failure in a row" in p.stderr
    # ... unless the host handles failures itself: false both times, both reported, the process lives
    p = _run_failing_per_ray_calls(2, soft=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "first True" in p.stdout and "second True" in p.stdout and "third False" in p.stdout
    assert p.stderr.count("FAILED, not a miss") == 2


def test_the_fault_injection_hook_is_dead_without_its_environment_switch(api):
    """rtk_amd_test_fail_next_calls is exported for the test above only: in a process that was not started with
    RTK_AMD_TEST_HOOKS=1 it does nothing."""
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "RTK_AMD_TEST_HOOKS"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _FAIL_SNIPPET % root, "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "first False" in p.stdout and "second False" in p.stdout and "FAILED" not in p.stderr


def test_per_ray_host_path_equals_the_gpu_batch_path(api):
    """rtk_trace_ray / rtk_trace_ray_filter are served on the calling thread from the caller's blob (rtk_host_trace.cpp); the batch
    calls run on the GPU from the device copy of the same blob. Every field of every rtk_hit must agree: a device-built scene
    (leaves of <= 3 triangles: always the double-precision group) and a CPU-task-builder scene (leaves up to 63: whole groups)."""
    L = api.lib()
    tris = synth.triangle_soup(60_000, 0.04, seed=13)
    rays = np.concatenate([synth.rays_config1(3000, seed=3), synth.rays_incoherent(3000, seed=4)])
    for builder in (0, 1):
        L.rtk_amd_set_builder(builder)
        try:
            scene, keep = api.build_scene([dict(positions=tris)])
        finally:
            L.rtk_amd_set_builder(0)
        try:
            hits, mask = api.trace_rays(scene, rays)
            assert 0.3 < mask.mean() < 1.0
            for i in range(len(rays)):
                one = api.trace_ray(scene, rays[i])
                assert (one is not None) == bool(mask[i]), (builder, i)
                assert one is None or one.tobytes() == hits[i].tobytes(), (builder, i, one, hits[i])
            # the host filter loop and the GPU's rounds offer the same candidates in the same order: reject the two closest
            seen = {}

            def reject_two(i, h):
                seen[i] = seen.get(i, 0) + 1
                return seen[i] > 2
            fh, fm = api.trace_rays_filter(scene, rays[:500], reject_two)
            FILTER = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_void_p, C.c_void_p)
            L.rtk_trace_ray_filter.restype = C.c_bool
            for i in range(500):
                count = [0]

                def cb(user, ray_ptr, hit_ptr):
                    count[0] += 1
                    return count[0] > 2
                fn = FILTER(cb)
                one = np.zeros(1, HIT_DTYPE)
                ok = L.rtk_trace_ray_filter(C.c_void_p(scene), C.c_void_p(rays[i:i + 1].ctypes.data), C.c_void_p(one.ctypes.data), C.cast(fn, C.c_void_p), None)
                assert bool(ok) == bool(fm[i]) and (not ok or one[0].tobytes() == fh[i].tobytes()), (builder, i)
        finally:
            api.free_scene(scene)


@pytest.fixture
def per_ray_on_gpu(api):
    api.lib().rtk_amd_set_per_ray(1)
    yield
    api.lib().rtk_amd_set_per_ray(0)


def test_single_ray_kernel_equals_the_batch_path(api, per_ray_on_gpu):
    """rtk_amd_set_per_ray(RTK_AMD_PER_RAY_GPU): rtk_trace_ray runs a kernel of its own (one wave walks the ray's frontier breadth first, exact nodes, one launch);
    every field of its rtk_hit must be what the batch call returns for the same ray -- ordinary rays, exotic ones (zeros,
    infinities, NaN intervals: the reference's operand order decides), a scene where one ray meets thousands of boxes (the
    frontier does not fit LDS: the call falls back to the batch path), an empty scene."""
    tris = synth.triangle_soup(60_000, 0.04, seed=13)
    scene, keep = api.build_scene([dict(positions=tris)])
    try:
        rays = np.concatenate([synth.rays_config1(200, seed=3), synth.rays_incoherent(200, seed=4), synth.rays_exotic(256, seed=9, tris=tris.reshape(-1, 3, 3))])
        hits, mask = api.trace_rays(scene, rays)
        assert 0.3 < mask[:400].mean() < 1.0
        # (the batch call traces exotic rays on compressed or exact nodes depending on their wave-mates; where boxes are below
        # float resolution at the origin the two may differ, DESIGN.md 4: such rays are compared with the exact-node batch)
        diff = 0
        for i in range(len(rays)):
            one = api.trace_ray(scene, rays[i])
            if (one is not None) != bool(mask[i]) or (one is not None and one.tobytes() != hits[i].tobytes()):
                diff += 1
                assert i >= 400, (i, one, hits[i], rays[i])
        assert diff <= 2
    finally:
        api.free_scene(scene)
    # 3000 copies of one triangle, and a ray through all of them: 3000 leaves on one ray
    one_tri = np.array([[0, 0, 1], [1, 0, 1], [0, 1, 1]], np.float32)
    pile = np.tile(one_tri, (3000, 1)) + (np.arange(3000, dtype=np.float32).repeat(3) * np.float32(1e-4))[:, None] * np.array([0, 0, 1], np.float32)
    scene, keep = api.build_scene([dict(positions=pile)])
    try:
        ray = np.zeros(1, RAY_DTYPE)
        ray["origin"] = (0.25, 0.25, 0)
        ray["direction"] = (0, 0, 1)
        ray["max_t"] = 100.0
        hits, mask = api.trace_rays(scene, np.repeat(ray, 2))
        one = api.trace_ray(scene, ray[0])
        assert mask[0] and one is not None and one.tobytes() == hits[0].tobytes() and one["triangle_index"] == 0
    finally:
        api.free_scene(scene)
