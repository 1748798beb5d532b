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
