"""CPU tests of the task-graph builder (reference rtk.h:108-120, rtk.c:1362-1507): the blob it
emits is a valid rtk scene (oracle validator), the oracle traversing it reproduces the fixtures made by the REAL
rtk.c, the graph can be driven by several host threads at once, and indexed meshes come out with de-duplicated
vertex groups. No GPU: building is host work; tracing the blob on the device is covered in test_gpu_build.py."""
import ctypes as C
import threading

import numpy as np
import pytest

from rtk_amd import api, synth
from rtk_amd.types import MeshSet, RAY_DTYPE, SceneHeader, Task
from tests.util import compare_hits_struct, load_golden

RTK_AMD_BUILDER_DEVICE, RTK_AMD_BUILDER_CPU_TASKS = 0, 1


@pytest.fixture()
def cpu_builder():
    L = api.lib()
    assert L.rtk_amd_set_builder(RTK_AMD_BUILDER_CPU_TASKS) == 0
    yield L
    L.rtk_amd_set_builder(RTK_AMD_BUILDER_DEVICE)


def _finish(L, oracle, b):
    size = L.rtk_get_build_size(b)
    assert size > 0, api.last_error()
    buf = oracle._aligned_bytes(size)
    assert not L.rtk_finish_build_to(b, buf.ctypes.data, size - 1)          # too small: NULL, build stays alive (rtk.c:1735)
    assert L.rtk_finish_build_to(b, buf.ctypes.data, size) == buf.ctypes.data
    return oracle.Blob(buf)


def _run_serial(L, first, capacity=256):
    """The host's side of the contract: run tasks until none is pending (rtk.c:1692-1717)."""
    pending, ran = [first], 0
    spawned = (Task * capacity)()
    while pending:
        t = pending.pop()
        n = L.rtk_run_task(C.byref(t), spawned, capacity)
        ran += 1
        for k in range(n):
            c = Task()
            C.memmove(C.byref(c), C.byref(spawned[k]), C.sizeof(Task))
            pending.append(c)
    return ran


def test_task_graph_build_matches_the_reference_fixture(cpu_builder, oracle, golden_dir):
    L = cpu_builder
    tris = synth.scene_for_config(1)
    ms = MeshSet([dict(positions=tris)])
    first = Task()
    b = L.rtk_start_build(C.byref(ms.desc), C.byref(first))
    assert b and first.build == b and first.fn
    assert L.rtk_get_build_size(b) == 0                                      # nothing has run yet
    ran = _run_serial(L, first)
    assert ran > 12                                                          # 1 start + 10 setup ranges + node tasks + finalize
    blob = _finish(L, oracle, b)
    rc, counts = oracle.validate_blob(blob)
    assert rc == 0 and counts["tris"] == 10000
    rays = synth.rays_config1(65536)
    hits, mask = oracle.trace(blob, rays)
    compare_hits_struct(hits, mask, load_golden(golden_dir, "cfg1_full.npz"), "cpu task builder vs reference fixture")
    # the product's own loader accepts it (host-side validation; then fails loudly for lack of a GPU here)
    import torch
    if not torch.cuda.is_available():
        assert not L.rtk_dev_scene_upload_buffer(C.c_void_p(blob.ptr), blob.size)
        assert "HIP device" in api.last_error()


def test_task_graph_runs_on_many_threads(cpu_builder, oracle):
    """rtk_run_task from 6 threads sharing one work list (ctypes releases the GIL in the call): same hits as the
    serially scheduled build, whatever the interleaving."""
    L = cpu_builder
    tris = synth.triangle_soup(200_000, 0.03, seed=12)
    rays = synth.rays_config1(16384)

    def build(threads):
        ms = MeshSet([dict(positions=tris)])
        first = Task()
        b = L.rtk_start_build(C.byref(ms.desc), C.byref(first))
        assert b
        if threads == 1:
            _run_serial(L, first)
        else:
            lock, work, idle = threading.Lock(), [first], [0]

            def worker():
                spawned = (Task * 192)()
                while True:
                    with lock:
                        t = work.pop() if work else None
                        if t is None:
                            if idle[0] == 0:
                                return                                   # nothing queued and nobody running: done
                        else:
                            idle[0] += 1
                    if t is None:
                        continue
                    n = L.rtk_run_task(C.byref(t), spawned, 192)
                    new = []
                    for k in range(n):
                        c = Task()
                        C.memmove(C.byref(c), C.byref(spawned[k]), C.sizeof(Task))
                        new.append(c)
                    with lock:
                        work.extend(new)
                        idle[0] -= 1
            ts = [threading.Thread(target=worker) for _ in range(threads)]
            [t.start() for t in ts]
            [t.join() for t in ts]
        blob = _finish(L, oracle, b)
        assert oracle.validate_blob(blob)[0] == 0
        return oracle.trace(blob, rays)

    h1, m1 = build(1)
    h6, m6 = build(6)
    assert (m1 == m6).all() and m1.sum() > 1000
    for k in ("triangle_index", "t", "u", "v"):
        assert (h1[k][m1] == h6[k][m6]).all()


def test_rtk_build_scene_inline_path_and_vertex_group_dedup(cpu_builder, oracle, golden_dir):
    """rtk_build_scene (start with no task -> the library schedules the graph itself, rtk.c:1682-1688) on an indexed
    grid mesh: vertices shared by neighbouring triangles are stored once per group (rtk.c:1186-1360), and the
    multi-mesh edge scene (u16 + u32 indices, float64 positions) reproduces the reference fixture."""
    L = cpu_builder
    n = 96
    gx, gy = np.meshgrid(np.arange(n + 1, dtype=np.float32), np.arange(n + 1, dtype=np.float32))
    verts = np.stack([gx.ravel() / n, gy.ravel() / n, (np.sin(gx.ravel()) * 0.01).astype(np.float32) + 1.0], axis=1).astype(np.float32)
    quads = (np.arange(n)[:, None] * (n + 1) + np.arange(n)[None, :]).ravel()
    idx = np.concatenate([np.stack([quads, quads + 1, quads + n + 1], 1), np.stack([quads + 1, quads + n + 2, quads + n + 1], 1)]).astype(np.uint32)
    ms = MeshSet([dict(positions=verts, indices=idx)])
    p = L.rtk_build_scene(C.byref(ms.desc))
    assert p, api.last_error()
    try:
        hdr = SceneHeader.from_address(p)
        blob = oracle.Blob(np.ctypeslib.as_array((C.c_uint8 * hdr.size_in_bytes).from_address(p)).copy())
        rc, counts = oracle.validate_blob(blob)
        assert rc == 0 and counts["tris"] == len(idx)
        stored_vertices = (hdr.size_in_bytes - hdr.vertex_offset) // 16
        assert stored_vertices < 1.6 * len(verts) < 3 * len(idx) / 3            # ~1.3x the mesh's vertices, not 3 per triangle
        rays = np.zeros(4096, RAY_DTYPE)
        u = synth.u01(5, 0, 8192).reshape(-1, 2)
        rays["origin"][:, 0] = u[:, 0]; rays["origin"][:, 1] = u[:, 1]; rays["origin"][:, 2] = 0
        rays["direction"][:, 2] = 1
        rays["max_t"] = 10
        hits, mask = oracle.trace(blob, rays)
        assert mask.mean() > 0.99
        # the hit triangle really contains the ray's (x, y), and reports the caller's vertex indices
        tri = idx[hits["triangle_index"][mask]]
        assert (np.sort(hits["vertex"]["index"][mask], axis=1) == np.sort(tri, axis=1)).all()
    finally:
        L.rtk_free_scene(C.c_void_p(p))
    g = load_golden(golden_dir, "edge_cases.npz")
    rays = np.ascontiguousarray(g["rays"]).view(RAY_DTYPE).reshape(-1)
    t0 = g["tris"][g["mesh"] == 0].reshape(-1, 3)
    t1 = g["tris"][g["mesh"] == 1].reshape(-1, 3)
    v0, inv0 = np.unique(t0, axis=0, return_inverse=True)
    v1, inv1 = np.unique(t1, axis=0, return_inverse=True)
    ms = MeshSet([dict(positions=v0.astype(np.float64), indices=inv0.reshape(-1, 3).astype(np.uint16)),
                  dict(positions=v1.astype(np.float32), indices=inv1.reshape(-1, 3).astype(np.uint32))])
    p = L.rtk_build_scene(C.byref(ms.desc))
    assert p
    try:
        hdr = SceneHeader.from_address(p)
        blob = oracle.Blob(np.ctypeslib.as_array((C.c_uint8 * hdr.size_in_bytes).from_address(p)).copy())
        assert oracle.validate_blob(blob)[0] == 0
        hits, mask = oracle.trace(blob, rays)
        compare_hits_struct(hits, mask, g, "cpu task builder, edge scene")
    finally:
        L.rtk_free_scene(C.c_void_p(p))


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 64, 257])
def test_tiny_scenes_through_the_task_graph(cpu_builder, oracle, n):
    L = cpu_builder
    tris = synth.triangle_soup(max(n, 1), 0.5, seed=9)[:3 * n]
    ms = MeshSet([dict(positions=tris)])
    first = Task()
    b = L.rtk_start_build(C.byref(ms.desc), C.byref(first))
    assert b
    _run_serial(L, first, capacity=2)                                        # a queue of 2: overflowing tasks run in place
    blob = _finish(L, oracle, b)
    rc, counts = oracle.validate_blob(blob)
    assert rc == 0 and counts["tris"] == n
    rays = synth.rays_config1(2048)
    hits, mask = oracle.trace(blob, rays)
    if n == 0:
        assert not mask.any()
        return
    ohits, omask = oracle.trace_chain(oracle.leaf_chain_blobs(tris.reshape(-1, 3, 3)), rays)
    assert (mask == omask).all() and (hits["triangle_index"][mask] == ohits["triangle_index"][omask]).all()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_mixed_meshes_through_the_task_graph(cpu_builder, oracle, seed):
    """Random scenes of several meshes with mixed index types (implicit, u16, u32), float32 / float64 positions,
    strided buffers and callback meshes: the task-graph build, traversed by the oracle, gives the hits of the
    REFERENCE-style leaf chain over the same triangles, with the caller's (mesh, triangle) identity and vertex indices."""
    from tests.util import random_mixed_scene
    L = cpu_builder
    desc, keep, tris, mesh_index, tri_index, vidx = random_mixed_scene(seed)
    first = Task()
    b = L.rtk_start_build(C.byref(desc), C.byref(first))
    assert b, api.last_error()
    _run_serial(L, first, capacity=[3, 16, 256][seed % 3])
    blob = _finish(L, oracle, b)
    rc, counts = oracle.validate_blob(blob)
    assert rc == 0 and counts["tris"] == len(tris)
    rays = synth.rays_config1(4096, seed=seed + 20)
    hits, mask = oracle.trace(blob, rays)
    chain = oracle.leaf_chain_blobs(tris, mesh_index, tri_index, vidx)
    ohits, omask = oracle.trace_chain(chain, rays)
    assert (mask == omask).all() and mask.sum() > 100
    for k in ("mesh_index", "triangle_index"):
        assert (hits[k][mask] == ohits[k][omask]).all()
    assert np.allclose(hits["t"][mask], ohits["t"][omask], rtol=1e-5, atol=0)
    assert (np.sort(hits["vertex"]["index"][mask], axis=1) == np.sort(ohits["vertex"]["index"][omask], axis=1)).all()


def _adverse_scenes():
    from tests.test_gpu_sizes import _degenerate_mix
    scenes = {"mix%d" % s: _degenerate_mix(s) for s in (0, 3, 5, 8, 13)}
    xs = np.arange(3000, dtype=np.float32) * np.float32(1e-3)
    row = np.zeros((3000, 3, 3), np.float32)
    row[:, 0, 0] = xs; row[:, 1, 0] = xs + np.float32(3e-4); row[:, 2, 0] = xs + np.float32(6e-4)
    scenes["zero_area_row"] = np.ascontiguousarray(row.reshape(-1, 3))
    nf = synth.triangle_soup(3000, 0.05, seed=5).reshape(3000, 3, 3).copy()
    k = np.arange(3000)
    nf[k % 97 == 0, 0, 0] = np.nan
    nf[k % 101 == 1] = np.nan
    nf[k % 103 == 2, 1, 2] = np.inf
    nf[k % 107 == 3, 2, 1] = -np.inf
    scenes["non_finite"] = np.ascontiguousarray(nf.reshape(-1, 3))
    return scenes


@pytest.mark.parametrize("name", ["mix0", "mix3", "mix5", "mix8", "mix13", "zero_area_row", "non_finite"])
def test_task_graph_builder_survives_degenerate_input(cpu_builder, oracle, name):
    """Duplicates, points, needles, axis-flat and collinear geometry, non-finite vertices: the task graph terminates and
    the blob is structurally valid (every triangle in exactly one leaf, leaves of at most 63, offsets in range)."""
    L = cpu_builder
    tris = _adverse_scenes()[name]
    ms = MeshSet([dict(positions=tris)])
    first = Task()
    b = L.rtk_start_build(C.byref(ms.desc), C.byref(first))
    assert b
    _run_serial(L, first)
    blob = _finish(L, oracle, b)
    assert oracle.validate_blob(blob)[0] == 0, name
    rays = synth.rays_config1(512)
    oracle.trace(blob, rays)          # terminates


def test_cost_knobs_change_the_leaves_not_the_hits(oracle):
    """RTK_AMD_CPU_SAH_SPLIT_COST / RTK_AMD_CPU_LEAF_MIN (read once per process: a child process) make the task-graph builder
    split down to single triangles -- the tree bench.py --bvh cpu-sah uploads -- and the oracle finds the same hits in it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
from rtk_amd import api, synth
from rtk_amd.types import MeshSet, SceneHeader
from oracle import pyoracle
tris = synth.triangle_soup(4000, 0.05, 3)
L = api.lib(); ms = MeshSet([dict(positions=tris)])
L.rtk_amd_set_builder(1); p = L.rtk_build_scene(C.byref(ms.desc)); L.rtk_amd_set_builder(0)
assert p, api.last_error()
hdr = SceneHeader.from_address(p)
blob = pyoracle.Blob(np.ctypeslib.as_array((C.c_uint8 * hdr.size_in_bytes).from_address(p)).copy())
rc, counts = pyoracle.validate_blob(blob)
hits, mask = pyoracle.trace(blob, synth.rays_config1(2048))
print(rc, counts["leaves"], counts["tris"], int(mask.sum()), int(hits["triangle_index"][mask].astype(np.int64).sum()), float(hits["t"][mask].astype(np.float64).sum()))
""" % root
    outs = []
    for env in ({}, {"RTK_AMD_CPU_SAH_SPLIT_COST": "0.5", "RTK_AMD_CPU_LEAF_MIN": "1"}):
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1].split())
    (rc0, leaves0, tris0, *hits0), (rc1, leaves1, tris1, *hits1) = outs
    assert rc0 == "0" and rc1 == "0" and tris0 == tris1 == "4000"
    assert int(leaves1) > 0.95 * 4000 > int(leaves0)            # single-triangle leaves with the knobs, ~3 per leaf without
    # the same triangles from either tree; t differs in its last bits where a triangle sits in a different group of four
    # (rtk.c:302-336 computes partial groups in double precision)
    assert hits0[:2] == hits1[:2]
    assert abs(float(hits0[2]) - float(hits1[2])) <= 1e-6 * abs(float(hits0[2]))
