"""The per-ray symbols of rtk.h (rtk_trace_ray, rtk_trace_ray_filter) as the product serves them: on the calling thread, from the
caller's blob (rtk_amd/csrc/rtk_host_trace.cpp). No GPU needed for any of this.

Pinned three ways: (1) against the golden fixtures the REAL rtk.c produced, through the same single-leaf blob chains the reference
traced (bit for bit: hit/miss, ids, t, u, v); (2) against the oracle on full BVH4 blobs (the oracle's SAH blob and the product's own
CPU task-graph builder's blob), bit for bit; (3) filter semantics: candidates in (t, mesh, triangle) order, all equal-t ones offered.
(The same rays against the GPU batch path on the same blob: tests/test_gpu_api_rows.py::test_per_ray_host_path_equals_the_gpu_batch_path.)"""
import ctypes as C

import numpy as np
import pytest

from rtk_amd import synth
from rtk_amd.types import HIT_DTYPE, RAY_DTYPE
from tests.util import compare_hits_struct, load_golden, sha


def _per_ray(api, blob_ptr, rays, max_t=None):
    """rtk_trace_ray once per ray through the C ABI. Returns (hits, mask)."""
    L = api.lib()
    rays = np.ascontiguousarray(rays).copy()
    if max_t is not None:
        rays["max_t"] = max_t
    hits = np.zeros(len(rays), HIT_DTYPE)
    mask = np.zeros(len(rays), bool)
    rp, hp = rays.ctypes.data, hits.ctypes.data
    fn = L.rtk_trace_ray
    scene = C.c_void_p(blob_ptr)
    for i in range(len(rays)):
        mask[i] = fn(scene, C.c_void_p(rp + 32 * i), C.c_void_p(hp + 68 * i))
    return hits, mask


def _chain(api, blobs, rays):
    """What oracle.trace_chain / the real rtk.c did for the fixtures: every blob in turn, max_t fed forward."""
    best = np.zeros(len(rays), HIT_DTYPE)
    any_hit = np.zeros(len(rays), bool)
    cur = np.ascontiguousarray(rays).copy()
    for b in blobs:
        h, m = _per_ray(api, b.ptr, cur)
        best[m] = h[m]
        any_hit |= m
        cur["max_t"][m] = h["t"][m]
    return best, any_hit


@pytest.fixture(scope="module")
def host_lib(api):
    L = api.lib()
    L.rtk_trace_ray.restype = C.c_bool
    L.rtk_trace_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    return api


def test_edge_cases_bit_exact_with_the_reference(host_lib, oracle, golden_dir):
    g = load_golden(golden_dir, "edge_cases.npz")
    rays = np.ascontiguousarray(g["rays"]).view(RAY_DTYPE).reshape(-1)
    blobs = oracle.leaf_chain_blobs(g["tris"], g["mesh"], g["tri_index"])
    hits, mask = _chain(host_lib, blobs, rays)
    st = compare_hits_struct(hits, mask, g, "edge/host per-ray")
    assert st["bit_exact"] == 1.0
    assert not mask[11] and mask[10] and hits["t"][10] == 2.0          # open interval at both ends


def test_exotic_rays_bit_exact_with_the_reference(host_lib, oracle, golden_dir):
    """Zeros, negative zeros, denormals, huge and tied direction components, NaN intervals: the ray set-up and the leaf arithmetic
    where they are most fragile, against what the REAL rtk.c returned on the same leaf chain."""
    scene1 = synth.scene_for_config(1)
    g = load_golden(golden_dir, "exotic_rays.npz")
    rays = synth.rays_exotic(2048, tris=scene1)
    assert sha(scene1) == str(g["scene_sha256"]) and sha(rays) == str(g["rays_sha256"])
    blobs = oracle.leaf_chain_blobs(scene1.reshape(-1, 3, 3))
    hits, mask = _chain(host_lib, blobs, rays)
    gm = g["hit_mask"].astype(bool)
    assert (mask == gm).all() and gm.sum() > 300
    assert (hits["triangle_index"][mask] == g["hit_tri"][gm]).all()
    for f, k in (("t", "hit_t"), ("u", "hit_u"), ("v", "hit_v")):
        assert (hits[f][mask].view(np.uint32) == g[k][gm].view(np.uint32)).all(), f


def test_config1_full_on_a_bvh4_blob(host_lib, oracle, golden_dir):
    """All 65 536 rays of config 1 on the oracle's SAH blob: the fixture's ids (real rtk.c), and every byte of (mesh, triangle, t,
    u, v, the three vertices) the oracle returns on the same blob."""
    scene1 = synth.scene_for_config(1)
    g = load_golden(golden_dir, "cfg1_full.npz")
    rays = synth.rays_config1(65536)
    blob = oracle.build_scene([dict(positions=scene1)])
    hits, mask = _per_ray(host_lib, blob.ptr, rays)
    compare_hits_struct(hits, mask, g, "cfg1/host per-ray")
    oh, om = oracle.trace(blob, rays)
    assert (mask == om).all()
    assert hits[mask].tobytes() == oh[om].tobytes()


def test_the_products_own_cpu_built_blob(host_lib, oracle):
    """rtk_build_scene with the task-graph CPU builder (no GPU anywhere), several meshes with mixed index types, then rtk_trace_ray:
    equal to the oracle on the same blob, miss leaves *hit untouched."""
    from tests.util import random_mixed_scene
    L = host_lib.lib()
    desc, keep, tris, mesh_of, tri_of, vidx_of = random_mixed_scene(5)
    L.rtk_amd_set_builder(1)
    try:
        p = L.rtk_build_scene(C.byref(desc))
    finally:
        L.rtk_amd_set_builder(0)
    assert p, host_lib.last_error()
    try:
        rays = synth.rays_config1(8192, seed=11)
        hits, mask = _per_ray(host_lib, p, rays)
        blob = oracle.Blob(host_lib.scene_bytes(p))
        oh, om = oracle.trace(blob, rays)
        assert (mask == om).all() and mask.sum() > 500
        assert hits[mask].tobytes() == oh[om].tobytes()
        assert not hits[~mask].tobytes().strip(b"\0")                     # untouched on a miss (rtk.c:571-576)
    finally:
        L.rtk_free_scene(C.c_void_p(p))


def test_filter_offers_candidates_in_order_until_one_is_accepted(host_lib, oracle):
    """rtk_trace_ray_filter on the host: candidates in increasing (t, mesh, triangle) order, equal-t ones all offered, the first
    accepted one returned; the oracle's filtered trace gives the same hit."""
    L = host_lib.lib()
    # five parallel sheets, the second and third exact duplicates (equal t, different ids)
    sheet = np.array([[-1, -1, 0], [3, -1, 0], [-1, 3, 0]], np.float32)
    zs = [1.0, 2.0, 2.0, 3.0, 4.0]
    tris = np.stack([sheet + np.array([0, 0, z], np.float32) for z in zs])
    blob = oracle.build_scene([dict(positions=tris.reshape(-1, 3))])
    ray = np.zeros(1, RAY_DTYPE)
    ray["origin"] = (0.2, 0.2, 0.0); ray["direction"] = (0, 0, 1); ray["min_t"] = 0.0; ray["max_t"] = 100.0
    FILTER = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_void_p, C.c_void_p)
    for accept_from in range(6):
        seen = []

        def cb(user, ray_ptr, hit_ptr):
            h = np.ctypeslib.as_array((C.c_uint8 * 68).from_address(hit_ptr)).view(HIT_DTYPE)[0]
            seen.append((float(h["t"]), int(h["triangle_index"])))
            return len(seen) > accept_from
        fn = FILTER(cb)
        hit = np.zeros(1, HIT_DTYPE)
        L.rtk_trace_ray_filter.restype = C.c_bool
        ok = L.rtk_trace_ray_filter(C.c_void_p(blob.ptr), C.c_void_p(ray.ctypes.data), C.c_void_p(hit.ctypes.data), C.cast(fn, C.c_void_p), None)
        want = [(1.0, 0), (2.0, 1), (2.0, 2), (3.0, 3), (4.0, 4)]
        assert seen == want[:min(accept_from + 1, 5)]
        assert ok == (accept_from < 5)
        if ok:
            assert (float(hit["t"][0]), int(hit["triangle_index"][0])) == want[accept_from]


def test_a_blob_that_is_not_a_tree_is_a_reported_failure_not_a_hang(host_lib, oracle):
    """A node that points at itself: the walk ends (step cap / stack bound), the call reports a failure on stderr and -- as the
    host asked for soft errors -- returns false with the error text set."""
    import os
    import subprocess
    import sys
    snippet = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r)
from rtk_amd import api, synth
from rtk_amd.types import HIT_DTYPE, RAY_DTYPE
from oracle import pyoracle
blob = pyoracle.build_scene([dict(positions=synth.triangle_soup(64, 0.3, seed=2))])
data = blob.data
np.frombuffer(data, np.uint64, 4, 128 + 96)[:] = 128       # every child of the root is the root
L = api.lib()
L.rtk_trace_ray.restype = C.c_bool
ray = np.zeros(1, RAY_DTYPE); ray["origin"] = (0.5, 0.5, -1); ray["direction"] = (0, 0, 1); ray["max_t"] = 100.0
hit = np.zeros(1, HIT_DTYPE)
ok = L.rtk_trace_ray(C.c_void_p(data.ctypes.data), C.c_void_p(ray.ctypes.data), C.c_void_p(hit.ctypes.data))
print("returned", ok, api.last_error(), flush=True)
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RTK_AMD_SOFT_ERRORS="1")
    r = subprocess.run([sys.executable, "-c", snippet % root], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "returned False" in r.stdout and ("not a tree" in r.stdout or "exhausted" in r.stdout or "deeper" in r.stdout)
    assert "FAILED, not a miss" in r.stderr
