"""CPU tests: the oracle (CPU restatement) against the golden fixtures that were produced by
the REAL reference (oracle/gen_golden.py, leaf-chain through rtk.c's own rtk_trace_ray).

No GPU needed. These pin the checker itself: generators (input checksums), triangle
arithmetic (bit-exact on identical blobs), and the oracle's builder + traversal (ids exact,
t/u/v within tolerance on its own BVH).
"""
import numpy as np
import pytest

from rtk_amd import synth
from rtk_amd.types import RAY_DTYPE
from tests.util import compare_hits_struct, load_golden, sha


@pytest.fixture(scope="module")
def scene1():
    return synth.scene_for_config(1)


@pytest.fixture(scope="module")
def scene2():
    return synth.scene_for_config(2)


def test_generators_are_pinned(golden_dir, scene1):
    g = load_golden(golden_dir, "cfg1_full.npz")
    assert sha(scene1) == str(g["scene_sha256"])
    assert sha(synth.rays_config1(65536)) == str(g["rays_sha256"])
    g3 = load_golden(golden_dir, "cfg3_sample.npz")
    assert sha(synth.rays_incoherent(4096)) == str(g3["rays_sha256"])


def test_edge_cases_bit_exact_on_same_blob(oracle, golden_dir):
    """Same single-leaf blob as the reference saw -> every byte of (t,u,v,ids) must match."""
    g = load_golden(golden_dir, "edge_cases.npz")
    rays = np.ascontiguousarray(g["rays"]).view(RAY_DTYPE).reshape(-1)
    blobs = oracle.leaf_chain_blobs(g["tris"], g["mesh"], g["tri_index"])
    hits, mask = oracle.trace_chain(blobs, rays, threads=1)
    st = compare_hits_struct(hits, mask, g, "edge/chain")
    assert st["bit_exact"] == 1.0
    # the constructed cases really exercise what they claim
    assert mask[0] and hits["t"][0] == 1.0 and hits["triangle_index"][0] == 0 and hits["mesh_index"][0] == 0
    assert mask[1] and hits["triangle_index"][1] == 0          # diagonal tie resolved to the lower id
    assert mask[10] and hits["t"][10] == 2.0                   # t == min_t rejected, next surface found
    assert not mask[11]                                        # t == max_t rejected
    assert mask[19] and hits["triangle_index"][19] == 4        # in the plane of 0/1 (det == 0 there), lands on 4


@pytest.mark.parametrize("ties", ["canonical", "reference"])
def test_edge_cases_on_oracle_bvh(oracle, golden_dir, ties):
    """Oracle-built BVH over the edge scene: ids exact (canonical ties), values within tolerance."""
    g = load_golden(golden_dir, "edge_cases.npz")
    rays = np.ascontiguousarray(g["rays"]).view(RAY_DTYPE).reshape(-1)
    tris = g["tris"]
    meshes = []
    for m in np.unique(g["mesh"]):
        meshes.append(dict(positions=tris[g["mesh"] == m].reshape(-1, 3)))
    blob = oracle.build_scene(meshes)
    rc, counts = oracle.validate_blob(blob)
    assert rc == 0 and counts["tris"] == len(tris)
    mode = oracle.TIES_CANONICAL if ties == "canonical" else oracle.TIES_REFERENCE
    hits, mask = oracle.trace(blob, rays, ties=mode, threads=1)
    if ties == "canonical":
        compare_hits_struct(hits, mask, g, "edge/bvh")
    else:
        # first-encountered semantics may pick another of the exact duplicates, never another t
        assert (mask == g["hit_mask"].astype(bool)).all()
        assert np.allclose(hits["t"][mask], g["hit_t"][mask], rtol=1e-5, atol=0)


def test_config1_full_on_oracle_bvh(oracle, golden_dir, scene1):
    g = load_golden(golden_dir, "cfg1_full.npz")
    rays = synth.rays_config1(65536)
    blob = oracle.build_scene([dict(positions=scene1)])
    rc, counts = oracle.validate_blob(blob)
    assert rc == 0 and counts["tris"] == 10000
    hits, mask, ctr = oracle.trace(blob, rays, counters=True)
    st = compare_hits_struct(hits, mask, g, "cfg1/bvh")
    assert st["hits"] == int(g["hit_mask"].sum()) == ctr["hits"]
    assert st["max_rel_t"] < 2e-6


def test_config1_chain_bit_exact_subset(oracle, golden_dir, scene1):
    g = load_golden(golden_dir, "cfg1_full.npz")
    sel = np.arange(0, 65536, 16)
    rays = synth.rays_config1(65536)[sel]
    blobs = oracle.leaf_chain_blobs(scene1.reshape(-1, 3, 3))
    hits, mask = oracle.trace_chain(blobs, rays)
    gs = {k: g[k][sel] for k in ("hit_mask", "hit_mesh", "hit_tri", "hit_t", "hit_u", "hit_v")}
    st = compare_hits_struct(hits, mask, gs, "cfg1/chain")
    assert st["bit_exact"] == 1.0


@pytest.mark.slow
def test_config2_and_3_samples_on_oracle_bvh(oracle, golden_dir, scene2):
    g2 = load_golden(golden_dir, "cfg2_sample.npz")
    g3 = load_golden(golden_dir, "cfg3_sample.npz")
    assert sha(scene2) == str(g2["scene_sha256"])
    blob = oracle.build_scene([dict(positions=scene2)])
    rc, counts = oracle.validate_blob(blob)
    assert rc == 0 and counts["tris"] == 1_000_000
    r2 = np.concatenate([synth.rays_pinhole(first=int(i), count=1) for i in g2["ray_index"]])
    assert sha(r2) == str(g2["rays_sha256"])
    hits, mask = oracle.trace(blob, r2)
    compare_hits_struct(hits, mask, g2, "cfg2/bvh")
    hits, mask = oracle.trace(blob, synth.rays_incoherent(4096))
    compare_hits_struct(hits, mask, g3, "cfg3/bvh")


def test_ray_setup_vectors(oracle):
    """kz = first axis with |d| == max (x, y, z order); sign mask from sign bits (rtk.c:550-556)."""
    def ray(d):
        r = np.zeros(1, RAY_DTYPE)
        r["direction"] = d
        return r
    k, sh, sm = oracle.ray_setup(ray((1, 1, 1)))
    assert list(k) == [1, 2, 0] and sm == 0 and list(sh) == [-1.0, -1.0, 1.0]
    k, sh, sm = oracle.ray_setup(ray((0, -1, 1)))
    assert list(k) == [2, 0, 1] and sm == 2
    k, sh, sm = oracle.ray_setup(ray((-0.0, 0.25, -2)))
    assert list(k) == [0, 1, 2] and sm == 5 and sh[2] == -0.5 and sh[1] == 0.125


def test_exotic_rays_chain_bit_exact(oracle, golden_dir, scene1):
    """2048 rays made of zeros, negative zeros, denormals, huge and tied direction components, far / on-vertex origins,
    empty, reversed and NaN-min intervals (synth.rays_exotic): same leaf chain as the REAL rtk.c saw -> identical hit/miss,
    ids and t/u/v bits. Pins the oracle's ray set-up (rtk.c:550-566) and leaf arithmetic where it is most fragile."""
    g = load_golden(golden_dir, "exotic_rays.npz")
    rays = synth.rays_exotic(2048, tris=scene1)
    assert sha(scene1) == str(g["scene_sha256"]) and sha(rays) == str(g["rays_sha256"])
    blobs = oracle.leaf_chain_blobs(scene1.reshape(-1, 3, 3))
    hits, mask = oracle.trace_chain(blobs, rays)
    gm = g["hit_mask"].astype(bool)
    assert (mask == gm).all()
    assert gm.sum() > 300
    assert (hits["triangle_index"][mask] == g["hit_tri"][gm]).all()
    for f, k in (("t", "hit_t"), ("u", "hit_u"), ("v", "hit_v")):
        assert (hits[f][mask].view(np.uint32) == g[k][gm].view(np.uint32)).all(), f
