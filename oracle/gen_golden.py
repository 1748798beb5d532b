#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/librtk_ref.so).

TEST INFRASTRUCTURE. Runs only in the build container, where /root/reference exists:
the reference's own rtk_trace_ray (rtk.c:543-577, compiled verbatim by oracle/Makefile)
is driven through the "leaf chain" of SURVEY.md section 8c -- the triangle list is cut
into chunks of 60, each chunk becomes a minimal single-leaf blob, and every ray visits
every blob in order with ray.max_t = best.t. All hit decisions and all t/u/v values in
the fixtures therefore come from the reference's code, not from the restatement.

Fixtures hold data only: seeds/inputs (or their SHA-256) and expected hit records.

    python3 oracle/gen_golden.py [--only cfg1,edge,...] [--cfg5-rays N]
"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402
from rtk_amd import synth  # noqa: E402
from rtk_amd.types import RAY_DTYPE, RTK_INF  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def compact(hits, mask):
    return dict(hit_mask=mask.astype(np.uint8),
                hit_mesh=np.where(mask, hits["mesh_index"], 0xFFFFFFFF).astype(np.uint32),
                hit_tri=np.where(mask, hits["triangle_index"], 0xFFFFFFFF).astype(np.uint32),
                hit_t=np.where(mask, hits["t"], 0).astype(np.float32),
                hit_u=np.where(mask, hits["u"], 0).astype(np.float32),
                hit_v=np.where(mask, hits["v"], 0).astype(np.float32))


def save(name, **kw):
    path = os.path.join(GOLDEN, name)
    np.savez_compressed(path, **kw)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


def chain_reference(tris, rays, mesh_index=None, triangle_index=None):
    blobs = po.leaf_chain_blobs(tris.reshape(-1, 3, 3), mesh_index, triangle_index)
    t0 = time.time()
    hits, mask = po.ref_trace_chain(blobs, rays)
    print("  reference leaf chain: %d rays x %d blobs in %.1f s, %d hits"
          % (len(rays), len(blobs), time.time() - t0, int(mask.sum())))
    return hits, mask


def gen_cfg1():
    tris = synth.scene_for_config(1)
    rays = synth.rays_config1(65536)
    hits, mask = chain_reference(tris, rays)
    save("cfg1_full.npz", scene_sha256=sha(tris), rays_sha256=sha(rays), **compact(hits, mask))


def gen_exotic():
    """2048 rays made of special values (synth.rays_exotic) through the REAL rtk_trace_ray over the config-1 leaf chain:
    pins the oracle's ray set-up and leaf arithmetic where inputs are zero, denormal, huge, NaN or tied."""
    tris = synth.scene_for_config(1)
    rays = synth.rays_exotic(2048, tris=tris)
    hits, mask = chain_reference(tris, rays)
    save("exotic_rays.npz", scene_sha256=sha(tris), rays_sha256=sha(rays), **compact(hits, mask))


def sample_indices(total, count, mult=4099):
    return ((np.arange(count, dtype=np.int64) * mult) % total).astype(np.int64)


def gen_cfg2(tris):
    idx = sample_indices(4096 * 4096, 4096)
    allr = np.concatenate([synth.rays_pinhole(first=int(i), count=1) for i in idx])
    hits, mask = chain_reference(tris, allr)
    save("cfg2_sample.npz", scene_sha256=sha(tris), ray_index=idx, rays_sha256=sha(allr), **compact(hits, mask))


def gen_cfg3(tris):
    rays = synth.rays_incoherent(4096)
    hits, mask = chain_reference(tris, rays)
    save("cfg3_sample.npz", scene_sha256=sha(tris), rays_sha256=sha(rays), **compact(hits, mask))


def gen_cfg5(nrays):
    tris = synth.scene_for_config(5)
    rays = synth.rays_shadow(nrays)
    hits, mask = chain_reference(tris, rays)
    # config 5 is any-hit: the golden value is the boolean; the closest hit is kept too
    save("cfg5_sample.npz", scene_sha256=sha(tris), rays_sha256=sha(rays), **compact(hits, mask))


def edge_scene():
    """Small hand-made scene; every coordinate is exactly representable."""
    T = []
    M = []
    def tri(a, b, c, mesh=0):
        T.append([a, b, c]); M.append(mesh)
    tri((0, 0, 1), (1, 0, 1), (0, 1, 1))                 # 0: axis-aligned, z = 1
    tri((1, 0, 1), (1, 1, 1), (0, 1, 1))                 # 1: shares the diagonal with 0
    tri((0, 0, 2), (1, 0, 2), (0, 1, 2))                 # 2: behind 0
    tri((.25, .25, .5), (.25, .25, .5), (.375, .375, .5))  # 3: zero area, in front of 0
    tri((2, 0, 0), (3, 0, 1), (2, 1, 2))                 # 4: slanted
    tri((4, 0, 1), (4, 1, 1), (5, 0, 1))                 # 5: opposite winding
    tri((6, 0, 1), (7, 0, 1), (6, 1, 1))                 # 6: next to 7 along x
    tri((7, 0, 1), (8, 0, 1), (7, 1, 1))                 # 7: touches 6 in one vertex
    tri((0, 0, 1), (1, 0, 1), (0, 1, 1), mesh=1)         # 8: exact duplicate of 0 in mesh 1
    tri((0, 0, 1), (1, 0, 1), (0, 1, 1), mesh=1)         # 9: and again
    tris = np.array(T, dtype=np.float32)
    mesh = np.array(M, dtype=np.uint32)
    # triangle_index counts inside each mesh (reference rtk.c:1168-1169)
    tri_index = np.zeros(len(T), np.uint32)
    for m in np.unique(mesh):
        sel = mesh == m
        tri_index[sel] = np.arange(sel.sum(), dtype=np.uint32)
    return tris, mesh, tri_index


def edge_rays():
    R = []
    def ray(o, d, tmin=0.0, tmax=float(RTK_INF)):
        R.append((o, d, tmin, tmax))
    nz = -0.0
    ray((.25, .25, 0), (0, 0, 1))                 # interior, two zero direction components
    ray((.5, .5, 0), (0, 0, 1))                   # exactly on the shared diagonal (f64 fallback, tie 0/1)
    ray((1, 0, 0), (0, 0, 1))                     # through a vertex shared by 0 and 1
    ray((0, 0, 0), (0, 0, 1))                     # through the corner vertex of 0
    ray((0, .5, 0), (0, 0, 1))                    # on an outer edge of 0
    ray((1.0000001, 0, 0), (0, 0, 1))             # just outside
    ray((-1e-7, .5, 0), (0, 0, 1))                # just outside the outer edge
    ray((.25, .25, 0), (nz, 0.0, 1))              # negative zero in the direction
    ray((.25, .25, 0), (nz, nz, 1))
    ray((.25, .25, 1.5), (0, 0, 1))               # starts between 0 and 2
    ray((.25, .25, 0), (0, 0, 1), tmin=1.0)       # t == min_t is rejected (open interval)
    ray((.25, .25, 0), (0, 0, 1), tmax=1.0)       # t == max_t is rejected
    ray((.25, .25, 0), (0, 0, 1), tmax=2.0)
    ray((.25, .25, 0), (0, 0, 1), tmin=0.999999, tmax=1.000001)
    ray((.25, .25, 0), (0, 0, 2))                 # unnormalised direction: t = 0.5
    ray((.25, .25, 0), (0, 0, .5))
    ray((.25, .25, 3), (0, 0, -1))                # from behind: hits 2 first
    ray((.3, .3, 0), (0, 0, 1))                   # passes the zero-area triangle 3
    ray((.25, .25, 0), (.01, .02, 1))
    ray((-1, .25, 1), (1, 0, 0))                  # in the plane of 0: det == 0
    ray((2.25, .25, -1), (0, 0, 1))               # slanted 4
    ray((2.25, .25, 3), (0, 0, -1))               # slanted 4 from the other side
    ray((1, -1, -1), (1, 1, 1))                   # |dx|=|dy|=|dz|: kz = x
    ray((2.5, -1, -1), (0, 1, 1))                 # |dy|=|dz|: kz = y
    ray((2.5, -1, -1), (0, -1, -1))
    ray((4.25, .25, 0), (0, 0, 1))                # opposite winding, front
    ray((4.25, .25, 2), (0, 0, -1))               # opposite winding, back
    ray((7, 0, 0), (0, 0, 1))                     # vertex shared by 6 and 7
    ray((7, .5, 0), (0, 0, 1))                    # edge of 7 only
    ray((6.5, .5, 0), (0, 0, 1))                  # hypotenuse of 6
    ray((.25, .25, 0), (0, 0, 1e-30))             # tiny direction
    ray((.25, .25, 0), (0, 0, 1e30))              # huge direction
    ray((.25, .25, -1e6), (0, 0, 1))              # far origin
    ray((100, 100, 0), (0, 0, 1))                 # misses everything
    ray((.25, .25, 0), (1, 0, 0))                 # parallel, misses
    out = np.zeros(len(R), dtype=RAY_DTYPE)
    for i, (o, d, a, b) in enumerate(R):
        out[i]["origin"] = o; out[i]["direction"] = d; out[i]["min_t"] = a; out[i]["max_t"] = b
    # plus deterministic pseudo-random rays over the same scene
    n = 512
    u = synth.u01(77, 0, n * 6).reshape(n, 6)
    rnd = np.zeros(n, dtype=RAY_DTYPE)
    rnd["origin"][:, 0] = u[:, 0] * np.float32(9.0) - np.float32(0.5)
    rnd["origin"][:, 1] = u[:, 1] * np.float32(2.0) - np.float32(0.5)
    rnd["origin"][:, 2] = np.float32(-1.0)
    rnd["direction"][:, 0] = (u[:, 3] - np.float32(0.5))
    rnd["direction"][:, 1] = (u[:, 4] - np.float32(0.5))
    rnd["direction"][:, 2] = np.float32(1.0)
    rnd["min_t"] = 0
    rnd["max_t"] = RTK_INF
    # a grid of axis-parallel rays landing exactly on lattice points (edges/vertices/diagonal)
    g = []
    for ix in range(0, 17):
        for iy in range(0, 17):
            g.append(((ix / 16.0, iy / 16.0, 0.0), (0, 0, 1), 0.0, float(RTK_INF)))
    grid = np.zeros(len(g), dtype=RAY_DTYPE)
    for i, (o, d, a, b) in enumerate(g):
        grid[i]["origin"] = o; grid[i]["direction"] = d; grid[i]["min_t"] = a; grid[i]["max_t"] = b
    return np.concatenate([out, rnd, grid])


def gen_edge():
    tris, mesh, tri_index = edge_scene()
    rays = edge_rays()
    blobs = po.leaf_chain_blobs(tris, mesh, tri_index)
    assert len(blobs) == 1
    hits, mask = po.ref_trace_chain(blobs, rays, threads=1)
    print("  edge cases: %d rays, %d hits" % (len(rays), int(mask.sum())))
    save("edge_cases.npz", tris=tris, mesh=mesh, tri_index=tri_index, rays=rays.view(np.float32).reshape(-1, 8),
         **compact(hits, mask))


def _f32_ulps(a, b):
    """Distance in units in the last place between positive float32 arrays."""
    return np.abs(a.astype(np.float32).view(np.int32).astype(np.int64) - b.astype(np.float32).view(np.int32).astype(np.int64))


def find_near_ties(blob, make_rays, total, ulps, chunk=1 << 21, max_cand=4):
    """Rays of a batch whose two closest candidates lie within `ulps` float32 ulps of each other in t.
    Returns (ray_index[k], cand_prim[k, max_cand]) with 0xffffffff padding. Uses the CPU restatement's own
    SAH tree: the candidates' t values can move by an ulp with the leaf grouping (rtk.c:302-336), so the
    threshold is several ulps wide."""
    out_idx, out_cand = [], []
    for a in range(0, total, chunk):
        n = min(chunk, total - a)
        rays = make_rays(a, n)
        h1, m1 = po.trace(blob, rays)
        t1 = np.where(m1, h1["t"], 0).astype(np.float32)
        mesh1 = np.where(m1, h1["mesh_index"], 0xFFFFFFFF).astype(np.uint32)
        h2, m2 = po.trace_filtered(blob, rays, after=(t1, mesh1, h1["triangle_index"]))
        both = m1 & m2
        near = both & (_f32_ulps(np.where(both, h2["t"], 1), np.where(both, h1["t"], 1)) <= ulps)
        sel = np.nonzero(near)[0]
        if sel.size == 0:
            continue
        cand = np.full((sel.size, max_cand), 0xFFFFFFFF, np.uint32)
        cand[:, 0] = h1["triangle_index"][sel]
        cand[:, 1] = h2["triangle_index"][sel]
        cur_t, cur_tri, sub, alive = h2["t"][sel].copy(), h2["triangle_index"][sel].copy(), rays[sel], np.ones(sel.size, bool)
        for k in range(2, max_cand):
            hk, mk = po.trace_filtered(blob, sub, after=(cur_t, np.where(alive, 0, 0xFFFFFFFF).astype(np.uint32), cur_tri))
            alive = alive & mk & (_f32_ulps(np.where(mk, hk["t"], 1), h1["t"][sel]) <= ulps)
            cand[alive, k] = hk["triangle_index"][alive]
            cur_t = np.where(alive, hk["t"], cur_t).astype(np.float32)
            cur_tri = np.where(alive, hk["triangle_index"], cur_tri).astype(np.uint32)
        out_idx.append(sel.astype(np.int64) + a)
        out_cand.append(cand)
        print("    rays %d..%d: %d near ties" % (a, a + n, sel.size), flush=True)
    if not out_idx:
        return np.zeros(0, np.int64), np.zeros((0, max_cand), np.uint32)
    return np.concatenate(out_idx), np.concatenate(out_cand)


def reference_values_both_groupings(tris, rays, cand):
    """For every (ray, candidate triangle X): what the REAL rtk_trace_ray returns for X under the two groupings
    a leaf can put it in -- alone with three padding slots (rtk.c:306: the whole group takes the double-precision
    edge functions) and in a full group of four (here four copies of X: the float path unless X itself has an
    exactly-zero edge function). Returns hit[k, c, 2] and tuv[k, c, 2, 3]; index 0 = padded group, 1 = full group."""
    k, mc = cand.shape
    hit = np.zeros((k, mc, 2), np.uint8)
    tuv = np.zeros((k, mc, 2, 3), np.float32)
    tv = tris.reshape(-1, 3, 3)
    for i in range(k):
        for c in range(mc):
            p = int(cand[i, c])
            if p == 0xFFFFFFFF:
                continue
            for g, reps in enumerate((1, 4)):
                blobs = po.leaf_chain_blobs(np.repeat(tv[p:p + 1], reps, axis=0), triangle_index=np.full(reps, p, np.uint32))
                h, m = po.ref_trace_chain(blobs, rays[i:i + 1], threads=1)
                hit[i, c, g] = m[0]
                if m[0]:
                    assert h["triangle_index"][0] == p
                    tuv[i, c, g] = (h["t"][0], h["u"][0], h["v"][0])
    return hit, tuv


def gen_near_ties(tris, ulps=8):
    """tests/golden/near_ties.npz: every ray of the FULL config-2 and config-3 batches (2^24 rays each) whose two
    closest candidates are within `ulps` ulps in t, with the reference's own values for each candidate under both
    leaf groupings. Which of two such candidates rtk.c reports depends on its leaf grouping;
    the GPU test checks that whatever the device BVH reports for these rays is one of the reference's answers."""
    t0 = time.time()
    blob = po.build_scene([dict(positions=tris)])
    print("  oracle SAH scene built in %.1f s" % (time.time() - t0))
    total = 4096 * 4096
    out = {}
    for name, make in (("cfg2", lambda a, n: synth.rays_pinhole(4096, 4096, first=a, count=n)),
                       ("cfg3", lambda a, n: synth.rays_incoherent(n, first=a))):
        t0 = time.time()
        idx, cand = find_near_ties(blob, make, total, ulps)
        rays = np.concatenate([make(int(i), 1) for i in idx]) if len(idx) else np.zeros(0, RAY_DTYPE)
        hit, tuv = reference_values_both_groupings(tris, rays, cand)
        print("  %s: %d near-tie rays of %d (%.0f s)" % (name, len(idx), total, time.time() - t0))
        out[name + "_ray_index"] = idx
        out[name + "_rays"] = rays.view(np.float32).reshape(-1, 8)
        out[name + "_cand_prim"] = cand
        out[name + "_cand_hit"] = hit
        out[name + "_cand_tuv"] = tuv
    save("near_ties.npz", scene_sha256=sha(tris), ulps=np.int64(ulps), **out)


def gen_tiny_t(tris, thresh=1e-6):
    """tests/golden/tiny_t.npz: every ray of the FULL config-2 and config-3 batches whose closest hit lies at |t| < 1e-6
    (a ray origin on, or a hair off, a triangle). There a relative tolerance on t means nothing -- the value is what is
    left of a cancellation, and rtk.c's group-of-four rule (rtk.c:302-336) moves it by tens of percent with the leaf
    grouping (round 2's bench line: max_rel_t 0.25 at t = 2.9e-8 between two BVHs) -- so the pin is exact: the REAL
    rtk.c's (t, u, v) for the hit triangle, and for the next candidates behind it, under both groupings."""
    blob = po.build_scene([dict(positions=tris)])
    total = 4096 * 4096
    out = {}
    for name, make in (("cfg2", lambda a, n: synth.rays_pinhole(4096, 4096, first=a, count=n)),
                       ("cfg3", lambda a, n: synth.rays_incoherent(n, first=a))):
        t0 = time.time()
        idxs, cands = [], []
        chunk = 1 << 21
        for a in range(0, total, chunk):
            n = min(chunk, total - a)
            rays = make(a, n)
            h1, m1 = po.trace(blob, rays)
            sel = np.nonzero(m1 & (np.abs(h1["t"]) < thresh))[0]
            if sel.size == 0:
                continue
            cand = np.full((sel.size, 4), 0xFFFFFFFF, np.uint32)
            cand[:, 0] = h1["triangle_index"][sel]
            # the candidates right behind it (within the same threshold), as find_near_ties does
            cur_t, cur_tri, sub, alive = h1["t"][sel].copy(), h1["triangle_index"][sel].copy(), rays[sel], np.ones(sel.size, bool)
            for k in range(1, 4):
                hk, mk = po.trace_filtered(blob, sub, after=(cur_t, np.where(alive, 0, 0xFFFFFFFF).astype(np.uint32), cur_tri))
                alive = alive & mk & (np.abs(np.where(mk, hk["t"], 1)) < thresh)
                cand[alive, k] = hk["triangle_index"][alive]
                cur_t = np.where(alive, hk["t"], cur_t).astype(np.float32)
                cur_tri = np.where(alive, hk["triangle_index"], cur_tri).astype(np.uint32)
            idxs.append(sel.astype(np.int64) + a)
            cands.append(cand)
        idx = np.concatenate(idxs) if idxs else np.zeros(0, np.int64)
        cand = np.concatenate(cands) if cands else np.zeros((0, 4), np.uint32)
        rays = np.concatenate([make(int(i), 1) for i in idx]) if len(idx) else np.zeros(0, RAY_DTYPE)
        hit, tuv = reference_values_both_groupings(tris, rays, cand)
        print("  %s: %d rays with |t| < %g of %d (%.0f s)" % (name, len(idx), thresh, total, time.time() - t0))
        out[name + "_ray_index"] = idx
        out[name + "_rays"] = rays.view(np.float32).reshape(-1, 8)
        out[name + "_cand_prim"] = cand
        out[name + "_cand_hit"] = hit
        out[name + "_cand_tuv"] = tuv
    save("tiny_t.npz", scene_sha256=sha(tris), thresh=np.float64(thresh), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="edge,cfg1,cfg2,cfg3,cfg5")
    ap.add_argument("--cfg5-rays", type=int, default=1024)
    a = ap.parse_args()
    if not po.have_ref():
        po.build_oracle()
    if not po.have_ref():
        sys.exit("oracle/_ref/librtk_ref.so missing: this script needs the reference (build container only)")
    os.makedirs(GOLDEN, exist_ok=True)
    only = set(a.only.split(","))
    if "edge" in only:
        gen_edge()
    if "cfg1" in only:
        gen_cfg1()
    if "exotic" in only:
        gen_exotic()
    if only & {"cfg2", "cfg3", "near_ties", "tiny_t"}:
        tris = synth.scene_for_config(2)
        if "near_ties" in only:
            gen_near_ties(tris)
        if "tiny_t" in only:
            gen_tiny_t(tris)
        if "cfg2" in only:
            gen_cfg2(tris)
        if "cfg3" in only:
            gen_cfg3(tris)
    if "cfg5" in only:
        gen_cfg5(a.cfg5_rays)


if __name__ == "__main__":
    main()
