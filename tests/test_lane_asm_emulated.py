"""The hand-written per-lane kernels (rtk_amd/csrc/rtk_lane_hot.S) run on the CPU, in tests/gfx950_emu.py, against the oracle.

No GPU is involved: the object file the Makefile assembles is disassembled and executed instruction by instruction on
small scenes, every memory and LDS access range-checked, every wave under an instruction budget. Oracle: the C restatement
of rtk.c over a chain of ONE-triangle leaf blobs -- every triangle alone in a padded group, i.e. the double-precision
edge functions (rtk.c:306) the kernels use for their leaves of fewer than four triangles -- so hit / miss, primitive and
t, u, v must agree bit for bit.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from rtk_amd import synth
from rtk_amd.types import HIT_RECORD_DTYPE, RAY_DTYPE

from . import gfx950_emu as emu

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "rtk_amd", "csrc")
OBJ = os.path.join(CSRC, "obj", "rtk_lane_hot.o")

NODEQ = np.dtype([("org", "<f4", (3,)), ("scale", "<f4", (3,)), ("q", "<u4", (3, 2)), ("child", "<u4", (4,))])
NODE = np.dtype([("bx", "<f4", (2, 4)), ("by", "<f4", (2, 4)), ("bz", "<f4", (2, 4)), ("child", "<u4", (4,)), ("order", "<u4", (4,))])
assert NODE.itemsize == 128
TRI = np.dtype([("v0", "<f4", (3,)), ("prim", "<u4"), ("v1", "<f4", (3,)), ("flags", "<u4"), ("v2", "<f4", (3,)), ("count", "<u4")])
assert NODEQ.itemsize == 64 and TRI.itemsize == 48
NONE = 0xFFFFFFFF
COUNTER_WORDS = 16 + 16 * 8 + 1
LEFTOVER_WORD = 12


@pytest.fixture(scope="module")
def lane_obj():
    subprocess.check_call(["make", "-s", "-C", CSRC, os.path.join(os.path.abspath(CSRC), "obj", "rtk_lane_hot.hsaco")])
    return OBJ


def test_no_wait_state_findings(lane_obj):
    """scripts/asm_hazards.py: the gfx940-family hazards an assembler does not pad for."""
    subprocess.check_call(["make", "-s", "-C", CSRC, "lint"])


def grid_step(extent):
    """rtk_node_finish.h grid_step"""
    if not extent > 0:
        return np.float32(1.17549435e-38)
    m, e = np.frexp(np.float32(extent))
    s = np.float32(np.ldexp(1.0, int(e) - 8))
    if np.float32(254.0) * s < np.float32(extent):
        s = np.float32(s * 2)
    return s


def child_order(nd):
    """rtk_node_finish.h child_order: per direction octant the front-to-back order of the children (a permutation and six
    pair bits) by the centre of the child box along the octant's diagonal."""
    order = [0, 0, 0, 0]
    f = np.float32
    c = [[f(nd[ax][0][k]) + f(nd[ax][1][k]) for k in range(4)] for ax in ("bx", "by", "bz")]
    for o in range(8):
        key = []
        for k in range(4):
            sv = f((-c[0][k] if o & 1 else c[0][k]) + (-c[1][k] if o & 2 else c[1][k]))
            sv = f(sv + (-c[2][k] if o & 4 else c[2][k]))
            key.append(np.inf if int(nd["child"][k]) == NONE else (np.inf if np.isnan(sv) else float(sv)))
        b = {(i, j): int(key[i] <= key[j]) for i in range(4) for j in range(i + 1, 4)}
        r = [(1 - b[0, 1]) + (1 - b[0, 2]) + (1 - b[0, 3]), b[0, 1] + (1 - b[1, 2]) + (1 - b[1, 3]), b[0, 2] + b[1, 2] + (1 - b[2, 3]), b[0, 3] + b[1, 3] + b[2, 3]]
        word = (0 << (2 * r[0])) | (1 << (2 * r[1])) | (2 << (2 * r[2])) | (3 << (2 * r[3]))
        pair = (1 - b[0, 1]) | ((1 - b[0, 2]) << 1) | ((1 - b[0, 3]) << 2) | ((1 - b[1, 2]) << 3) | ((1 - b[1, 3]) << 4) | ((1 - b[2, 3]) << 5)
        word |= pair << 8
        order[o >> 1] |= word << (16 * (o & 1))
    return order


def build_bvh4(tv, leaf_max=3, seed=0, want_exact=False):
    """A 4-wide tree over triangles tv [n,3,3] (median splits on the longest axis, two binary levels per node) in the
    device layout: 64-byte compressed nodes whose 8-bit boxes CONTAIN the exact ones (rtk_node_finish.h quantize_node),
    48-byte triangle records, leaves = runs of records."""
    n = tv.shape[0]
    lo_t, hi_t = tv.min(axis=1), tv.max(axis=1)
    cen = (lo_t + hi_t) * 0.5
    nodes, tris = [], []

    def split(ids):
        ext = cen[ids].max(axis=0) - cen[ids].min(axis=0)
        ax = int(np.argmax(ext))
        order = ids[np.argsort(cen[ids, ax], kind="stable")]
        h = len(order) // 2
        return order[:h], order[h:]

    def emit_leaf(ids):
        first = len(tris)
        for k, i in enumerate(ids):
            tris.append((tv[i, 0], int(i), tv[i, 1], (1 if k == len(ids) - 1 else 0), tv[i, 2], len(ids) if k == 0 else 0))
        return 0x80000000 | first

    def make(ids):
        me = len(nodes)
        nodes.append(None)
        parts = []
        if len(ids) <= leaf_max:
            parts = [ids]
        else:
            a, b = split(ids)
            for half in (a, b):
                if len(half) > leaf_max:
                    parts += list(split(half))
                else:
                    parts.append(half)
        child, boxes = [], []
        for p in parts:
            boxes.append((lo_t[p].min(axis=0), hi_t[p].max(axis=0)))
            child.append(emit_leaf(p) if len(p) <= leaf_max else None)
        ex = np.zeros((), dtype=NODE)
        for k in range(4):
            for a, ax in enumerate(("bx", "by", "bz")):
                ex[ax][0][k] = boxes[k][0][a] if k < len(boxes) else 1.0        # empty slot: the inverted box +1 / -1
                ex[ax][1][k] = boxes[k][1][a] if k < len(boxes) else -1.0
        rec = np.zeros((), dtype=NODEQ)
        for a in range(3):
            mn = np.float32(min(b[0][a] for b in boxes))
            mx = np.float32(max(b[1][a] for b in boxes))
            s = grid_step(np.float32(mx - mn))
            while True:
                wl = wh = 0
                fits = True
                for k in range(4):
                    ql, qh = 255, 0
                    if k < len(boxes):
                        ql = int(np.floor((float(boxes[k][0][a]) - float(mn)) / float(s)))
                        qh = int(np.ceil((float(boxes[k][1][a]) - float(mn)) / float(s)))
                        ql = max(ql, 0)
                        while ql > 0 and float(mn) + ql * float(s) > float(boxes[k][0][a]):
                            ql -= 1
                        while float(mn) + qh * float(s) < float(boxes[k][1][a]):
                            qh += 1
                        fits = fits and qh <= 255
                    wl |= (ql & 255) << (8 * k)
                    wh |= (qh & 255) << (8 * k)
                if fits:
                    break
                s = np.float32(s * 2)
            rec["org"][a], rec["scale"][a] = mn, s
            rec["q"][a] = (wl, wh)
        for k in range(4):
            if k >= len(parts):
                rec["child"][k] = NONE
            elif child[k] is not None:
                rec["child"][k] = child[k]
            else:
                rec["child"][k] = make(parts[k])
        ex["child"] = rec["child"]
        ex["order"] = child_order(ex)
        nodes[me] = rec
        exact[me] = ex
        return me

    exact = {}
    make(np.arange(n))
    qn = np.array(nodes, dtype=NODEQ)
    tr = np.zeros(len(tris), dtype=TRI)
    for i, (a, prim, b, flags, c, cnt) in enumerate(tris):
        tr[i] = (a, prim, b, flags, c, cnt)
    if want_exact:
        return qn, tr, np.array([exact[i] for i in range(len(nodes))], dtype=NODE)
    return qn, tr


def run_lane_kernel(obj, any_hit, qn, tr, rays, perm=None, refill_min=8, node_exit=32, workgroups=2, bound=None, spill_cap=40):
    mem = emu.Memory()
    n = rays.shape[0]
    a_q, a_t, a_r = mem.add("qnodes", qn), mem.add("tris", tr), mem.add("rays", rays)
    out = np.full(n, 0x7e, dtype=np.uint8) if any_hit else np.zeros(n, dtype=HIT_RECORD_DTYPE)
    if not any_hit:
        out.view(np.uint32)[:] = 0x7e7e7e7e
    a_o = mem.add("out", out)
    a_c = mem.add("counter", np.zeros(COUNTER_WORDS, dtype=np.uint64))
    a_l = mem.add("leftover", np.zeros(max(n, 1), dtype=np.uint64))
    a_p = mem.add("perm", perm) if perm is not None else 0
    if bound is None:
        bound = max(1.0, float(np.abs(tr["v0"]).max()), float(np.abs(tr["v1"]).max()), float(np.abs(tr["v2"]).max()))
    lanes = workgroups * 256
    a_s = mem.add("spill", np.zeros(max(1, lanes * spill_cap), dtype=np.uint64))
    karg = struct.pack("<7Q3IfQ2I", a_q, a_t, a_r, a_o, a_c, a_l, a_p, n, refill_min, node_exit, bound, a_s, lanes, spill_cap)
    assert len(karg) == 88
    stats = emu.run_kernel(obj, "rtk_lane_hot_any" if any_hit else "rtk_lane_hot_closest", mem, karg, workgroups, 30720)
    res = mem.get(a_o).view(out.dtype).copy()
    counter = mem.get(a_c).view(np.uint64)
    left = mem.get(a_l).view(np.uint64)[:int(counter[LEFTOVER_WORD])].copy()
    # the queue heads rtk_trace_kernel deals the left-over list from (first word of each queue's line) must be untouched
    assert all(int(counter[16 + 16 * q]) == 0 for q in range(8)) and sum(int(counter[16 + 16 * q + 8]) for q in range(8)) >= (n + 63) // 64
    return res, left, stats


def chain_oracle(oracle, tv, rays):
    blobs = oracle.leaf_chain_blobs(tv, chunk=1)
    return oracle.trace_chain(blobs, rays, threads=4)


def leaf_oracle(oracle, tr, rays):
    """rtk.c's arithmetic with the LEAVES of the device records as its groups: one single-leaf blob per leaf, its triangles in the
    records' order -- full groups of four (float edge functions, all four redone in double on an exact zero, rtk.c:302-336) and a
    padded last group, exactly what the kernels must reproduce for leaves of four or more triangles."""
    starts = np.nonzero(tr["count"] > 0)[0]
    blobs = []
    for s0 in starts:
        c = int(tr["count"][s0])
        tv = np.stack([tr["v0"][s0:s0 + c], tr["v1"][s0:s0 + c], tr["v2"][s0:s0 + c]], axis=1)
        blobs += oracle.leaf_chain_blobs(tv, triangle_index=tr["prim"][s0:s0 + c], chunk=64)
    return oracle.trace_chain(blobs, rays, threads=4)


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle
    pyoracle.build_oracle()
    return pyoracle


@pytest.fixture(scope="module")
def soup():
    tv = synth.triangle_soup(600, 0.12, seed=7).reshape(-1, 3, 3)
    return tv, build_bvh4(tv)


def some_rays(n, seed):
    rays = np.concatenate([synth.rays_config1(n // 2, seed=seed), synth.rays_incoherent(n - n // 2, seed=seed + 1)])
    return np.ascontiguousarray(rays)


def check_closest(res, left, g_hits, g_mask, rays):
    n = rays.shape[0]
    done = np.ones(n, bool)
    done[(left & np.uint64(0xffffffff)).astype(np.int64)] = False
    untouched = res.view(np.uint32).reshape(n, 4)[:, 0] == 0x7e7e7e7e
    assert (untouched == ~done).all(), "a ray is either answered or handed back, never both / neither"
    hit = res["prim"] != NONE
    assert (hit[done] == g_mask[done]).all()
    sel = done & hit
    assert (res["prim"][sel] == g_hits["triangle_index"][sel]).all()
    for f in ("t", "u", "v"):
        assert (res[f][sel].view(np.uint32) == g_hits[f][sel].view(np.uint32)).all(), f
    miss = done & ~hit
    assert (res["t"][miss] == rays["max_t"][miss]).all()
    return done


def test_closest_hit_equals_oracle(lane_obj, oracle, soup):
    tv, (qn, tr) = soup
    rays = some_rays(700, 11)
    g_hits, g_mask = chain_oracle(oracle, tv, rays)
    res, left, stats = run_lane_kernel(lane_obj, False, qn, tr, rays)
    done = check_closest(res, left, g_hits, g_mask, rays)
    assert done.all() and 0.2 < g_mask.mean() < 1.0


def test_any_hit_equals_oracle_and_ray_order(lane_obj, oracle, soup):
    tv, (qn, tr) = soup
    rays = synth.rays_shadow(500, seed=5)
    rays["origin"] = rays["origin"] * 0.8 + 0.1
    g_hits, g_mask = chain_oracle(oracle, tv, rays)
    res, left, _ = run_lane_kernel(lane_obj, True, qn, tr, rays)
    assert len(left) == 0
    assert (res == g_mask.astype(np.uint8)).all() and 0.05 < g_mask.mean() < 0.95
    # the same rays through a ray order (sort words: key above, ray number below)
    rng = np.random.default_rng(3)
    order = rng.permutation(len(rays)).astype(np.uint64)
    perm = (np.arange(len(rays), dtype=np.uint64) << np.uint64(32)) | order
    res2, left2, _ = run_lane_kernel(lane_obj, True, qn, tr, rays, perm=perm, refill_min=1, node_exit=64)
    assert len(left2) == 0 and (res2 == res).all()


def test_launch_parameters_do_not_change_results(lane_obj, oracle, soup):
    tv, (qn, tr) = soup
    rays = some_rays(300, 23)
    ref, left, _ = run_lane_kernel(lane_obj, False, qn, tr, rays)
    assert len(left) == 0
    for refill_min, node_exit, wgs in ((1, 1, 1), (64, 64, 3), (17, 5, 2)):
        res, left, _ = run_lane_kernel(lane_obj, False, qn, tr, rays, refill_min=refill_min, node_exit=node_exit, workgroups=wgs)
        assert len(left) == 0 and res.tobytes() == ref.tobytes()


def test_untame_rays_are_handed_back_and_big_leaves_follow_the_group_rule(lane_obj, oracle):
    """Leaves of four to nine triangles: the closest-hit kernel computes the edge functions of their FULL groups in float (rtk.c:298-300)
    and those of the padded last group in double, bit for bit the reference on the same leaves; a ray that meets an exact zero in a
    full group (the group would be redone in double) is handed back, as are untame rays. The any-hit kernel still hands every ray
    back that meets such a leaf."""
    tv = synth.triangle_soup(200, 0.2, seed=3).reshape(-1, 3, 3)
    qn, tr = build_bvh4(tv, leaf_max=13)               # leaves of up to thirteen triangles: full groups and a padded one
    assert (tr["count"] > 3).any() and (tr["count"] > 7).any()
    rays = some_rays(256, 5)
    ex = synth.rays_exotic(64, seed=9, tris=tv)
    rays[::4] = ex[:64]
    g_hits, g_mask = leaf_oracle(oracle, tr, rays)
    res, left, _ = run_lane_kernel(lane_obj, False, qn, tr, rays)
    done = check_closest(res, left, g_hits, g_mask, rays)
    assert 0 < done.sum() < len(rays)
    d, o = rays["direction"], rays["origin"]
    untame = (d == 0).any(axis=1) | ~np.isfinite(d).all(axis=1) | ~np.isfinite(o).all(axis=1) | np.isnan(rays["min_t"]) | np.isnan(rays["max_t"])
    assert not done[untame].any()
    assert done[~untame].mean() > 0.85                 # (big leaves are no reason to hand a ray back any more; exotic rays outside 2^+-60 and exact zeros still are)
    assert len(np.unique(left)) == len(left)
    # the grouping matters: the same rays against one-triangle groups (double precision throughout) differ in some low bits
    f_hits, f_mask = chain_oracle(oracle, tv, rays)
    both = g_mask & f_mask & done
    assert (g_hits["t"][both].view(np.uint32) != f_hits["t"][both].view(np.uint32)).any()
    res_a, left_a, _ = run_lane_kernel(lane_obj, True, qn, tr, rays)
    done_a = np.ones(len(rays), bool)
    done_a[(left_a & np.uint64(0xffffffff)).astype(np.int64)] = False
    assert ((res_a == 0x7e) == ~done_a).all()
    assert (res_a[done_a] == g_mask[done_a].astype(np.uint8)).all()


def test_deep_stacks_spill_and_are_not_overrun(lane_obj, oracle):
    """Depth is forced with a degenerate chain of nodes (every node: three leaves + one inner child, all boxes the same),
    which a ray along the line must keep on its stack: 13 levels x 3 pushes > 15 LDS entries. The entries beyond the LDS
    column go to the spill area; with a spill area that is too small the rays are handed back, never written out of range."""
    m = 40
    tv = np.zeros((m, 3, 3), np.float32)
    for i in range(m):
        x = np.float32(0.02 * i)
        tv[i] = [[x, -1, -1], [x, 1, -1], [x, 0, 1]]
    # chain: node k has children [leaf(k) , leaf'(k) , leaf''(k), node k+1] with identical big boxes -> 3 pushes per level
    lo, hi = tv.reshape(-1, 3).min(axis=0) - 0.5, tv.reshape(-1, 3).max(axis=0) + 0.5
    tr = np.zeros(m, dtype=TRI)
    for i in range(m):
        tr[i] = (tv[i, 0], i, tv[i, 1], 1, tv[i, 2], 1)
    levels = m // 3
    qn = np.zeros(levels, dtype=NODEQ)
    for k in range(levels):
        for a in range(3):
            s = grid_step(np.float32(hi[a] - lo[a]))
            qn[k]["org"][a], qn[k]["scale"][a] = lo[a], s
            top = int(np.ceil((hi[a] - lo[a]) / s))
            qn[k]["q"][a] = (0, top | (top << 8) | (top << 16) | (top << 24))
        kids = [0x80000000 | (3 * k), 0x80000000 | (3 * k + 1), 0x80000000 | (3 * k + 2), (k + 1) if k + 1 < levels else NONE]
        # the inner child first in slot 0: equal keys keep the order of the references, so the chain is entered and the leaves pile up
        qn[k]["child"] = [kids[3], kids[0], kids[1], kids[2]] if k + 1 < levels else [kids[0], kids[1], kids[2], NONE]
        if k + 1 >= levels:
            for a in range(3):
                w = qn[k]["q"][a]
                qn[k]["q"][a] = (int(w[0]) | (255 << 24), int(w[1]) & 0x00ffffff)
    rays = np.zeros(8, dtype=RAY_DTYPE)
    rays["origin"] = [-1.0, 0.01, 0.0]
    rays["direction"] = [1.0, 0.001, 0.002]
    rays["origin"][:, 1] += np.arange(8) * 0.01
    rays["max_t"] = 3.0e38
    used = tv[:3 * levels]
    g_hits, g_mask = chain_oracle(oracle, used, rays)
    res, left, _ = run_lane_kernel(lane_obj, False, qn, tr, rays, bound=2.0)
    done = check_closest(res, left, g_hits, g_mask, rays)
    assert done.all() and g_mask.all()
    res_a, left_a, _ = run_lane_kernel(lane_obj, True, qn, tr, rays, bound=2.0)
    assert len(left_a) == 0 and (res_a == 1).all()
    # 13 levels x 3 pushes - 15 entries in LDS = 24 > 10: these rays cannot be finished, and are handed back
    res, left, _ = run_lane_kernel(lane_obj, False, qn, tr, rays, bound=2.0, spill_cap=10)
    done = check_closest(res, left, g_hits, g_mask, rays)
    assert not done.any() and len(left) == len(rays)
