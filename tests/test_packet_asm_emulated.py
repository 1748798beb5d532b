"""The hand-written packet kernel (rtk_amd/csrc/rtk_packet_hot.S) run on the CPU, in tests/gfx950_emu.py, against the oracle:
tiles that start at the root, tiles that start at their block's shared entry points (a Python port of
rtk_packet_entries_kernel builds the lists), tiles whose rays do not fit the block's beam, and tiles it hands back.

Oracle as in tests/test_lane_asm_emulated.py: rtk.c's arithmetic over a chain of one-triangle leaf blobs (double-precision
edge functions, rtk.c:306: what the kernel uses for its leaves of fewer than four triangles), so ids and t, u, v are bit-exact.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from rtk_amd import synth
from rtk_amd.types import HIT_RECORD_DTYPE, RAY_DTYPE

from . import gfx950_emu as emu
from .test_lane_asm_emulated import CSRC, NODE, NONE, TRI, build_bvh4, chain_oracle, child_order, oracle   # noqa: F401  (oracle: a fixture)

OBJ = os.path.join(CSRC, "obj", "rtk_packet_hot.o")
COUNTER_WORDS = 16 + 16 * 8 + 1
ENTRIES = np.dtype([("olo", "<f4", (3,)), ("ohi", "<f4", (3,)), ("rlo", "<f4", (3,)), ("rhi", "<f4", (3,)), ("count", "<u4"), ("tmin", "<f4"),
                    ("pad", "<u4", (2,)), ("e", [("ref", "<u4"), ("tlo", "<f4")], (56,))])
assert ENTRIES.itemsize == 512
W, H = 128, 64            # two 64x64-pixel blocks = 128 tiles


# the two kernels assembled from rtk_packet_hot.S: per-lane slab tests (20 KB of LDS for the stacks) and, with -DRTK_BEAM, the
# interval test of the tile's own beam (one child plane per lane; no LDS); and rtk_packet_beam2.S: two adjacent tiles per wave
# ... and rtk_packet_count2 = rtk_packet_beam2.S with -DRTK_COUNT (three step counters per pair of tiles): every test here runs on it too
KERNELS = {"rtk_packet_hot": ("rtk_packet_hot.o", 20480), "rtk_packet_beam": ("rtk_packet_beam.o", 0), "rtk_packet_beam2": ("rtk_packet_beam2.o", 0),
           "rtk_packet_count2": ("rtk_packet_count2.o", 0)}


@pytest.fixture(scope="module", params=sorted(KERNELS))
def packet_obj(request):
    subprocess.check_call(["make", "-s", "-C", CSRC, os.path.join(os.path.abspath(CSRC), "obj", "rtk_packet_hot.hsaco")])
    return request.param


@pytest.fixture(scope="module")
def scene():
    tv = synth.triangle_soup(500, 0.15, seed=11).reshape(-1, 3, 3)
    qn, tr, nodes = build_bvh4(tv, want_exact=True)
    return tv, tr, nodes


def camera(x0, y0, span, w=W, h=H, origin=(0.5, 0.5, -1.5)):
    """w x h rays from one origin, directions ((x0 + span * (x + 0.5) / w), (y0 + span * (y + 0.5) / w), 1), row-major."""
    r = np.zeros(w * h, dtype=RAY_DTYPE)
    x, y = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    r["origin"] = origin
    r["direction"][:, 0] = (np.float32(x0) + np.float32(span) * (x.reshape(-1) + np.float32(0.5)) / np.float32(w))
    r["direction"][:, 1] = (np.float32(y0) + np.float32(span) * (y.reshape(-1) + np.float32(0.5)) / np.float32(w))
    r["direction"][:, 2] = 1.0
    r["max_t"] = 3.0e38
    return r


def beam_entries(nodes, rays, w, h, bound, target=20):
    """Python port of rtk_packet_entries_kernel: per 64x64-pixel block the beam of its boundary rays and the nodes the beam
    reaches, level by level until `target` are listed, front to back."""
    f = np.float32
    bpr, rows = w // 64, h // 64
    out = np.zeros(bpr * rows, dtype=ENTRIES)
    img = rays.reshape(h, w)
    with np.errstate(all="ignore"):
        for blk in range(bpr * rows):
            bx, by = blk % bpr, blk // bpr
            sub = img[by * 64:by * 64 + 64, bx * 64:bx * 64 + 64]
            at = np.array([0, 31, 63])                                # corners, edge midpoints, centre
            edge = sub[np.ix_(at, at)].reshape(-1)
            o, d = edge["origin"], edge["direction"]
            rd = (f(1.0) / d).astype(f)
            e = out[blk]
            e["olo"], e["ohi"], e["rlo"], e["rhi"] = o.min(axis=0), o.max(axis=0), rd.min(axis=0), rd.max(axis=0)
            e["tmin"] = edge["min_t"].min()
            neg = np.signbit(d)
            ok = (neg.all(axis=0) | (~neg).all(axis=0)).all() and (np.abs(o) < 2.0 ** 19).all() and (np.abs(rd) > 2.0 ** -100).all() and \
                (np.abs(rd) < 2.0 ** 100).all() and bound < 2.0 ** 19 and not np.isnan(edge["min_t"]).any() and not np.isnan(edge["max_t"]).any()
            if not ok:
                continue
            neg = neg[0]
            m = (f(2.0 ** -21) * (np.maximum(np.abs(e["rlo"]), np.abs(e["rhi"])) * (np.maximum(np.abs(e["olo"]), np.abs(e["ohi"])) + f(bound)))).astype(f)

            def child(nd, k):
                n, far = f(e["tmin"]), f(np.inf)
                for a, ax in enumerate(("bx", "by", "bz")):
                    lo, hi = f(nd[ax][0][k]), f(nd[ax][1][k])
                    pn, pf = (hi, lo) if neg[a] else (lo, hi)
                    ns = [f(f(pn - oo) * rr) for oo in (e["olo"][a], e["ohi"][a]) for rr in (e["rlo"][a], e["rhi"][a])]
                    fs = [f(f(pf - oo) * rr) for oo in (e["olo"][a], e["ohi"][a]) for rr in (e["rlo"][a], e["rhi"][a])]
                    n = max(n, f(min(ns) - m[a]))
                    far = min(far, f(max(fs) + m[a]))
                return n <= far, n
            cur, listed, over = [(0, f(e["tmin"]))], [], False
            for level in range(14):
                if not cur or (level > 0 and len(listed) + len(cur) >= target):
                    break
                nxt = []
                for ref, t_self in cur:
                    nd = nodes[ref]
                    reached = []
                    for k in range(4):
                        c = int(nd["child"][k])
                        if c == NONE:
                            continue
                        ok_k, tlo = child(nd, k)
                        if ok_k:
                            reached.append((c, tlo))
                    if any(c & 0x80000000 for c, _ in reached):
                        listed.append((ref, t_self))         # a node with a leaf child the beam reaches is listed itself
                    else:
                        nxt += reached
                over = over or len(listed) > 56 or len(nxt) > 128
                if over:
                    break
                cur = nxt
            listed += cur
            if over or len(listed) > 56:
                continue
            order = sorted(range(len(listed)), key=lambda i: (listed[i][1], i))
            e["count"] = len(listed)
            for q, i in enumerate(order):
                e["e"][q] = (listed[i][0], listed[i][1])
    return out


def run_packet_kernel(obj, nodes, tr, rays, w, h, entries=None, bound=None, workgroups=2):
    mem = emu.Memory()
    n = w * h
    a_n, a_t, a_r = mem.add("nodes", nodes), mem.add("tris", tr), mem.add("rays", rays)
    out = np.zeros(n, dtype=HIT_RECORD_DTYPE)
    out.view(np.uint32)[:] = 0x7e7e7e7e
    a_o = mem.add("hits", out)
    a_c = mem.add("counter", np.zeros(COUNTER_WORDS, dtype=np.uint64))
    a_l = mem.add("leftover", np.zeros(n // 64, dtype=np.uint32))
    a_e = mem.add("entries", entries) if entries is not None else 0
    if bound is None:
        bound = max(1.0, float(np.abs(tr["v0"]).max()), float(np.abs(tr["v1"]).max()), float(np.abs(tr["v2"]).max()))
    bpr = w // 64
    karg = struct.pack("<6Q4IfIQ", a_n, a_t, a_r, a_o, a_c, a_l, (w // 64) * (h // 64), w, bpr, (0x100000000 + bpr - 1) // bpr, bound, 0, a_e)
    assert len(karg) == 80
    stats = emu.run_kernel(os.path.join(CSRC, "obj", KERNELS[obj][0]), obj, mem, karg, workgroups, KERNELS[obj][1], max_instructions=6_000_000)
    res = mem.get(a_o).view(HIT_RECORD_DTYPE).copy()
    counter = mem.get(a_c).view(np.uint64)
    left = mem.get(a_l).view(np.uint32)[:int(counter[10])].copy()
    run_packet_kernel.last_counters = counter[:16].copy()
    return res, left, stats


def tile_of_pixels(w, h):
    """tile number (blocks of 8x8 tiles, row-major blocks) of every pixel, row-major"""
    y, x = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    blk = (y // 64) * (w // 64) + (x // 64)
    return (blk * 64 + ((y % 64) // 8) * 8 + (x % 64) // 8).reshape(-1)


def check(res, left, g_hits, g_mask, rays, w, h):
    tiles = tile_of_pixels(w, h)
    handed = np.isin(tiles, left)
    untouched = res.view(np.uint32).reshape(-1, 4)[:, 0] == 0x7e7e7e7e
    assert (untouched == handed).all(), "a tile is either answered or handed back"
    done = ~handed
    hit = res["prim"] != NONE
    assert (hit[done] == g_mask[done]).all()
    sel = done & hit
    assert (res["prim"][sel] == g_hits["triangle_index"][sel]).all()
    for fld in ("t", "u", "v"):
        assert (res[fld][sel].view(np.uint32) == g_hits[fld][sel].view(np.uint32)).all(), fld
    return done


def test_packet_kernel_from_the_root_and_from_shared_entry_points(packet_obj, oracle, scene):
    tv, tr, nodes = scene
    rays = camera(0.02, 0.05, 0.55)                         # every direction component positive: no tile is handed back
    g_hits, g_mask = chain_oracle(oracle, tv, rays)
    assert 0.2 < g_mask.mean() < 0.98
    root, left, st_root = run_packet_kernel(packet_obj, nodes, tr, rays, W, H)
    assert len(left) == 0 and check(root, left, g_hits, g_mask, rays, W, H).all()
    ent = beam_entries(nodes, rays, W, H, bound=2.0, target=12)
    assert (ent["count"] > 1).all()
    shared, left, st_ent = run_packet_kernel(packet_obj, nodes, tr, rays, W, H, entries=ent)
    assert len(left) == 0 and shared.tobytes() == root.tobytes()
    # a zoomed-in camera (a block's beam is small against the nodes, as at 4096 x 4096 on the 1M-triangle scene): here the shared
    # walk through the top of the tree pays -- fewer instructions for the same records
    zoom = camera(0.20, 0.22, 0.05)
    z_hits, z_mask = chain_oracle(oracle, tv, zoom)
    z_root, left, st_root = run_packet_kernel(packet_obj, nodes, tr, zoom, W, H, workgroups=1)
    assert len(left) == 0 and check(z_root, left, z_hits, z_mask, zoom, W, H).all()
    z_ent = beam_entries(nodes, zoom, W, H, bound=2.0, target=6)
    z_shared, left, st_ent = run_packet_kernel(packet_obj, nodes, tr, zoom, W, H, entries=z_ent, workgroups=1)
    assert len(left) == 0 and z_shared.tobytes() == z_root.tobytes()
    print("instructions per tile: from the root %.0f, from shared entry points %.0f (lists of %s entries)" % (
        sum(s["total"] for s in st_root) / 128.0, sum(s["total"] for s in st_ent) / 128.0, z_ent["count"].tolist()))


def test_tiles_outside_the_beam_start_at_the_root(packet_obj, oracle, scene):
    """A list made for OTHER rays (a narrower camera): the tiles' own rays fail the beam check and the results are unchanged."""
    tv, tr, nodes = scene
    rays = camera(0.02, 0.05, 0.55)
    root, left, _ = run_packet_kernel(packet_obj, nodes, tr, rays, W, H, workgroups=1)
    other = beam_entries(nodes, camera(0.2, 0.2, 0.1), W, H, bound=2.0, target=12)
    res, left2, _ = run_packet_kernel(packet_obj, nodes, tr, rays, W, H, entries=other, workgroups=1)
    assert len(left) == 0 and len(left2) == 0 and res.tobytes() == root.tobytes()


def test_mixed_sign_tiles_are_handed_back(packet_obj, oracle, scene):
    """A camera that looks straight at the scene: the tiles across the image's axes have rays of both signs and go to the C++
    kernel (their block has no list either); everything else is answered, with and without lists."""
    tv, tr, nodes = scene
    rays = camera(-0.25, -0.12, 0.5)
    g_hits, g_mask = chain_oracle(oracle, tv, rays)
    ent = beam_entries(nodes, rays, W, H, bound=2.0, target=12)
    res, left, _ = run_packet_kernel(packet_obj, nodes, tr, rays, W, H, entries=ent)
    done = check(res, left, g_hits, g_mask, rays, W, H)
    assert 0 < len(left) < W * H // 64 and done.any()
    assert len(np.unique(left)) == len(left)


def comb_tree(levels):
    """A degenerate tree that makes a traversal's stack deep: node k = [node k + 1, leaf, leaf, leaf], all four boxes the same big
    box (equal order keys keep the slot order: the chain is entered, three leaves pile up per level); triangles across the view."""
    m = 3 * levels
    tv = np.zeros((m, 3, 3), np.float32)
    for i in range(m):
        z = np.float32(0.02 * i)
        tv[i] = [[0.56, 0.62, z], [0.70, 0.62, z], [0.63, 0.76, z]]       # (a corner of the view: most tiles miss the box at once)
    lo, hi = tv.reshape(-1, 3).min(axis=0) - np.float32(0.01), tv.reshape(-1, 3).max(axis=0) + np.float32(0.01)
    tr = np.zeros(m, dtype=TRI)
    for i in range(m):
        tr[i] = (tv[i, 0], i, tv[i, 1], 1, tv[i, 2], 1)
    nodes = np.zeros(levels, dtype=NODE)
    for k in range(levels):
        last = k + 1 >= levels
        kids = [NONE if last else k + 1, 0x80000000 | (3 * k), 0x80000000 | (3 * k + 1), 0x80000000 | (3 * k + 2)]
        if last:
            kids = kids[1:] + [NONE]
        for a, ax in enumerate(("bx", "by", "bz")):
            for s in range(4):
                empty = kids[s] == NONE
                nodes[k][ax][0][s] = 1.0 if empty else lo[a]
                nodes[k][ax][1][s] = -1.0 if empty else hi[a]
        nodes[k]["child"] = kids
        nodes[k]["order"] = child_order(nodes[k])
    return tv, tr, nodes


def test_a_stack_deeper_than_the_registers_hands_the_tile_back(packet_obj, oracle):
    """15 levels x 3 pending leaves fit both kernels' stacks... or not: rtk_packet_hot keeps 20 entries per lane in LDS and
    hands deeper tiles back, rtk_packet_beam 64 in two registers (a round may push seven: it hands back beyond 56). Whatever a
    kernel answers equals the oracle; what it hands back it has not touched."""
    for levels, beam_answers in ((6, True), (15, True), (22, False)):
        tv, tr, nodes = comb_tree(levels)
        rays = camera(0.02, 0.05, 0.3)
        g_hits, g_mask = chain_oracle(oracle, tv, rays)
        assert 0.01 < g_mask.mean() < 0.3
        res, left, _ = run_packet_kernel(packet_obj, nodes, tr, rays, W, H, workgroups=1, bound=6.0)
        done = check(res, left, g_hits, g_mask, rays, W, H)
        answers = beam_answers if packet_obj in ("rtk_packet_beam", "rtk_packet_beam2", "rtk_packet_count2") else levels == 6
        # (the tiles that reach the box: all answered, or all handed back)
        assert done.all() if answers else (len(left) > 0 and not g_mask[done].any())


def test_one_tile_of_a_pair_outside_the_beam(packet_obj, oracle, scene):
    """rtk_packet_beam2 walks two adjacent tiles per wave: when only the left, or only the right, tile of every pair has rays
    outside its block's beam (origins moved, except in the pixels the lists are made from), the pair starts at the root; the
    records are the oracle's either way (the one-tile kernels see the same frames)."""
    tv, tr, nodes = scene
    xs = np.arange(W * H) % W
    for pick in ((xs % 16 < 8) & (xs % 64 != 0), (xs % 16 >= 8) & (xs % 64 != 31) & (xs % 64 != 63)):
        rays = camera(0.02, 0.05, 0.55)
        rays["origin"][pick] += np.float32(2e-3)
        g_hits, g_mask = chain_oracle(oracle, tv, rays)
        ent = beam_entries(nodes, rays, W, H, bound=2.0, target=12)
        assert (ent["count"] > 1).all()
        res, left, _ = run_packet_kernel(packet_obj, nodes, tr, rays, W, H, entries=ent, workgroups=1)
        assert len(left) == 0 and check(res, left, g_hits, g_mask, rays, W, H).all()


def test_the_counting_form_counts_what_the_kernel_loads(oracle, scene):
    """rtk_packet_count2 (rtk_packet_beam2.S assembled with -DRTK_COUNT) is the kernel that is timed plus three scalar counters per
    pair of tiles. Its records are rtk_packet_beam2's; its counters equal what the emulator saw the kernel DO: one node step per
    128-byte node line asked for (global_load_dword: the kernel's only use of that opcode), one triangle fetched per 48-byte record
    asked for (s_load_dwordx8: triangles, and once per wave the kernel argument), pairs = tiles / 2; tests lie between the triangles
    fetched (every one is tested for at least one group) and twice that."""
    subprocess.check_call(["make", "-s", "-C", CSRC, os.path.join(os.path.abspath(CSRC), "obj", "rtk_packet_hot.hsaco")])
    tv, tr, nodes = scene
    for rays, with_lists in ((camera(0.02, 0.05, 0.55), False), (camera(0.20, 0.22, 0.05), True), (camera(-0.25, -0.12, 0.5), True)):
        ent = beam_entries(nodes, rays, W, H, bound=2.0, target=8) if with_lists else None
        want, left_b, _ = run_packet_kernel("rtk_packet_beam2", nodes, tr, rays, W, H, entries=ent, workgroups=1)
        got, left_c, stats = run_packet_kernel("rtk_packet_count2", nodes, tr, rays, W, H, entries=ent, workgroups=1)
        c = run_packet_kernel.last_counters
        assert got.tobytes() == want.tobytes() and sorted(left_b.tolist()) == sorted(left_c.tolist())
        pairs, node_steps, fetched, tests = (int(c[k]) for k in (11, 12, 13, 14))
        assert pairs == W * H // 128
        assert node_steps == sum(s["global_load_dword"] for s in stats) > 0
        assert fetched == sum(s["s_load_dwordx8"] for s in stats) - len(stats) > 0
        assert fetched <= tests <= 2 * fetched
        assert int(c[10]) == len(left_c)


@pytest.mark.parametrize("shift", [1, 2])
def test_the_other_two_dominant_axes(oracle, shift):
    """The triangle code exists three times, once per dominant axis of the packet; the cameras above look down z (the packed form,
    TRI_BODY_PK). Here scene and rays are rotated through the axes (x -> y -> z -> x, once and twice), so the tiles' dominant axis is
    x or y and the single-instruction form with its own permutation of vertex and origin components runs: bit for bit the oracle."""
    subprocess.check_call(["make", "-s", "-C", CSRC, os.path.join(os.path.abspath(CSRC), "obj", "rtk_packet_hot.hsaco")])
    tv = np.roll(synth.triangle_soup(500, 0.15, seed=11).reshape(-1, 3, 3), shift, axis=2).copy()
    qn, tr, nodes = build_bvh4(tv, want_exact=True)
    rays = camera(0.02, 0.05, 0.55)
    rays["origin"] = np.roll(rays["origin"], shift, axis=1)
    rays["direction"] = np.roll(rays["direction"], shift, axis=1)
    g_hits, g_mask = chain_oracle(oracle, tv, rays)
    assert 0.2 < g_mask.mean() < 0.98
    for obj in ("rtk_packet_beam2", "rtk_packet_beam"):
        res, left, _ = run_packet_kernel(obj, nodes, tr, rays, W, H, workgroups=1)
        assert len(left) == 0 and check(res, left, g_hits, g_mask, rays, W, H).all()


def test_leaves_of_four_and_more_triangles_follow_the_group_rule(oracle):
    """The reference's own builder makes leaves of 4 to 64 triangles (rtk.c:6-7); their first count & ~3 triangles form FULL groups of
    four -- float edge functions, rtk.c:298-300 -- and only the padded last group is computed in double. rtk_packet_beam2 does that
    (bit for bit the reference on the same leaves: the oracle walks one single-leaf blob per leaf); a pair that meets an exact zero
    in a full group is handed back. The one-tile kernels still hand every tile back that meets such a leaf."""
    from .test_lane_asm_emulated import leaf_oracle
    subprocess.check_call(["make", "-s", "-C", CSRC, os.path.join(os.path.abspath(CSRC), "obj", "rtk_packet_hot.hsaco")])
    tv = synth.triangle_soup(500, 0.15, seed=11).reshape(-1, 3, 3)
    qn, tr, nodes = build_bvh4(tv, leaf_max=13, want_exact=True)
    assert (tr["count"] > 7).any()
    rays = camera(0.02, 0.05, 0.55)
    g_hits, g_mask = leaf_oracle(oracle, tr, rays)
    res, left, _ = run_packet_kernel("rtk_packet_beam2", nodes, tr, rays, W, H, workgroups=1)
    done = check(res, left, g_hits, g_mask, rays, W, H)
    assert done.mean() > 0.9 and 0.2 < g_mask.mean() < 0.98
    # the grouping matters: against one-triangle groups (double precision throughout) some low bits differ
    f_hits, f_mask = chain_oracle(oracle, tv, rays)
    both = g_mask & f_mask & done
    assert (g_hits["t"][both].view(np.uint32) != f_hits["t"][both].view(np.uint32)).any()
    res1, left1, _ = run_packet_kernel("rtk_packet_beam", nodes, tr, rays, W, H, workgroups=1)
    done1 = check(res1, left1, g_hits, g_mask, rays, W, H)
    assert done1.mean() < done.mean()


def run_any_kernel(nodes, tr, rays, w, h, entries=None, workgroups=2):
    """rtk_packet_any2 (rtk_packet_beam2.S with -DRTK_ANY): the same arguments, the hits pointer is the flags', one byte per ray"""
    mem = emu.Memory()
    n = w * h
    a_n, a_t, a_r = mem.add("nodes", nodes), mem.add("tris", tr), mem.add("rays", rays)
    a_o = mem.add("flags", np.full(n, 0x7e, dtype=np.uint8))
    a_c = mem.add("counter", np.zeros(COUNTER_WORDS, dtype=np.uint64))
    a_l = mem.add("leftover", np.zeros(n // 64, dtype=np.uint32))
    a_e = mem.add("entries", entries) if entries is not None else 0
    bound = max(1.0, float(np.abs(tr["v0"]).max()), float(np.abs(tr["v1"]).max()), float(np.abs(tr["v2"]).max()))
    bpr = w // 64
    karg = struct.pack("<6Q4IfIQ", a_n, a_t, a_r, a_o, a_c, a_l, (w // 64) * (h // 64), w, bpr, (0x100000000 + bpr - 1) // bpr, bound, 0, a_e)
    stats = emu.run_kernel(os.path.join(CSRC, "obj", "rtk_packet_any2.o"), "rtk_packet_any2", mem, karg, workgroups, 0, max_instructions=6_000_000)
    flags = mem.get(a_o).view(np.uint8).copy()
    counter = mem.get(a_c).view(np.uint64)
    left = mem.get(a_l).view(np.uint32)[:int(counter[10])].copy()
    return flags, left, stats


def test_the_any_hit_form_retires_rays_at_their_first_hit(oracle, scene):
    """rtk_packet_any2: one flag per ray = "the closest hit exists" (the oracle's mask), for unbounded rays, for rays that end inside
    the scene and for rays that start behind its first surfaces; tiles it hands back are untouched. A ray is retired at its first
    accepted hit and a pair ends when every ray has its answer: never more work than the closest-hit kernel."""
    subprocess.check_call(["make", "-s", "-C", CSRC, os.path.join(os.path.abspath(CSRC), "obj", "rtk_packet_hot.hsaco")])
    tv, tr, nodes = scene
    tiles = tile_of_pixels(W, H)
    total = {}
    for name, lo, hi in (("unbounded", 0.0, 3.0e38), ("ends inside", 0.0, 2.0), ("starts inside", 1.9, 3.0e38)):
        rays = camera(0.02, 0.05, 0.55)
        rays["min_t"] = np.float32(lo)
        rays["max_t"] = np.float32(hi)
        _, g_mask = chain_oracle(oracle, tv, rays)
        assert 0.03 < g_mask.mean() < 0.98, name
        flags, left, st = run_any_kernel(nodes, tr, rays, W, H)
        assert len(left) == 0
        assert (flags == g_mask.astype(np.uint8)).all(), name
        ent = beam_entries(nodes, rays, W, H, bound=2.0, target=12)
        flags_e, left, _ = run_any_kernel(nodes, tr, rays, W, H, entries=ent)
        assert len(left) == 0 and (flags_e == flags).all(), name
        total[name] = sum(s["total"] for s in st)
    rays = camera(0.02, 0.05, 0.55)
    _, _, st_closest = run_packet_kernel("rtk_packet_beam2", nodes, tr, rays, W, H)
    # (on this sparse scene next to no tile has an answer for all of its 64 rays before the traversal ends anyway: the same work)
    assert total["unbounded"] < 1.03 * sum(s["total"] for s in st_closest), (total, sum(s["total"] for s in st_closest))
    # a frame with rays of both signs: the centre tiles are handed back untouched, the others answered
    mixed = camera(-0.25, -0.12, 0.5)
    _, m_mask = chain_oracle(oracle, tv, mixed)
    flags, left, _ = run_any_kernel(nodes, tr, mixed, W, H, entries=beam_entries(nodes, mixed, W, H, bound=2.0, target=12))
    handed = np.isin(tiles, left)
    assert 0 < handed.sum() < len(handed)
    assert (flags[handed] == 0x7e).all() and (flags[~handed] == m_mask[~handed].astype(np.uint8)).all()
