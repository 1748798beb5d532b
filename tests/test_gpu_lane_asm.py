"""GPU tests of the hand-written per-lane kernels (rtk_amd/csrc/rtk_lane_hot.S, the default for plain closest-hit and any-hit
batches): what they write must be what rtk_trace_kernel (C++, RTK_TRACE_NO_ASM) writes, byte for byte, in given and
re-ordered ray order, at batch sizes that are not whole chunks, with rays they must hand back (non-finite or zero
components, deep stacks) in the batch, and against the CPU oracle on the exported BVH."""
import numpy as np
import pytest

from rtk_amd import synth

pytestmark = pytest.mark.gpu

N = (1 << 19) + 37          # not a multiple of 64


@pytest.fixture(scope="module")
def scene(api):
    tris = synth.triangle_soup(200_000, 0.03, 7)
    return tris, api.DeviceScene.build([dict(positions=tris)])


def test_asm_equals_cpp_closest_hit(api, oracle, scene):
    tris, ds = scene
    rays = synth.rays_incoherent(N, seed=11)
    asm = ds.trace(rays, full=False)
    cpp = ds.trace(rays, opts=api.make_opts(no_asm=True), full=False)
    assert asm.tobytes() == cpp.tobytes()
    assert (asm["prim"] != 0xFFFFFFFF).mean() > 0.9
    # ... and in a re-ordered ray order, with other launch parameters
    assert ds.trace(rays, opts=api.make_opts(sort_rays=True), full=False).tobytes() == cpp.tobytes()
    assert ds.trace(rays, opts=api.make_opts(refill_min=64, node_exit=1, blocks_per_cu=2), full=False).tobytes() == cpp.tobytes()
    assert ds.trace(rays, opts=api.make_opts(refill_min=1, node_exit=64), full=False).tobytes() == cpp.tobytes()
    # against the oracle on the same BVH (the exported blob), a sample
    blob = oracle.Blob(ds.export_blob())
    sel = np.arange(0, N, 97)
    g_hits, g_mask = oracle.trace(blob, rays[sel])
    g = asm[sel]
    assert ((g["prim"] != 0xFFFFFFFF) == g_mask).all()
    h = g_mask
    assert (g["t"][h].view(np.uint32) == g_hits["t"][h].view(np.uint32)).all()
    assert (g["u"][h].view(np.uint32) == g_hits["u"][h].view(np.uint32)).all()
    assert (g["v"][h].view(np.uint32) == g_hits["v"][h].view(np.uint32)).all()


def test_asm_equals_cpp_any_hit(api, scene):
    tris, ds = scene
    rays = synth.rays_shadow(N, seed=5)
    asm = ds.trace_any(rays)
    cpp = ds.trace_any(rays, opts=api.make_opts(no_asm=True))
    assert (asm == cpp).all()
    assert 0.05 < asm.mean() < 1.0
    assert (ds.trace_any(rays, opts=api.make_opts(sort_rays=True)) == cpp).all()
    # any-hit == "closest-hit found something"
    assert (asm == (ds.trace(rays, full=False)["prim"] != 0xFFFFFFFF)).all()


def test_asm_hands_untame_rays_to_the_exact_path(api, scene):
    """Rays full of zeros, denormals, infinities and NaN intervals between ordinary ones: the assembly kernels hand them to
    rtk_trace_kernel's exact path; every record must still be what the C++ kernel alone writes."""
    tris, ds = scene
    rays = synth.rays_incoherent(N, seed=12)
    ex = synth.rays_exotic(2048, seed=9, tris=tris.reshape(-1, 3, 3))
    rays[::257][:len(ex)] = ex[:len(rays[::257])]
    asm = ds.trace(rays, full=False)
    cpp = ds.trace(rays, opts=api.make_opts(no_asm=True), full=False)
    # A ray the assembly hands back shares its wave with special rays in the C++ pass and is then traced on the EXACT nodes
    # (rtk_trace_kernel's wave_fast is wave-uniform). For rays whose boxes are below float resolution at the origin
    # (the exotic set has origins 2.5e7 away with |d| = 1e30) exact and compressed nodes may cull differently (DESIGN.md 4,
    # "where parity is undefined"): every record must be one of the C++ kernel's two answers, and THE answer where they agree.
    exact = ds.trace(rays, opts=api.make_opts(no_asm=True, exact_nodes=True), full=False)
    a, c, e = (x.view(np.uint32).reshape(-1, 4) for x in (asm, cpp, exact))
    same_c, same_e = (a == c).all(axis=1), (a == e).all(axis=1)
    assert (same_c | same_e).all()
    assert (~(same_c & same_e)).sum() <= 8 and same_c[np.arange(len(rays)) % 257 != 0].all()
    any_a, any_c = ds.trace_any(rays), ds.trace_any(rays, opts=api.make_opts(no_asm=True))
    any_e = ds.trace_any(rays, opts=api.make_opts(no_asm=True, exact_nodes=True))
    assert ((any_a == any_c) | (any_a == any_e)).all() and (any_a != any_c).sum() <= 8


def test_asm_on_a_deep_tree(api):
    """A scene whose LBVH is deep (triangles on a line with shrinking spacing): stacks outgrow the 15 LDS entries; those rays
    are handed back and finished by the C++ kernel with its spill area."""
    n = 60_000
    k = np.arange(n, dtype=np.float64)
    x = (1.0 - 0.9997 ** k).astype(np.float32)
    c = np.stack([x, np.full(n, 0.5, np.float32), np.full(n, 0.5, np.float32)], axis=1)
    off = (synth.u01(21, 0, n * 9).reshape(n, 3, 3) - np.float32(0.5)) * np.float32(0.02)
    tris = (c[:, None, :] + off).astype(np.float32).reshape(n * 3, 3)
    ds = api.DeviceScene.build([dict(positions=tris)])
    rays = synth.rays_incoherent(N, seed=14)
    asm = ds.trace(rays, full=False)
    assert asm.tobytes() == ds.trace(rays, opts=api.make_opts(no_asm=True), full=False).tobytes()
    assert (asm["prim"] != 0xFFFFFFFF).any()
    assert (ds.trace_any(rays) == ds.trace_any(rays, opts=api.make_opts(no_asm=True))).all()


def test_asm_on_far_away_and_tiny_scenes(api):
    """Coordinates around 1e4 with 0.5 detail, and a scene 1e-6 wide: the slab margin is relative to |origin| + scene bound
    (no floor of 1: the tiny scene must not open every box)."""
    n = 50_000
    for centre, size, spread in ((1.0e4, 100.0, 0.5), (0.0, 1.0e-6, 3.0e-8)):
        base = synth.u01(8, 0, n * 3).reshape(n, 1, 3) * np.float32(size) + np.float32(centre)
        off = (synth.u01(9, 0, n * 9).reshape(n, 3, 3) - np.float32(0.5)) * np.float32(spread)
        tris = (base + off).astype(np.float32).reshape(-1, 3)
        ds = api.DeviceScene.build([dict(positions=tris)])
        rays = synth.rays_incoherent(1 << 16, seed=15)
        rays["origin"] = rays["origin"] * np.float32(size) + np.float32(centre)
        rays["direction"] = rays["direction"] * np.float32(size)
        asm = ds.trace(rays, full=False)
        assert asm.tobytes() == ds.trace(rays, opts=api.make_opts(no_asm=True), full=False).tobytes()
        assert (asm["prim"] != 0xFFFFFFFF).mean() > 0.02


def test_packet_entry_points_do_not_change_a_record(api):
    """Image-shaped batches: tiles that start at their 64x64-pixel block's shared entry points (rtk_packet_entries_kernel)
    against tiles that start at the root, in the assembly and in the C++ packet kernel; a pinhole frame, a frame whose rays
    do not fit their blocks' beams everywhere (perturbed origins inside the blocks), and a frame seen from inside the scene."""
    tris = synth.triangle_soup(400_000, 0.03, 5)
    ds = api.DeviceScene.build([dict(positions=tris)])
    w = h = 1024
    frames = [synth.rays_pinhole(w, h)]
    bent = synth.rays_pinhole(w, h).copy()
    wob = (synth.u01(31, 0, w * h * 3).reshape(-1, 3) - np.float32(0.5)) * np.float32(1e-3)
    inner = np.ones((h, w), bool)
    inner[::64, :] = inner[63::64, :] = False
    inner[:, ::64] = inner[:, 63::64] = False                      # the blocks' boundary pixels keep the common origin
    bent["origin"][inner.reshape(-1)] += wob[inner.reshape(-1)]
    frames.append(bent)
    inside = synth.rays_pinhole(w, h).copy()
    inside["origin"] = (0.5, 0.5, 0.5)
    frames.append(inside)
    for rays in frames:
        img = dict(image=(w, h))
        ref = ds.trace(rays, opts=api.make_opts(no_entries=True, **img), full=False)
        assert (ref["prim"] != 0xFFFFFFFF).mean() > 0.5
        assert ds.trace(rays, opts=api.make_opts(**img), full=False).tobytes() == ref.tobytes()
        assert ds.trace(rays, opts=api.make_opts(no_asm=True, **img), full=False).tobytes() == ref.tobytes()
        assert ds.trace(rays, opts=api.make_opts(no_asm=True, no_entries=True, **img), full=False).tobytes() == ref.tobytes()
        assert ds.trace(rays, opts=api.make_opts(no_packet=True, **img), full=False).tobytes() == ref.tobytes()
        # rtk_packet_hot (per-lane slab tests) against the default, rtk_packet_beam (the tile's beam against one child plane per lane)
        assert ds.trace(rays, opts=api.make_opts(no_beam=True, **img), full=False).tobytes() == ref.tobytes()
        assert ds.trace(rays, opts=api.make_opts(no_beam=True, no_entries=True, **img), full=False).tobytes() == ref.tobytes()
        assert ds.trace(rays, opts=api.make_opts(one_tile_beam=True, **img), full=False).tobytes() == ref.tobytes()
        assert ds.trace(rays, opts=api.make_opts(one_tile_beam=True, no_entries=True, **img), full=False).tobytes() == ref.tobytes()
    # (whether the lists pay depends on how a block's beam compares with the nodes at the cut: they do at 4096 x 4096 on the
    # 1M-triangle scene -- bench.py's roofline block counts the steps -- and need not at this size; records never depend on it)
    _, with_lists = ds.trace_counted(frames[0], api.make_opts(image=(w, h)))
    _, from_root = ds.trace_counted(frames[0], api.make_opts(image=(w, h), no_entries=True))
    assert with_lists["wave_node_steps"] != from_root["wave_node_steps"] and with_lists["rays"] == from_root["rays"] == w * h


def test_packet_beam_kernel_on_every_octant_and_on_rays_of_their_own(api):
    """rtk_packet_beam: cameras on all eight sides of the scene (every direction octant has its own plane rows and order word),
    a frame whose rays all have origins, min_t and max_t of their own (the wave-wide minima / maxima of the tile's beam instead
    of the pinhole shortcut; short rays end at a max_t inside the scene), and a frame with negative min_t (its tiles are handed
    to the C++ kernel). Records equal the C++ per-lane kernel's."""
    tris = synth.triangle_soup(300_000, 0.03, 7)
    ds = api.DeviceScene.build([dict(positions=tris)])
    w = h = 512
    base = synth.rays_pinhole(w, h)
    frames = []
    for o in range(8):
        r = base.copy()
        sx, sy, sz = (-1.0 if o & 1 else 1.0), (-1.0 if o & 2 else 1.0), (-1.0 if o & 4 else 1.0)
        # an off-axis camera: every ray of the frame has the octant's signs
        r["direction"][:, 0] = (np.abs(r["direction"][:, 0]) * np.float32(0.5) + np.float32(0.05)) * np.float32(sx)
        r["direction"][:, 1] = (np.abs(r["direction"][:, 1]) * np.float32(0.5) + np.float32(0.05)) * np.float32(sy)
        r["direction"][:, 2] = np.float32(sz)
        r["origin"] = (0.5 - 0.35 * sx, 0.5 - 0.35 * sy, 0.5 - 2.0 * sz)
        frames.append(r)
    own = base.copy()
    own["origin"] += (synth.u01(41, 0, w * h * 3).reshape(-1, 3) - np.float32(0.5)) * np.float32(2e-3)
    own["min_t"] = synth.u01(42, 0, w * h) * np.float32(1.6)
    own["max_t"] = own["min_t"] + synth.u01(43, 0, w * h) * np.float32(1.5)
    frames.append(own)
    behind = base.copy()
    behind["min_t"] = -1.0
    frames.append(behind)
    # rtk_packet_beam2 walks two tiles per wave: frames in which only the left / only the right tile of every pair leaves its block's
    # beam (origins moved, except in the pixels the pre-pass samples): the pair starts at the root, the other tile is unharmed
    xs = np.arange(w * h) % w
    for pick in ((xs % 16 < 8) & (xs % 64 != 0), (xs % 16 >= 8) & (xs % 64 != 31) & (xs % 64 != 63)):
        half = base.copy()
        half["origin"][pick] += np.float32(1e-3)
        frames.append(half)
    for i, rays in enumerate(frames):
        img = dict(image=(w, h))
        ref = ds.trace(rays, opts=api.make_opts(no_packet=True, **img), full=False)
        if i != 9:
            assert (ref["prim"] != 0xFFFFFFFF).mean() > 0.2, i
        got = ds.trace(rays, opts=api.make_opts(**img), full=False)
        assert got.tobytes() == ref.tobytes(), i
        assert ds.trace(rays, opts=api.make_opts(no_beam=True, **img), full=False).tobytes() == ref.tobytes(), i
        assert ds.trace(rays, opts=api.make_opts(one_tile_beam=True, **img), full=False).tobytes() == ref.tobytes(), i
