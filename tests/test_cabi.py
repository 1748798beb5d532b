"""CPU tests of the drop-in boundary: librtk_amd.so loads and exports every function that
include/rtk.h and include/rtk_amd.h declare; POD layouts match the reference ABI.
No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rtk_amd import api, types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    names = []
    for m in re.finditer(r"^[A-Za-z_][\w\s\*]*?\b(rtk_\w+)\s*\(", text, flags=re.M):
        line_start = text.rfind("\n", 0, m.start()) + 1
        if text[line_start:m.start()].strip().startswith("typedef") or "typedef" in text[line_start:m.end()]:
            continue
        names.append(m.group(1))
    return names


def test_header_lists_are_complete():
    assert sorted(declared_functions("rtk.h")) == sorted(api.RTK_H_SYMBOLS)
    assert sorted(declared_functions("rtk_amd.h")) == sorted(api.RTK_AMD_H_SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(api.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = C.CDLL(api.LIB_PATH)
    for name in api.RTK_H_SYMBOLS + api.RTK_AMD_H_SYMBOLS:
        assert hasattr(L, name), "librtk_amd.so does not export " + name


def test_pod_layouts_match_reference_abi():
    assert types.RAY_DTYPE.itemsize == 32 and types.HIT_DTYPE.itemsize == 68
    assert types.HIT_DTYPE.fields["vertex"][1] == 12 and types.HIT_DTYPE.fields["mesh_index"][1] == 60
    assert C.sizeof(types.Mesh) == 96 and types.Mesh.position.offset == 16 and types.Mesh.index.offset == 40
    assert types.Mesh.position_cb.offset == 64 and types.Mesh.index_cb.offset == 80
    assert C.sizeof(types.SceneHeader) == 56 and types.SceneHeader.size_in_bytes.offset == 24
    assert C.sizeof(types.Task) == 40 and types.Task.cost.offset == 16
    assert np.float32(types.RTK_INF).view(np.uint32) == 0x7F7FFFFD


def test_missing_gpu_is_loud_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.RtkError):
        api.DeviceScene.upload(np.zeros(512, np.uint8))


def test_blob_validator_rejects_corrupt_scenes_without_a_gpu(oracle):
    """rtk_dev_scene_upload validates the whole blob on the host before touching HIP: every corruption
    is refused with a message; a VALID blob then fails loudly for lack of a device (no CPU fallback)."""
    import torch
    from rtk_amd import synth
    L = api.lib()
    blob = oracle.build_scene([dict(positions=synth.triangle_soup(500, 0.1, seed=5))])
    good = blob.data.copy()

    def upload(arr):
        a = np.ascontiguousarray(arr)
        h = L.rtk_dev_scene_upload(C.c_void_p(a.ctypes.data))
        return h, api.last_error()

    bad = good.copy(); bad[1] = ord("X")
    h, err = upload(bad); assert not h and "magic" in err
    bad = good.copy(); bad[8:10] = (0xaa, 0xbb)                    # big-endian marker
    h, err = upload(bad); assert not h and "endian" in err
    bad = good.copy(); bad[10] = 8                                  # sizeof_real = 8
    h, err = upload(bad); assert not h and "sizeof_real" in err
    hdr = types.SceneHeader.from_buffer_copy(good[:56].tobytes())
    bad = good.copy(); bad[24:32] = np.frombuffer(np.uint64(200).tobytes(), np.uint8)   # size_in_bytes too small
    h, err = upload(bad); assert not h
    # a child pointer far outside the blob
    bad = good.copy(); bad[128 + 96:128 + 104] = np.frombuffer(np.uint64(hdr.size_in_bytes * 4).tobytes(), np.uint8)
    h, err = upload(bad); assert not h and "out of range" in err
    # a leaf whose vertex group points outside
    nodes = good[128:hdr.leaf_offset].view(np.uint64).reshape(-1, 16)
    leaf_ptrs = nodes[:, 12:16].reshape(-1)
    first_leaf = int(leaf_ptrs[(leaf_ptrs & 1) == 1][0]) ^ 1
    while good[first_leaf:first_leaf + 8].view(np.uint64)[0] & 0x3f == 0:      # skip the null leaf
        first_leaf = int(leaf_ptrs[(leaf_ptrs & 1) == 1][np.random.RandomState(0).randint(1, 50)]) ^ 1
    bad = good.copy()
    info = int(bad[first_leaf:first_leaf + 8].view(np.uint64)[0])
    bad[first_leaf:first_leaf + 8] = np.frombuffer(np.uint64((info & 0x3f) | ((hdr.size_in_bytes * 2) & ~0x3f)).tobytes(), np.uint8)
    h, err = upload(bad); assert not h and "vertex" in err
    # offsets near 2^64 must not wrap around the range checks
    bad = good.copy(); bad[128 + 96:128 + 104] = np.frombuffer(np.uint64(0xFFFFFFFFFFFFFF80).tobytes(), np.uint8)
    h, err = upload(bad); assert not h and "out of range" in err
    bad = good.copy(); bad[128 + 96:128 + 104] = np.frombuffer(np.uint64(0xFFFFFFFFFFFFFFF9).tobytes(), np.uint8)   # leaf-tagged
    h, err = upload(bad); assert not h and "out of range" in err
    bad = good.copy()
    bad[first_leaf:first_leaf + 8] = np.frombuffer(np.uint64((info & 0x3f) | 0xFFFFFFFFFFFFFFC0).tobytes(), np.uint8)
    h, err = upload(bad); assert not h and "vertex" in err
    # a misaligned node pointer
    inner = [int(p) for p in nodes[0, 12:16] if (int(p) & 1) == 0]
    assert inner, "root has no inner child"
    bad = good.copy()
    slot = [k for k in range(4) if int(nodes[0, 12 + k]) == inner[0]][0]
    bad[128 + 96 + 8 * slot:128 + 104 + 8 * slot] = np.frombuffer(np.uint64(inner[0] + 8).tobytes(), np.uint8)
    h, err = upload(bad); assert not h and "misaligned" in err
    # not a tree: a child pointing back at the root (cycle), and one inner node referenced by two slots (DAG)
    bad = good.copy(); bad[128 + 96 + 8 * slot:128 + 104 + 8 * slot] = np.frombuffer(np.uint64(128).tobytes(), np.uint8)
    h, err = upload(bad); assert not h and "twice" in err
    if len(inner) >= 2:
        other = [k for k in range(4) if int(nodes[0, 12 + k]) == inner[1]][0]
        bad = good.copy(); bad[128 + 96 + 8 * other:128 + 104 + 8 * other] = np.frombuffer(np.uint64(inner[0]).tobytes(), np.uint8)
        h, err = upload(bad); assert not h and "twice" in err
    # the sized loader also refuses a header that claims more bytes than the file has
    a = np.ascontiguousarray(good)
    h = L.rtk_dev_scene_upload_buffer(C.c_void_p(a.ctypes.data), a.size - 64)
    assert not h and "only" in api.last_error()
    if not torch.cuda.is_available():
        h, err = upload(good)
        assert not h and "HIP device" in err


def test_c_shard_range_equals_the_python_rule():
    """rtk_amd_shard_range (the C host's partitioning, rtk_mgpu.hip) == rtk_amd.shard.shard_range (bench.py --gpus N)."""
    from rtk_amd import shard
    L = api.lib()
    f, c = C.c_size_t(), C.c_size_t()
    for n in (0, 1, 7, 64, 1000, 16777216, 16777217, 134217728, 2 ** 40 + 3):
        for world in (1, 2, 3, 4, 7, 8):
            end = 0
            for r in range(world):
                L.rtk_amd_shard_range(n, r, world, C.byref(f), C.byref(c))
                b, e = shard.shard_range(n, r, world)
                assert (f.value, f.value + c.value) == (b, e)
                assert f.value == end
                end = f.value + c.value
            assert end == n
    L.rtk_amd_shard_range(100, 5, 4, C.byref(f), C.byref(c))      # out-of-range rank: empty
    assert c.value == 0


def test_torch_generators_equal_numpy_generators():
    """bench.py makes its large scenes / batches with the torch versions of the counter-based generators: same bits."""
    import torch
    from rtk_amd import synth
    a = synth.triangle_soup(3000, 0.02, 1)
    b = synth.t_triangle_soup(3000, 0.02, 1, device="cpu", chunk=1000).numpy()
    assert (a.view(np.uint32) == b.view(np.uint32)).all()
    for np_fn, t_fn in ((synth.rays_incoherent, synth.t_rays_incoherent), (synth.rays_shadow, synth.t_rays_shadow)):
        r = np_fn(2048, first=123)
        t = t_fn(2048, first=123, device="cpu").numpy()
        rr = np.concatenate([r["origin"], r["direction"], r["min_t"][:, None], r["max_t"][:, None]], 1)
        assert (rr.view(np.uint32) == t.view(np.uint32)).all()


def test_c_striped_layout_equals_the_python_exchange():
    """rtk_mgpu_striped_segment (where the C host's striped exchange puts stripe j of shard r, rtk_mgpu.hip) is the layout of
    rtk_amd.shard.exchange_striped_start: segments in shard order, stripes by the shard_range rule, unequal shards included."""
    from rtk_amd import shard
    L = api.lib()
    f, c = C.c_size_t(), C.c_size_t()
    for world, n in ((1, 10), (2, 1001), (3, 1000), (8, 16777216 + 5), (7, 3)):
        counts = shard.shard_sizes(n, world)
        arr = (C.c_size_t * world)(*counts)
        for stripe in range(world):
            at = 0
            for r in range(world):
                b, e = shard.stripe_bounds(counts[r], world)[stripe]
                L.rtk_mgpu_striped_segment(arr, world, r, stripe, C.byref(f), C.byref(c))
                assert (f.value, c.value) == (at, e - b)
                at += e - b
        # every record of every shard lands in exactly one stripe
        total = 0
        for stripe in range(world):
            L.rtk_mgpu_striped_segment(arr, world, world - 1, stripe, C.byref(f), C.byref(c))
            total += f.value + c.value
        assert total == n
