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
