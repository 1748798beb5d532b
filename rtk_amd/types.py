"""ctypes / numpy mirrors of the POD types in include/rtk.h and include/rtk_amd.h.

Pure declarations (no compute): shared by the product binding (rtk_amd/api.py) and by
the test-only oracle binding (oracle/pyoracle.py). Sizes follow reference rtk.h:15-115
(rtk_ray 32, rtk_hit 68, rtk_vertex 16, rtk_mesh 96, rtk_scene 56, rtk_scene_desc 32).
"""
import ctypes as C

import numpy as np

RTK_INF = np.float32(3.402823e+38)  # reference rtk.h:11
RTK_NO_HIT = 0xFFFFFFFF

# -- numpy record layouts ---------------------------------------------------------------

VERTEX_DTYPE = np.dtype([("position", "<f4", (3,)), ("index", "<u4")])
RAY_DTYPE = np.dtype([("origin", "<f4", (3,)), ("direction", "<f4", (3,)), ("min_t", "<f4"), ("max_t", "<f4")])
HIT_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("vertex", VERTEX_DTYPE, (3,)),
                      ("mesh_index", "<u4"), ("triangle_index", "<u4")])
# compact device hit record of include/rtk_amd.h
HIT_RECORD_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<u4")])

assert VERTEX_DTYPE.itemsize == 16 and RAY_DTYPE.itemsize == 32
assert HIT_DTYPE.itemsize == 68 and HIT_RECORD_DTYPE.itemsize == 16

# rtk_type (reference rtk.h:45-52)
RTK_TYPE_DEFAULT, RTK_TYPE_F32, RTK_TYPE_F64, RTK_TYPE_REAL, RTK_TYPE_U16, RTK_TYPE_U32 = range(6)


# -- ctypes structures ------------------------------------------------------------------

class Buffer(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride", C.c_size_t), ("type", C.c_int)]


class Mesh(C.Structure):
    _fields_ = [("user", C.c_void_p), ("num_triangles", C.c_size_t),
                ("position", Buffer), ("index", Buffer),
                ("position_cb", C.c_void_p), ("position_cb_user", C.c_void_p),
                ("index_cb", C.c_void_p), ("index_cb_user", C.c_void_p)]


class SceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(Mesh)), ("num_meshes", C.c_size_t),
                ("log_fn", C.c_void_p), ("log_user", C.c_void_p)]


class SceneHeader(C.Structure):
    _fields_ = [("magic", C.c_char * 8), ("endian", C.c_uint16), ("sizeof_real", C.c_uint8),
                ("pad_0", C.c_uint8), ("version", C.c_uint32), ("pad_1", C.c_uint32),
                ("size_in_bytes", C.c_uint64), ("node_offset", C.c_uint64),
                ("leaf_offset", C.c_uint64), ("vertex_offset", C.c_uint64)]


class Task(C.Structure):
    _fields_ = [("build", C.c_void_p), ("fn", C.c_void_p), ("cost", C.c_double),
                ("index", C.c_size_t), ("arg", C.c_size_t)]


assert C.sizeof(Buffer) == 24 and C.sizeof(Mesh) == 96 and C.sizeof(SceneDesc) == 32
assert C.sizeof(SceneHeader) == 56 and C.sizeof(Task) == 40


class MeshSet:
    """Keeps numpy buffers alive and exposes them as an rtk_scene_desc.

    meshes: list of dicts with 'positions' (float32 [nv,3] or float64) and optional
    'indices' (uint16/uint32 [nt,3]); without indices triangle i uses vertices 3i..3i+2
    (reference rtk.c:1061-1068).
    """

    def __init__(self, meshes):
        self._keep = []
        arr = (Mesh * max(1, len(meshes)))()
        self.num_triangles = 0
        for i, m in enumerate(meshes):
            me = arr[i]
            pos = m["positions"]
            if hasattr(pos, "data_ptr"):
                # a torch tensor; if it lives on the GPU the builder reads it in place (no PCIe copy)
                assert pos.is_contiguous() and pos.dim() == 2 and pos.shape[1] == 3
                f64 = str(pos.dtype) == "torch.float64"
                assert f64 or str(pos.dtype) == "torch.float32"
                self._keep.append(pos)
                me.position.data = pos.data_ptr()
                me.position.stride = 0
                me.position.type = RTK_TYPE_F64 if f64 else RTK_TYPE_F32
                assert m.get("indices") is None or hasattr(m["indices"], "data_ptr") or True
            else:
                pos = np.ascontiguousarray(pos)
                assert pos.dtype in (np.float32, np.float64) and pos.ndim == 2 and pos.shape[1] == 3
                self._keep.append(pos)
                me.position.data = pos.ctypes.data
                me.position.stride = 0
                me.position.type = RTK_TYPE_F64 if pos.dtype == np.float64 else RTK_TYPE_F32
            idx = m.get("indices")
            if idx is not None and hasattr(idx, "data_ptr"):
                assert idx.is_contiguous() and idx.dim() == 2 and idx.shape[1] == 3
                u16 = str(idx.dtype) in ("torch.uint16", "torch.int16")
                assert u16 or str(idx.dtype) in ("torch.int32", "torch.uint32")
                self._keep.append(idx)
                me.index.data = idx.data_ptr()
                me.index.stride = 0
                me.index.type = RTK_TYPE_U16 if u16 else RTK_TYPE_U32
                me.num_triangles = idx.shape[0]
            elif idx is not None:
                idx = np.ascontiguousarray(idx)
                assert idx.dtype in (np.uint16, np.uint32) and idx.ndim == 2 and idx.shape[1] == 3
                self._keep.append(idx)
                me.index.data = idx.ctypes.data
                me.index.stride = 0
                me.index.type = RTK_TYPE_U16 if idx.dtype == np.uint16 else RTK_TYPE_U32
                me.num_triangles = idx.shape[0]
            else:
                assert pos.shape[0] % 3 == 0
                me.num_triangles = pos.shape[0] // 3
            self.num_triangles += me.num_triangles
        self._arr = arr
        self.desc = SceneDesc()
        self.desc.meshes = C.cast(arr, C.POINTER(Mesh))
        self.desc.num_meshes = len(meshes)
        # global primitive id = position in concatenated mesh order (reference rtk.c:1131-1178)
        self.mesh_base = np.cumsum([0] + [arr[i].num_triangles for i in range(len(meshes))]).astype(np.uint64)


def as_rays(a):
    a = np.ascontiguousarray(a)
    assert a.dtype == RAY_DTYPE
    return a
