"""ctypes binding of the CPU oracle (oracle/librtk_oracle.so) and, where it exists, of the
real reference build (oracle/_ref/librtk_ref.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from rtk_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from rtk_amd.types import HIT_DTYPE, RAY_DTYPE, VERTEX_DTYPE, MeshSet, SceneDesc

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "librtk_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "librtk_ref.so")

TIES_REFERENCE, TIES_CANONICAL = 0, 1


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes", C.c_uint64), ("leaves", C.c_uint64),
                ("tri_groups", C.c_uint64), ("hits", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Filter(C.Structure):
    _fields_ = [("mesh_mask", C.c_void_p), ("mesh_mask_bits", C.c_uint32), ("ignore_mesh", C.c_void_p),
                ("ignore_tri", C.c_void_p), ("after_t", C.c_void_p), ("after_mesh", C.c_void_p),
                ("after_tri", C.c_void_p), ("callback", C.c_void_p), ("user", C.c_void_p)]


FILTER_CB = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_size_t, C.c_void_p)


def build_oracle(force=False):
    """Compile the C restatement (and the reference shim when /root/reference is present)."""
    if force or not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "rtk_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    ref_root = os.environ.get("RTK_REFERENCE", "/root/reference")
    if os.path.exists(os.path.join(ref_root, "rtk.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref", "RTK_REFERENCE=" + ref_root])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.ora_build_scene.restype = C.c_void_p
        L.ora_build_scene.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_size_t)]
        L.ora_free.argtypes = [C.c_void_p]
        L.ora_make_leaf_blob.restype = C.c_size_t
        L.ora_make_leaf_blob.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ora_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.POINTER(Counters)]
        L.ora_trace_chain.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_void_p, C.c_int]
        L.ora_ray_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_validate_blob.restype = C.c_int
        L.ora_validate_blob.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ora_max_threads.restype = C.c_int
        L.ora_trace_rays_records.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
        L.ora_alloc_spread.restype = C.c_void_p
        L.ora_alloc_spread.argtypes = [C.c_size_t, C.c_void_p, C.c_int]
        L.ora_trace_rays_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int,
                                            C.POINTER(Filter)]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(REF_SO)


_ref = None


def ref():
    """The REAL reference compiled verbatim (build container only)."""
    global _ref
    if _ref is None:
        R = C.CDLL(REF_SO)
        R.ref_trace_ray.restype = C.c_int
        R.ref_trace_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_trace_chain.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_void_p, C.c_int]
        R.ref_sizeof.restype = C.c_size_t
        R.ref_sizeof.argtypes = [C.c_int]
        _ref = R
    return _ref


def default_threads():
    """Threads for the CPU legs: the cores this process may use, capped at a one-GPU box's share (16)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16, lib().ora_max_threads()))


class Blob:
    """A scene blob held in a numpy byte array (64-byte aligned view)."""

    def __init__(self, data):
        self.data = data  # np.uint8 array, aligned

    @property
    def ptr(self):
        return self.data.ctypes.data

    @property
    def size(self):
        return self.data.size

    def tobytes(self):
        return self.data.tobytes()


def _aligned_bytes(n, align=128):
    raw = np.zeros(n + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n]


def build_scene(meshes):
    """CPU oracle build: list of mesh dicts (see rtk_amd.types.MeshSet) -> Blob."""
    ms = meshes if isinstance(meshes, MeshSet) else MeshSet(meshes)
    size = C.c_size_t(0)
    p = lib().ora_build_scene(C.byref(ms.desc), C.byref(size))
    if not p:
        raise MemoryError("ora_build_scene failed")
    out = _aligned_bytes(size.value)
    C.memmove(out.ctypes.data, p, size.value)
    lib().ora_free(p)
    return Blob(out)


def validate_blob(blob):
    nn, nl, nt = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = lib().ora_validate_blob(blob.ptr, blob.size, C.byref(nn), C.byref(nl), C.byref(nt))
    return rc, dict(nodes=nn.value, leaves=nl.value, tris=nt.value)


def trace(blob, rays, ties=TIES_CANONICAL, threads=None, counters=False):
    rays = np.ascontiguousarray(rays)
    assert rays.dtype == RAY_DTYPE
    n = rays.shape[0]
    hits = np.zeros(n, dtype=HIT_DTYPE)
    mask = np.zeros(n, dtype=np.uint8)
    ctr = Counters()
    lib().ora_trace_rays(blob.ptr, rays.ctypes.data, n, hits.ctypes.data, mask.ctypes.data, ties,
                         threads or default_threads(), C.byref(ctr) if counters else None)
    if counters:
        return hits, mask.astype(bool), ctr.as_dict()
    return hits, mask.astype(bool)


RECORD_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("triangle_index", "<u4")])


class SpreadBuffer:
    """Memory whose pages the OpenMP worker threads first-touched round robin (ora_alloc_spread), optionally a copy of
    `src` (a Blob or a numpy array): the CPU baseline's blob, rays and output live in these."""

    def __init__(self, nbytes, src=None, threads=None):
        self.size = int(nbytes)
        sp = None
        if src is not None:
            sp = src.ptr if isinstance(src, Blob) else np.ascontiguousarray(src).ctypes.data
        self.ptr = lib().ora_alloc_spread(self.size, sp, threads or default_threads())
        if not self.ptr:
            raise MemoryError("ora_alloc_spread")

    def view(self, dtype):
        a = np.ctypeslib.as_array((C.c_uint8 * self.size).from_address(self.ptr))
        return a.view(dtype)

    def free(self):
        if self.ptr:
            lib().ora_free(self.ptr)
            self.ptr = None


def trace_records(blob, rays, out, ties=TIES_CANONICAL, threads=None):
    """The timing driver of the CPU baseline: closest hits of `rays` (a RAY_DTYPE array or a SpreadBuffer holding one) as
    16-byte records into `out` (a SpreadBuffer of n * 16 bytes, allocated and touched by the caller). Returns the record view."""
    rp = rays.ptr if isinstance(rays, SpreadBuffer) else rays.ctypes.data
    n = (rays.size // RAY_DTYPE.itemsize) if isinstance(rays, SpreadBuffer) else rays.shape[0]
    assert out.size >= n * RECORD_DTYPE.itemsize
    lib().ora_trace_rays_records(blob.ptr, rp, n, out.ptr, ties, threads or default_threads())
    return out.view(RECORD_DTYPE)[:n]


def trace_filtered(blob, rays, mesh_mask=None, ignore=None, after=None, callback=None, threads=None):
    """Closest candidate accepted by the filters (oracle statement of rtk_dev_filter / rtk_trace_rays_filter).

    mesh_mask: bools per mesh; ignore: (mesh[n], tri[n]) uint32 arrays; after: (t[n], mesh[n], tri[n]) with mesh
    0xffffffff = off; callback(ray_index, hit_record) -> bool (HIT_DTYPE scalar)."""
    rays = np.ascontiguousarray(rays)
    assert rays.dtype == RAY_DTYPE
    n = rays.shape[0]
    hits = np.zeros(n, dtype=HIT_DTYPE)
    mask = np.zeros(n, dtype=np.uint8)
    f = Filter()
    keep = []
    if mesh_mask is not None:
        bits = np.asarray(mesh_mask, bool)
        words = np.zeros((len(bits) + 31) // 32, np.uint32)
        for m, b in enumerate(bits):
            if b:
                words[m >> 5] |= np.uint32(1 << (m & 31))
        keep.append(words)
        f.mesh_mask, f.mesh_mask_bits = words.ctypes.data, len(bits)
    if ignore is not None:
        im, it = (np.ascontiguousarray(a, np.uint32) for a in ignore)
        keep += [im, it]
        f.ignore_mesh, f.ignore_tri = im.ctypes.data, it.ctypes.data
    if after is not None:
        at = np.ascontiguousarray(after[0], np.float32)
        am, ai = (np.ascontiguousarray(a, np.uint32) for a in after[1:])
        keep += [at, am, ai]
        f.after_t, f.after_mesh, f.after_tri = at.ctypes.data, am.ctypes.data, ai.ctypes.data
    if callback is not None:
        def cb(user, i, hit_ptr):
            h = np.ctypeslib.as_array((C.c_uint8 * 68).from_address(hit_ptr)).view(HIT_DTYPE)[0]
            return bool(callback(int(i), h))
        fn = FILTER_CB(cb)
        keep.append(fn)
        f.callback = C.cast(fn, C.c_void_p)
    lib().ora_trace_rays_filter(blob.ptr, rays.ctypes.data, n, hits.ctypes.data, mask.ctypes.data,
                                threads or default_threads(), C.byref(f))
    return hits, mask.astype(bool)


def leaf_chain_blobs(tri_vertices, mesh_index=None, triangle_index=None, vertex_index=None, chunk=60):
    """Single-leaf blobs over consecutive chunks of triangles.

    tri_vertices: float32 [n,3,3]. Returns a list of Blob."""
    tv = np.ascontiguousarray(tri_vertices, dtype=np.float32).reshape(-1, 3, 3)
    n = tv.shape[0]
    mesh_index = np.zeros(n, np.uint32) if mesh_index is None else np.ascontiguousarray(mesh_index, np.uint32)
    triangle_index = np.arange(n, dtype=np.uint32) if triangle_index is None else np.ascontiguousarray(triangle_index, np.uint32)
    verts = np.zeros((n, 3), dtype=VERTEX_DTYPE)
    verts["position"] = tv
    verts["index"] = (np.arange(3 * n, dtype=np.uint32).reshape(n, 3) if vertex_index is None
                      else np.asarray(vertex_index, np.uint32).reshape(n, 3))
    blobs = []
    for a in range(0, max(n, 1), chunk):
        m = min(chunk, n - a)
        cap = 128 + 128 + 64 + 64 * ((8 + 8 * 64 + 4 * 64) // 64 + 1) + 128 + 16 * 3 * 64 + 256
        buf = _aligned_bytes(cap)
        v = np.ascontiguousarray(verts[a:a + m])
        mi = np.ascontiguousarray(mesh_index[a:a + m])
        ti = np.ascontiguousarray(triangle_index[a:a + m])
        sz = lib().ora_make_leaf_blob(v.ctypes.data, mi.ctypes.data, ti.ctypes.data, m, buf.ctypes.data, cap)
        assert sz > 0
        blobs.append(Blob(buf[:sz]))
    return blobs


def _chain(fn, blobs, rays, threads):
    rays = np.ascontiguousarray(rays)
    assert rays.dtype == RAY_DTYPE
    n = rays.shape[0]
    hits = np.zeros(n, dtype=HIT_DTYPE)
    mask = np.zeros(n, dtype=np.uint8)
    ptrs = (C.c_void_p * len(blobs))(*[b.ptr for b in blobs])
    fn(ptrs, len(blobs), rays.ctypes.data, n, hits.ctypes.data, mask.ctypes.data, threads or default_threads())
    return hits, mask.astype(bool)


def trace_chain(blobs, rays, threads=None):
    """Oracle restatement over a leaf chain."""
    return _chain(lib().ora_trace_chain, blobs, rays, threads)


def ref_trace_chain(blobs, rays, threads=None):
    """REAL reference rtk_trace_ray over a leaf chain (build container only)."""
    return _chain(ref().ref_trace_chain, blobs, rays, threads)


def ray_setup(ray):
    r = np.ascontiguousarray(ray).reshape(1)
    k = np.zeros(3, np.uint32)
    sh = np.zeros(3, np.float32)
    sm = np.zeros(1, np.uint32)
    lib().ora_ray_setup(r.ctypes.data, k.ctypes.data, sh.ctypes.data, sm.ctypes.data)
    return k, sh, int(sm[0])
