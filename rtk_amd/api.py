"""Python host-side mirror of the rtk interface over librtk_amd.so (ctypes).

Names follow the reference's API (reference rtk.h:119-130): build_scene / free_scene /
trace_ray, plus the additive batch calls of include/rtk_amd.h. Everything goes through
the C-ABI; there is no Python or CPU fallback -- if the HIP library is missing or no GPU
is present the calls raise.

torch is used only as plumbing: device buffers and the current HIP stream.
"""
import ctypes as C
import os

import numpy as np

from .types import (HIT_DTYPE, HIT_RECORD_DTYPE, RAY_DTYPE, MeshSet, SceneDesc, SceneHeader)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTK_AMD_LIB") or os.path.join(HERE, "librtk_amd.so")   # RTK_AMD_LIB: kernel A/B builds only

RTK_TRACE_STATIC = 1
RTK_TRACE_NO_PACKET = 2
RTK_TRACE_SORT_RAYS = 4
RTK_TRACE_EXACT_NODES = 8
RTK_TRACE_NO_ASM = 16


class RtkError(RuntimeError):
    pass


class SceneInfo(C.Structure):
    _fields_ = [("num_triangles", C.c_uint64), ("num_meshes", C.c_uint64), ("num_nodes", C.c_uint64),
                ("node_bytes", C.c_uint64), ("triangle_bytes", C.c_uint64), ("total_device_bytes", C.c_uint64),
                ("max_depth", C.c_uint32), ("stack_entries", C.c_uint32), ("build_ms", C.c_double)]

    def as_dict(self):
        return {k: (float(getattr(self, k)) if k == "build_ms" else int(getattr(self, k))) for k, _ in self._fields_}


class SceneCheck(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "nodes_checked", "leaves_checked", "triangles_checked", "box_violations", "loose_boxes", "bad_references",
        "leaf_format_errors", "triangles_missing", "triangles_duplicated", "nodes_unreachable", "nodes_shared",
        "primitive_id_errors", "compressed_node_errors", "first_bad_index", "content_hash")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class TraceOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32), ("image_width", C.c_uint32),
                ("image_height", C.c_uint32), ("refill_min", C.c_uint32), ("blocks_per_cu", C.c_uint32),
                ("node_exit", C.c_uint32)]


class DevFilter(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("mesh_mask_bits", C.c_uint32), ("d_mesh_mask", C.c_void_p),
                ("d_ignore_prim", C.c_void_p), ("d_after", C.c_void_p)]


FILTER_FN = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_void_p, C.c_void_p)   # rtk_filter_fn (rtk.h:117)


class TraceCounters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes", C.c_uint64), ("leaves", C.c_uint64),
                ("triangles", C.c_uint64), ("hits", C.c_uint64), ("stack_spills", C.c_uint64),
                ("wave_node_steps", C.c_uint64), ("wave_triangle_steps", C.c_uint64), ("wave_rays", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class PacketCounters(C.Structure):
    """rtk_packet_counters: step counts of rtk_packet_count2 (the counting form of rtk_packet_beam2)."""
    _fields_ = [("tiles", C.c_uint64), ("pairs", C.c_uint64), ("node_steps", C.c_uint64), ("triangles_fetched", C.c_uint64),
                ("triangle_group_tests", C.c_uint64), ("tiles_handed_back", C.c_uint64),
                ("handed_back_node_steps", C.c_uint64), ("handed_back_triangle_steps", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


# every symbol include/rtk.h and include/rtk_amd.h declare
RTK_H_SYMBOLS = ["rtk_start_build", "rtk_run_task", "rtk_get_build_size", "rtk_finish_build_to",
                 "rtk_finish_build", "rtk_build_scene", "rtk_free_scene", "rtk_trace_ray", "rtk_trace_ray_filter"]
RTK_AMD_H_SYMBOLS = ["rtk_amd_last_error", "rtk_amd_device_count", "rtk_amd_set_device",
                     "rtk_dev_scene_upload", "rtk_dev_scene_build", "rtk_dev_scene_free", "rtk_dev_scene_get_info",
                     "rtk_dev_scene_mesh_base", "rtk_dev_scene_primitive_order", "rtk_dev_scene_export_size", "rtk_dev_scene_export",
                     "rtk_dev_trace_rays", "rtk_dev_trace_rays_any", "rtk_dev_expand_hits",
                     "rtk_dev_trace_rays_counted", "rtk_dev_trace_rays_any_counted", "rtk_dev_trace_rays_packet_counted", "rtk_dev_detect_image", "rtk_trace_rays", "rtk_amd_forget_scene",
                     "rtk_dev_scene_validate", "rtk_amd_release_workspace", "rtk_dev_scene_upload_buffer",
                     "rtk_dev_trace_rays_filtered", "rtk_dev_trace_rays_any_filtered", "rtk_dev_trace_status",
                     "rtk_trace_rays_filter", "rtk_amd_shard_range", "rtk_mgpu_create", "rtk_mgpu_destroy", "rtk_mgpu_num_devices",
                     "rtk_mgpu_scene", "rtk_mgpu_build", "rtk_mgpu_upload", "rtk_mgpu_trace_rays", "rtk_mgpu_trace_rays_device",
                     "rtk_amd_set_builder", "rtk_amd_get_builder", "rtk_amd_set_per_ray",
                     "rtk_mgpu_trace_rays_device_striped", "rtk_mgpu_striped_segment"]

_lib = None


def lib():
    """Load librtk_amd.so (built by rtk_amd/csrc/Makefile). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtkError("%s not found: build it with `make -C rtk_amd/csrc` (or __graft_entry__.build()); "
                       "there is no CPU fallback" % LIB_PATH)
    # torch first: both link libamdhip64.so.7 and the process must end up with ONE HIP runtime
    # (the one torch ships); loading ours first leaves torch.cuda unusable.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.rtk_amd_last_error.restype = C.c_char_p
    L.rtk_amd_device_count.restype = C.c_int
    L.rtk_amd_set_device.argtypes = [C.c_int]
    L.rtk_dev_scene_upload.restype = C.c_void_p
    L.rtk_dev_scene_upload.argtypes = [C.c_void_p]
    L.rtk_dev_scene_build.restype = C.c_void_p
    L.rtk_dev_scene_build.argtypes = [C.POINTER(SceneDesc)]
    L.rtk_dev_scene_free.argtypes = [C.c_void_p]
    L.rtk_dev_scene_get_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
    L.rtk_dev_scene_mesh_base.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.rtk_dev_scene_primitive_order.restype = C.c_longlong
    L.rtk_dev_scene_primitive_order.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.rtk_dev_scene_export_size.restype = C.c_size_t
    L.rtk_dev_scene_export_size.argtypes = [C.c_void_p]
    L.rtk_dev_scene_export.restype = C.c_void_p
    L.rtk_dev_scene_export.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.rtk_dev_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(TraceOpts), C.c_void_p]
    L.rtk_dev_trace_rays_any.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(TraceOpts), C.c_void_p]
    L.rtk_dev_expand_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rtk_dev_trace_rays_packet_counted.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(TraceOpts), C.POINTER(PacketCounters)]
    L.rtk_dev_trace_rays_counted.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(TraceOpts),
                                             C.POINTER(TraceCounters)]
    L.rtk_dev_trace_rays_any_counted.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(TraceOpts),
                                                 C.POINTER(TraceCounters)]
    L.rtk_trace_rays.restype = C.c_size_t
    L.rtk_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.rtk_amd_forget_scene.argtypes = [C.c_void_p]
    L.rtk_dev_scene_validate.argtypes = [C.c_void_p, C.POINTER(SceneCheck)]
    L.rtk_dev_scene_upload_buffer.restype = C.c_void_p
    L.rtk_dev_scene_upload_buffer.argtypes = [C.c_void_p, C.c_size_t]
    L.rtk_dev_trace_rays_filtered.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(DevFilter),
                                              C.POINTER(TraceOpts), C.c_void_p]
    L.rtk_dev_trace_rays_any_filtered.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(DevFilter),
                                                  C.POINTER(TraceOpts), C.c_void_p]
    L.rtk_dev_trace_status.argtypes = [C.c_void_p, C.c_void_p]
    L.rtk_trace_rays_filter.restype = C.c_size_t
    L.rtk_trace_rays_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rtk_amd_shard_range.restype = None
    L.rtk_amd_shard_range.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.rtk_mgpu_create.restype = C.c_void_p
    L.rtk_mgpu_create.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.rtk_mgpu_destroy.restype = None
    L.rtk_mgpu_destroy.argtypes = [C.c_void_p]
    L.rtk_mgpu_num_devices.argtypes = [C.c_void_p]
    L.rtk_mgpu_scene.restype = C.c_void_p
    L.rtk_mgpu_scene.argtypes = [C.c_void_p, C.c_int]
    L.rtk_mgpu_build.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
    L.rtk_mgpu_upload.argtypes = [C.c_void_p, C.c_void_p]
    L.rtk_mgpu_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(TraceOpts)]
    L.rtk_mgpu_trace_rays_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(TraceOpts)]
    L.rtk_mgpu_trace_rays_device_striped.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(TraceOpts)]
    L.rtk_mgpu_striped_segment.restype = None
    L.rtk_mgpu_striped_segment.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.rtk_trace_ray_filter.restype = C.c_bool
    L.rtk_trace_ray_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rtk_amd_release_workspace.restype = None
    L.rtk_build_scene.restype = C.c_void_p
    L.rtk_build_scene.argtypes = [C.POINTER(SceneDesc)]
    L.rtk_free_scene.argtypes = [C.c_void_p]
    L.rtk_trace_ray.restype = C.c_bool
    L.rtk_trace_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.rtk_start_build.restype = C.c_void_p
    L.rtk_start_build.argtypes = [C.POINTER(SceneDesc), C.c_void_p]
    L.rtk_run_task.restype = C.c_size_t
    L.rtk_run_task.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.rtk_get_build_size.restype = C.c_size_t
    L.rtk_get_build_size.argtypes = [C.c_void_p]
    L.rtk_finish_build_to.restype = C.c_void_p
    L.rtk_finish_build_to.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.rtk_finish_build.restype = C.c_void_p
    L.rtk_finish_build.argtypes = [C.c_void_p]
    _lib = L
    return L


def last_error():
    return lib().rtk_amd_last_error().decode("utf-8", "replace")


def _check(rc, what):
    if rc != 0:
        raise RtkError("%s failed (%d): %s" % (what, rc, last_error()))


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RtkError("no GPU visible to torch: the rtk_amd trace path has no CPU fallback")
    return torch


def _stream_ptr():
    return C.c_void_p(_torch().cuda.current_stream().cuda_stream)


def to_device(a):
    """numpy structured/plain array -> uint8 cuda tensor holding the same bytes."""
    torch = _torch()
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()


def make_opts(image=None, static=False, refill_min=0, blocks_per_cu=0, node_exit=0, no_packet=False, sort_rays=False, exact_nodes=False,
              no_asm=False, no_entries=False, no_beam=False, one_tile_beam=False, no_detect=False):
    o = TraceOpts()
    o.struct_size = C.sizeof(TraceOpts)
    o.flags = (RTK_TRACE_STATIC if static else 0) | (RTK_TRACE_NO_PACKET if no_packet else 0) | (RTK_TRACE_SORT_RAYS if sort_rays else 0)
    if exact_nodes:
        o.flags |= RTK_TRACE_EXACT_NODES
    if no_asm:
        o.flags |= RTK_TRACE_NO_ASM
    if no_entries:
        o.flags |= 32      # RTK_TRACE_NO_ENTRIES
    if no_beam:
        o.flags |= 64      # RTK_TRACE_NO_BEAM
    if one_tile_beam:
        o.flags |= 128     # RTK_TRACE_ONE_TILE_BEAM
    if no_detect:
        o.flags |= 256     # RTK_TRACE_NO_DETECT
    if image:
        o.image_width, o.image_height = int(image[0]), int(image[1])
    o.refill_min = refill_min
    o.blocks_per_cu = blocks_per_cu
    o.node_exit = node_exit
    return o


class DeviceScene:
    """A device-resident scene (rtk_dev_scene*)."""

    def __init__(self, handle, keepalive=None):
        if not handle:
            raise RtkError("scene creation failed: " + last_error())
        self.handle = C.c_void_p(handle)
        self._keep = keepalive

    @classmethod
    def upload(cls, blob):
        """blob: bytes-like / numpy uint8 / object with .ptr -- a scene blob in rtk format."""
        _torch()
        if hasattr(blob, "ptr"):
            return cls(lib().rtk_dev_scene_upload_buffer(C.c_void_p(blob.ptr), blob.size), blob)
        arr = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
        arr = np.ascontiguousarray(arr)
        return cls(lib().rtk_dev_scene_upload_buffer(C.c_void_p(arr.ctypes.data), arr.size), arr)

    @classmethod
    def build(cls, meshes):
        """Device LBVH build from mesh dicts (see rtk_amd.types.MeshSet)."""
        _torch()
        ms = meshes if isinstance(meshes, MeshSet) else MeshSet(meshes)
        return cls(lib().rtk_dev_scene_build(C.byref(ms.desc)), ms)

    def free(self):
        if self.handle:
            lib().rtk_dev_scene_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def info(self):
        i = SceneInfo()
        _check(lib().rtk_dev_scene_get_info(self.handle, C.byref(i)), "rtk_dev_scene_get_info")
        return i.as_dict()

    def validate(self):
        """Device-side structural check; returns (ok, counts dict)."""
        c = SceneCheck()
        rc = lib().rtk_dev_scene_validate(self.handle, C.byref(c))
        if rc not in (0, -5):
            raise RtkError("rtk_dev_scene_validate failed (%d): %s" % (rc, last_error()))
        return rc == 0, c.as_dict()

    def mesh_base(self):
        n = self.info()["num_meshes"] + 1
        out = np.zeros(n, np.uint64)
        rc = lib().rtk_dev_scene_mesh_base(self.handle, out.ctypes.data, n)
        if rc < 0:
            raise RtkError(last_error())
        return out

    def primitive_order(self):
        n = self.info()["num_triangles"]
        out = np.zeros(max(n, 1), np.uint32)
        rc = lib().rtk_dev_scene_primitive_order(self.handle, out.ctypes.data, out.size)
        if rc < 0:
            raise RtkError(last_error())
        return out[:n]

    def export_blob(self):
        size = lib().rtk_dev_scene_export_size(self.handle)
        if size == 0:
            raise RtkError("rtk_dev_scene_export_size: " + last_error())
        raw = np.zeros(size + 128, np.uint8)
        off = (-raw.ctypes.data) % 128
        buf = raw[off:off + size]
        if not lib().rtk_dev_scene_export(self.handle, buf.ctypes.data, size):
            raise RtkError("rtk_dev_scene_export: " + last_error())
        return buf

    # -- device-pointer batch calls (tensors are uint8 cuda tensors) --

    def trace_device(self, d_rays, n, d_records=None, opts=None):
        torch = _torch()
        if d_records is None:
            d_records = torch.empty(n * 16, dtype=torch.uint8, device="cuda")
        _check(lib().rtk_dev_trace_rays(self.handle, C.c_void_p(d_rays.data_ptr()), n, C.c_void_p(d_records.data_ptr()),
                                        C.byref(opts) if opts is not None else None, _stream_ptr()), "rtk_dev_trace_rays")
        return d_records

    def trace_any_device(self, d_rays, n, d_occluded=None, opts=None):
        torch = _torch()
        if d_occluded is None:
            d_occluded = torch.empty(n, dtype=torch.uint8, device="cuda")
        _check(lib().rtk_dev_trace_rays_any(self.handle, C.c_void_p(d_rays.data_ptr()), n, C.c_void_p(d_occluded.data_ptr()),
                                            C.byref(opts) if opts is not None else None, _stream_ptr()), "rtk_dev_trace_rays_any")
        return d_occluded

    def expand_device(self, d_records, n):
        torch = _torch()
        d_hits = torch.empty(n * 68, dtype=torch.uint8, device="cuda")
        d_mask = torch.empty(n, dtype=torch.uint8, device="cuda")
        _check(lib().rtk_dev_expand_hits(self.handle, C.c_void_p(d_records.data_ptr()), n, C.c_void_p(d_hits.data_ptr()),
                                         C.c_void_p(d_mask.data_ptr()), _stream_ptr()), "rtk_dev_expand_hits")
        return d_hits, d_mask

    # -- numpy conveniences used by the tests --

    def trace(self, rays, opts=None, full=True):
        """rays: numpy RAY_DTYPE. Returns (hits HIT_DTYPE, mask bool, records HIT_RECORD_DTYPE)."""
        torch = _torch()
        rays = np.ascontiguousarray(rays)
        assert rays.dtype == RAY_DTYPE
        n = rays.shape[0]
        d_rays = to_device(rays)
        d_rec = self.trace_device(d_rays, n, opts=opts)
        rec = d_rec.cpu().numpy().view(HIT_RECORD_DTYPE)
        if not full:
            torch.cuda.synchronize()
            return rec
        d_hits, d_mask = self.expand_device(d_rec, n)
        hits = d_hits.cpu().numpy().view(HIT_DTYPE)
        mask = d_mask.cpu().numpy().astype(bool)
        return hits, mask, rec

    def trace_filtered(self, rays, mesh_mask=None, ignore_prim=None, after=None, any_hit=False, opts=None):
        """Built-in device filters (rtk_dev_filter). mesh_mask: iterable of bools per mesh; ignore_prim: uint32 per
        ray; after: HIT_RECORD_DTYPE per ray. Returns records (closest) or an occluded bool array (any_hit)."""
        torch = _torch()
        rays = np.ascontiguousarray(rays)
        n = rays.shape[0]
        d_rays = to_device(rays)
        f = DevFilter()
        f.struct_size = C.sizeof(DevFilter)
        keep = []
        if mesh_mask is not None:
            bits = np.asarray(mesh_mask, bool)
            words = np.zeros((len(bits) + 31) // 32, np.uint32)
            for m, b in enumerate(bits):
                if b:
                    words[m >> 5] |= np.uint32(1 << (m & 31))
            d = to_device(words); keep.append(d)
            f.d_mesh_mask, f.mesh_mask_bits = d.data_ptr(), len(bits)
        if ignore_prim is not None:
            d = to_device(np.ascontiguousarray(ignore_prim, np.uint32)); keep.append(d)
            f.d_ignore_prim = d.data_ptr()
        if after is not None:
            a = np.ascontiguousarray(after)
            assert a.dtype == HIT_RECORD_DTYPE and a.shape[0] == n
            d = to_device(a); keep.append(d)
            f.d_after = d.data_ptr()
        if any_hit:
            d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
            _check(lib().rtk_dev_trace_rays_any_filtered(self.handle, C.c_void_p(d_rays.data_ptr()), n, C.c_void_p(d_out.data_ptr()),
                                                         C.byref(f), C.byref(opts) if opts is not None else None, _stream_ptr()),
                   "rtk_dev_trace_rays_any_filtered")
            _check(lib().rtk_dev_trace_status(self.handle, _stream_ptr()), "rtk_dev_trace_status")
            return d_out.cpu().numpy().astype(bool)
        d_out = torch.empty(n * 16, dtype=torch.uint8, device="cuda")
        _check(lib().rtk_dev_trace_rays_filtered(self.handle, C.c_void_p(d_rays.data_ptr()), n, C.c_void_p(d_out.data_ptr()),
                                                 C.byref(f), C.byref(opts) if opts is not None else None, _stream_ptr()),
               "rtk_dev_trace_rays_filtered")
        _check(lib().rtk_dev_trace_status(self.handle, _stream_ptr()), "rtk_dev_trace_status")
        return d_out.cpu().numpy().view(HIT_RECORD_DTYPE)

    def trace_any(self, rays, opts=None):
        rays = np.ascontiguousarray(rays)
        n = rays.shape[0]
        d_rays = to_device(rays)
        return self.trace_any_device(d_rays, n, opts=opts).cpu().numpy().astype(bool)

    def trace_any_counted(self, rays, opts=None):
        torch = _torch()
        if hasattr(rays, "data_ptr"):       # a torch tensor of rays already on the device ([n, 8] float32 or raw bytes)
            d_rays, n = rays, rays.numel() * rays.element_size() // 32
        else:
            rays = np.ascontiguousarray(rays)
            n = rays.shape[0]
            d_rays = to_device(rays)
        d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
        ctr = TraceCounters()
        torch.cuda.synchronize()
        _check(lib().rtk_dev_trace_rays_any_counted(self.handle, C.c_void_p(d_rays.data_ptr()), n, C.c_void_p(d_occ.data_ptr()),
                                                    C.byref(opts) if opts is not None else None, C.byref(ctr)),
               "rtk_dev_trace_rays_any_counted")
        return d_occ.cpu().numpy().astype(bool), ctr.as_dict()

    def trace_counted(self, rays, opts=None):
        torch = _torch()
        if hasattr(rays, "data_ptr"):
            d_rays, n = rays, rays.numel() * rays.element_size() // 32
        else:
            rays = np.ascontiguousarray(rays)
            n = rays.shape[0]
            d_rays = to_device(rays)
        d_rec = torch.empty(n * 16, dtype=torch.uint8, device="cuda")
        ctr = TraceCounters()
        torch.cuda.synchronize()
        _check(lib().rtk_dev_trace_rays_counted(self.handle, C.c_void_p(d_rays.data_ptr()), n, C.c_void_p(d_rec.data_ptr()),
                                                C.byref(opts) if opts is not None else None, C.byref(ctr)),
               "rtk_dev_trace_rays_counted")
        return d_rec.cpu().numpy().view(HIT_RECORD_DTYPE), ctr.as_dict()


    def detect_image(self, rays):
        """rtk_dev_detect_image: (width, height) of the row-major image the batch is, or (0, 0)."""
        rays = np.ascontiguousarray(rays)
        d_rays = to_device(rays)
        w, h = C.c_uint32(0), C.c_uint32(0)
        _check(lib().rtk_dev_detect_image(self.handle, C.c_void_p(d_rays.data_ptr()), C.c_size_t(rays.shape[0]), C.byref(w), C.byref(h), _stream_ptr()), "rtk_dev_detect_image")
        return int(w.value), int(h.value)

    def trace_packet_counted(self, rays, opts):
        """rtk_dev_trace_rays_packet_counted: (records, counters of the hand-written packet kernel itself)."""
        torch = _torch()
        if hasattr(rays, "data_ptr"):
            d_rays, n = rays, rays.numel() * rays.element_size() // 32
        else:
            rays = np.ascontiguousarray(rays)
            n = rays.shape[0]
            d_rays = to_device(rays)
        d_rec = torch.empty(n * 16, dtype=torch.uint8, device="cuda")
        ctr = PacketCounters()
        torch.cuda.synchronize()
        _check(lib().rtk_dev_trace_rays_packet_counted(self.handle, C.c_void_p(d_rays.data_ptr()), n, C.c_void_p(d_rec.data_ptr()),
                                                       C.byref(opts) if opts is not None else None, C.byref(ctr)),
               "rtk_dev_trace_rays_packet_counted")
        return d_rec.cpu().numpy().view(HIT_RECORD_DTYPE), ctr.as_dict()


# -- the reference's own entry points, host pointers (reference rtk.h:126-130) --

def build_scene(meshes):
    """rtk_build_scene: returns (scene pointer, MeshSet keepalive). Release with free_scene."""
    _torch()
    ms = meshes if isinstance(meshes, MeshSet) else MeshSet(meshes)
    p = lib().rtk_build_scene(C.byref(ms.desc))
    if not p:
        raise RtkError("rtk_build_scene failed: " + last_error())
    return p, ms


def free_scene(scene_ptr):
    lib().rtk_free_scene(C.c_void_p(scene_ptr))


def scene_bytes(scene_ptr):
    hdr = SceneHeader.from_address(scene_ptr)
    return np.ctypeslib.as_array((C.c_uint8 * hdr.size_in_bytes).from_address(scene_ptr)).copy()


def trace_ray(scene_ptr, ray):
    """rtk_trace_ray: one ray (RAY_DTYPE scalar). Returns HIT_DTYPE record or None."""
    _torch()
    r = np.ascontiguousarray(ray).reshape(1)
    h = np.zeros(1, HIT_DTYPE)
    ok = lib().rtk_trace_ray(C.c_void_p(scene_ptr), r.ctypes.data, h.ctypes.data)
    return h[0] if ok else None


def trace_rays_filter(scene_ptr, rays, accept):
    """rtk_trace_rays_filter with a Python callback accept(ray_index, hit) -> bool; hit is a HIT_DTYPE record."""
    _torch()
    rays = np.ascontiguousarray(rays)
    n = rays.shape[0]
    hits = np.zeros(n, HIT_DTYPE)
    mask = np.zeros(n, np.uint8)
    base = rays.ctypes.data

    def cb(user, ray_ptr, hit_ptr):
        h = np.ctypeslib.as_array((C.c_uint8 * 68).from_address(hit_ptr)).view(HIT_DTYPE)[0]
        return bool(accept((ray_ptr - base) // 32, h))
    fn = FILTER_FN(cb)
    r = lib().rtk_trace_rays_filter(C.c_void_p(scene_ptr), rays.ctypes.data, n, hits.ctypes.data, mask.ctypes.data,
                                    C.cast(fn, C.c_void_p), None)
    if r == C.c_size_t(-1).value:
        raise RtkError("rtk_trace_rays_filter failed: " + last_error())
    return hits, mask.astype(bool)


def trace_rays(scene_ptr, rays):
    """rtk_trace_rays: host arrays in, host arrays out (PCIe inclusive)."""
    _torch()
    rays = np.ascontiguousarray(rays)
    n = rays.shape[0]
    hits = np.zeros(n, HIT_DTYPE)
    mask = np.zeros(n, np.uint8)
    r = lib().rtk_trace_rays(C.c_void_p(scene_ptr), rays.ctypes.data, n, hits.ctypes.data, mask.ctypes.data)
    if r == C.c_size_t(-1).value:
        raise RtkError("rtk_trace_rays failed: " + last_error())
    return hits, mask.astype(bool)
