"""Shared comparison helpers for the parity tests.

Parity bar (BASELINE.json north_star): hit/miss and primitive ids
bit-exact; |dt| <= 1e-5*|t|; |du|,|dv| <= 1e-5*max(1,|value|).
"""
import hashlib
import os

import numpy as np

REL_TOL = 1e-5


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def compare_hits(mask, mesh, tri, t, u, v, g_mask, g_mesh, g_tri, g_t, g_u, g_v, what=""):
    """Assert parity of one result set against an expected one; returns stats."""
    mask = np.asarray(mask).astype(bool)
    g_mask = np.asarray(g_mask).astype(bool)
    bad = np.nonzero(mask != g_mask)[0]
    assert bad.size == 0, "%s: hit/miss differs on %d rays, first %s" % (what, bad.size, bad[:8])
    m = mask
    bad = np.nonzero((np.asarray(mesh)[m] != np.asarray(g_mesh)[m]) | (np.asarray(tri)[m] != np.asarray(g_tri)[m]))[0]
    assert bad.size == 0, "%s: primitive id differs on %d hits, first rays %s" % (what, bad.size, np.nonzero(m)[0][bad[:8]])
    t, u, v = (np.asarray(x, np.float32)[m].astype(np.float64) for x in (t, u, v))
    gt, gu, gv = (np.asarray(x, np.float32)[m].astype(np.float64) for x in (g_t, g_u, g_v))
    if t.size == 0:
        return dict(hits=0, max_rel_t=0.0, max_abs_uv=0.0, bit_exact=1.0)
    rel_t = np.abs(t - gt) / np.maximum(np.abs(gt), 1e-300)
    du = np.abs(u - gu) / np.maximum(1.0, np.abs(gu))
    dv = np.abs(v - gv) / np.maximum(1.0, np.abs(gv))
    assert rel_t.max() <= REL_TOL, "%s: t off by %.3g rel" % (what, rel_t.max())
    assert du.max() <= REL_TOL and dv.max() <= REL_TOL, "%s: u/v off by %.3g" % (what, max(du.max(), dv.max()))
    exact = float(np.mean((t == gt) & (u == gu) & (v == gv)))
    return dict(hits=int(m.sum()), max_rel_t=float(rel_t.max()), max_abs_uv=float(max(du.max(), dv.max())), bit_exact=exact)


def compare_hits_struct(hits, mask, g, what=""):
    """hits: HIT_DTYPE array + bool mask; g: golden npz with hit_* arrays."""
    return compare_hits(mask, hits["mesh_index"], hits["triangle_index"], hits["t"], hits["u"], hits["v"],
                        g["hit_mask"], g["hit_mesh"], g["hit_tri"], g["hit_t"], g["hit_u"], g["hit_v"], what)


def random_mixed_scene(seed):
    """A random scene of several meshes with mixed index types (implicit, u16, u32), float32 / float64 positions, strided
    buffers and callback meshes (<= 128 triangles per call, rtk.c:1141-1148). Returns (scene_desc, keepalive, tris[n,3,3],
    mesh_index[n], triangle_index[n], vertex_index[n,3]) -- the last four in concatenated mesh order."""
    import ctypes as C
    from rtk_amd import synth
    from rtk_amd.types import Mesh, SceneDesc, RTK_TYPE_F32, RTK_TYPE_F64, RTK_TYPE_U16, RTK_TYPE_U32
    rng = np.random.RandomState(seed)
    POS_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Mesh), C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_size_t)
    IDX_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Mesh), C.POINTER(C.c_uint32), C.c_size_t, C.c_size_t)
    meshes, keep, all_tris, mesh_of, tri_of, vidx_of = [], [], [], [], [], []
    for mi in range(rng.randint(2, 6)):
        nt = int(rng.randint(1, 700))
        soup = synth.triangle_soup(nt, 0.2, seed=100 * seed + mi)
        verts, inv = np.unique(soup, axis=0, return_inverse=True)
        idx = inv.reshape(-1, 3)
        kind = rng.randint(0, 5)
        m = Mesh()
        m.num_triangles = nt
        if kind == 0:                                   # implicit indices, float32
            v = np.ascontiguousarray(soup.astype(np.float32)); keep.append(v)
            m.position.data = v.ctypes.data; m.position.type = RTK_TYPE_F32
            vidx = np.arange(3 * nt, dtype=np.uint32).reshape(-1, 3)
        elif kind == 1 and len(verts) < 65536:           # u16 indices, float64 positions
            v = np.ascontiguousarray(verts.astype(np.float64)); i16 = np.ascontiguousarray(idx.astype(np.uint16)); keep += [v, i16]
            m.position.data = v.ctypes.data; m.position.type = RTK_TYPE_F64
            m.index.data = i16.ctypes.data; m.index.type = RTK_TYPE_U16
            vidx = idx.astype(np.uint32)
        elif kind == 2:                                 # u32 indices inside 16-byte records, positions inside 32-byte vertices
            vb = np.zeros(len(verts), dtype=[("pad", "<f4"), ("pos", "<f4", (3,)), ("rest", "<f4", (4,))]); vb["pos"] = verts
            ib = np.zeros(nt, dtype=[("i", "<u4", (3,)), ("mat", "<u4")]); ib["i"] = idx
            keep += [vb, ib]
            m.position.data = vb.ctypes.data + 4; m.position.stride = 32; m.position.type = RTK_TYPE_F32
            m.index.data = ib.ctypes.data; m.index.stride = 16; m.index.type = RTK_TYPE_U32
            vidx = idx.astype(np.uint32)
        else:                                           # callbacks
            v32 = verts.astype(np.float32); i32 = idx.astype(np.uint32)

            def pos_cb(user, mesh, dst, indices, count, v32=v32):
                assert count <= 128
                ii = np.ctypeslib.as_array(indices, shape=(3 * count,))
                np.ctypeslib.as_array(dst, shape=(3 * count, 3))[:] = v32[ii]

            def idx_cb(user, mesh, dst, offset, count, i32=i32):
                np.ctypeslib.as_array(dst, shape=(3 * count,))[:] = i32[offset:offset + count].reshape(-1)
            pcb, icb = POS_CB(pos_cb), IDX_CB(idx_cb); keep += [pcb, icb]
            m.position_cb = C.cast(pcb, C.c_void_p); m.index_cb = C.cast(icb, C.c_void_p)
            vidx = idx.astype(np.uint32)
        meshes.append(m)
        all_tris.append(soup.reshape(-1, 3, 3).astype(np.float32))
        mesh_of.append(np.full(nt, mi, np.uint32)); tri_of.append(np.arange(nt, dtype=np.uint32)); vidx_of.append(vidx)
    arr = (Mesh * len(meshes))(*meshes)
    desc = SceneDesc()
    desc.meshes = C.cast(arr, C.POINTER(Mesh)); desc.num_meshes = len(meshes)
    keep.append(arr)
    return desc, keep, np.concatenate(all_tris), np.concatenate(mesh_of), np.concatenate(tri_of), np.concatenate(vidx_of)
