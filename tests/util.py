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
