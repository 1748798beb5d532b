"""Soak of the device build's cross-tile pass (k_refit_top: subtrees of different tiles meet through one compare-and-swap):
many sizes and seeds, every build validated on the device, built twice (the two builds must be byte-identical: content hash),
clustered inputs (duplicate Morton codes make deep chains across tile borders) next to uniform ones.
Usage: python scripts/soak_refit.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from rtk_amd import api, synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sizes = [1025, 2047, 2049, 4097, 10_000, 65_537, 300_000, 1_000_000, 2_500_000, 5_000_000]
bad = 0
for rnd in range(rounds):
    for n in sizes:
        for kind in ("uniform", "clustered", "line"):
            seed = 1000 * rnd + n % 997
            tris = synth.triangle_soup(n, 0.02, seed).reshape(n, 3, 3)
            if kind == "clustered":
                # a handful of tight clusters: long runs of equal Morton codes, ties broken by index
                c = (synth.u01(seed + 1, 0, 24).reshape(8, 3))[np.arange(n) % 8]
                tris = (c[:, None, :] + (tris - tris.mean(axis=1, keepdims=True)) * np.float32(1e-4)).astype(np.float32)
            elif kind == "line":
                tris = tris.copy()
                tris[:, :, 1:] *= np.float32(1e-6)
            tris = np.ascontiguousarray(tris.reshape(n * 3, 3))
            hashes = []
            for rep in range(2):
                ds = api.DeviceScene.build([dict(positions=tris)])
                ok, c = ds.validate()
                if not ok or c.get("loose_boxes"):
                    print("INVALID", n, kind, seed, c)
                    bad += 1
                hashes.append(c.get("content_hash"))
                ds.free()
            if hashes[0] != hashes[1]:
                print("NOT DETERMINISTIC", n, kind, seed, hashes)
                bad += 1
    print("round", rnd, "done, problems so far:", bad, flush=True)
sys.exit(1 if bad else 0)
