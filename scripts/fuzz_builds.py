"""One-off soak: many random degenerate scenes through the device build, the validator, the blob round trip and the
oracle on the same BVH (tests/test_gpu_sizes.py runs 24 seeds of this; this script runs as many as asked).
Usage: python scripts/fuzz_builds.py [first_seed] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import pyoracle  # noqa: E402
from rtk_amd import api, synth  # noqa: E402
from tests.test_gpu_sizes import _degenerate_mix  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
ties = 0      # rays whose two answers are the same hit up to rounding (duplicates, flat boxes: order decides)
rays = synth.rays_exotic(2048)
plain = synth.rays_config1(2048)
for seed in range(first, first + count):
    tris = _degenerate_mix(seed)
    ds = api.DeviceScene.build([dict(positions=tris)])
    ok, c = ds.validate()
    blob = pyoracle.Blob(ds.export_blob())
    problems = []
    if not ok or c["loose_boxes"]:
        problems.append(("validate", c))
    if pyoracle.validate_blob(blob)[0] != 0:
        problems.append("blob")
    for name, r in (("plain", plain), ("exotic", rays)):
        oh, om = pyoracle.trace(blob, r)
        for opts in (None, api.make_opts(image=(32, 64)), api.make_opts(exact_nodes=True)):
            rec = ds.trace(r, opts=opts, full=False)
            gm = rec["prim"] != 0xFFFFFFFF
            if not (gm == om).all():
                problems.append((name, "mask", int((gm != om).sum())))
            elif not ((rec["prim"][gm] == oh["triangle_index"][om]).all() and (rec["t"][gm] == oh["t"][om]).all()):
                d = (rec["prim"][gm] != oh["triangle_index"][om]) | (rec["t"][gm] != oh["t"][om])
                ta, tb = rec["t"][gm][d].astype(np.float64), oh["t"][om][d].astype(np.float64)
                rel = np.abs(ta - tb) / np.maximum(np.abs(tb), 1e-30)
                # farther than the oracle by more than rounding = a lost hit; closer = the oracle lost one (both happen where
                # boxes are flat: DESIGN.md section 4, "where parity is undefined")
                ties += int(d.sum())
                if rel.max() > 1e-5:
                    problems.append((name, "t differs", int(d.sum()), "max rel dt %.2e" % rel.max(), "gpu farther: %d" % int((ta > tb * (1 + 1e-5)).sum()),
                                     "gpu closer: %d" % int((ta < tb * (1 - 1e-5)).sum())))
    if api.lib().rtk_dev_trace_status(ds.handle, None) != 0:
        problems.append("trace status")
    if problems:
        bad += 1
        print("seed", seed, "n", len(tris) // 3, problems[:4], flush=True)
    ds.free()
print("fuzz_builds: %d seeds, %d with problems; %d ray answers differ within rounding (ties among duplicate / flat geometry)" % (count, bad, ties), flush=True)
