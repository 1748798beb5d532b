"""Device build at sizes around every internal boundary (sort tile 4096, refit tile 1024, collapse levels of 1024 jobs,
256-job blocks): structure validates, two builds agree, and the traversal matches the oracle on the exported blob."""
import numpy as np
import pytest

from rtk_amd import synth

pytestmark = pytest.mark.gpu

SIZES = [2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 2049, 4095, 4096, 4097, 12289, 65535, 65536, 65537, 262145]


@pytest.mark.parametrize("n", SIZES)
def test_build_at_boundary_sizes(api, oracle, n):
    tris = synth.triangle_soup(n, 0.2 if n < 100 else 0.05, seed=100 + n % 97)
    ds = api.DeviceScene.build([dict(positions=tris)])
    ok, c = ds.validate()
    assert ok and c["loose_boxes"] == 0, (n, c)
    assert c["triangles_checked"] == n
    assert api.DeviceScene.build([dict(positions=tris)]).validate()[1]["content_hash"] == c["content_hash"]
    rays = synth.rays_config1(1024)
    blob = oracle.Blob(ds.export_blob())
    assert oracle.validate_blob(blob)[0] == 0
    oh, om = oracle.trace(blob, rays)
    for opts in (None, api.make_opts(image=(32, 32))):
        rec = ds.trace(rays, opts=opts, full=False)
        gm = rec["prim"] != 0xFFFFFFFF
        assert (gm == om).all() and (rec["prim"][gm] == oh["triangle_index"][om]).all(), n
        assert (rec["t"][gm] == oh["t"][om]).all() and (rec["u"][gm] == oh["u"][om]).all() and (rec["v"][gm] == oh["v"][om]).all(), n
