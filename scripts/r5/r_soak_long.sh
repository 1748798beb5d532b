# round 5: longer soak of the two-stream build (every size / kind built twice per round, hashes must agree; degenerate mixes with the
# tile collapse forced; the 10M build ten times with one hash)
mkdir -p gpurun_out/r5
timeout -k 10 900 python scripts/soak_refit.py 6 > gpurun_out/r5/soak_refit_long.log 2>&1; rc=$?; tail -2 gpurun_out/r5/soak_refit_long.log; echo "soak rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
RTK_AMD_TILE_COLLAPSE_MIN=0 timeout -k 10 900 python scripts/fuzz_builds.py 1000 400 > gpurun_out/r5/fuzz_tile_long.log 2>&1; rc=$?; tail -2 gpurun_out/r5/fuzz_tile_long.log; echo "fuzz rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python - <<'PY'
import sys
sys.path.insert(0, '.')
from rtk_amd import api, synth
tris = synth.scene_for_config(5)
hashes = set()
for i in range(10):
    ds = api.DeviceScene.build([dict(positions=tris)])
    ok, c = ds.validate()
    assert ok and c["loose_boxes"] == 0, c
    hashes.add(c["content_hash"])
    ds.free()
print("10M build x10: hashes", len(hashes), "nodes", c["nodes"] if "nodes" in c else "")
assert len(hashes) == 1
PY
