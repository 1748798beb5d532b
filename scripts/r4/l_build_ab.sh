#!/bin/bash
# round 4: build time A/B: key width (radix passes) and where the tile collapse starts
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout -k 10 200 python scripts/build_timing.py 1000000 10000000 2>&1 | grep "device-resident"; }
run RTK_AMD_KEY_BITS=40
run X=1
run RTK_AMD_TILE_COLLAPSE_MIN=500000
run RTK_AMD_KEY_BITS=32
timeout -k 10 300 python -m pytest tests/test_gpu_build.py -m gpu -x -q 2>&1 | tail -3
