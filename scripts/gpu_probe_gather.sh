#!/bin/bash
# gather-rate probe on the GPU box (see scripts/gather_probe.hip); output under gpurun_out/
set -e
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_probe scripts/gather_probe.hip
timeout -k 10 120 /tmp/gather_probe 169 4 256 > gpurun_out/gather_probe_169.log 2>&1
timeout -k 10 180 /tmp/gather_probe 1700 4 256 > gpurun_out/gather_probe_1700.log 2>&1
cat gpurun_out/gather_probe_169.log gpurun_out/gather_probe_1700.log
