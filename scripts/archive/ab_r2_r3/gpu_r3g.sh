#!/bin/bash
# round 3: the ray-pool kernel against the bound-lanes kernel on the incoherent and shadow workloads
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=gpurun_out/r3g_pool.log
: > $L
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads"
for wl in incoherent shadow; do
  for pool in 1 0; do
    echo "== $wl pool=$pool" >> $L; RTK_AMD_POOL=$pool timeout -k 10 200 $B --workload $wl >> $L 2>&1
    echo "== $wl pool=$pool sorted(bits 6)" >> $L; RTK_AMD_SORT_CELL_BITS=6 RTK_AMD_POOL=$pool timeout -k 10 200 $B --workload $wl --sort-rays >> $L 2>&1
  done
done
python - <<'PY'
import json
for line in open("gpurun_out/r3g_pool.log"):
    if line.startswith("=="): print(line.strip())
    elif line.startswith("{"):
        d = json.loads(line); print("   value %.1f Mrays/s  kernel_ms %.3f  parity %s" % (d["value"], d["roofline"]["kernel_ms"], json.dumps(d.get("parity"))[:200]))
    elif "Error" in line or "error" in line: print("   !!", line.strip()[:200])
PY
