#!/bin/bash
# round 3: the shadow workload's re-ordering pre-pass at several cell sizes, with and without the direction octant in the key
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=gpurun_out/r3j_sort_shadow.log
: > $L
B="python bench.py --workload shadow --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads"
for bits in 5 6 7 8; do
  for oct in 0 1; do
    echo "== cell bits $bits octant $oct" >> $L; RTK_AMD_SORT_CELL_BITS=$bits RTK_AMD_SORT_OCTANT=$oct timeout -k 10 200 $B >> $L 2>&1 || exit 1
  done
done
python - <<'PY'
import json
for line in open("gpurun_out/r3j_sort_shadow.log"):
    if line.startswith("=="): print(line.strip(), end="")
    elif line.startswith("{"):
        d = json.loads(line); print("   value %.1f Mrays/s  kernel_ms %.3f" % (d["value"], d["roofline"]["kernel_ms"]))
PY
