#!/bin/bash
# round 3: the incoherent workload with the ray re-ordering pre-pass at several cell sizes (is the traversal bound by L2-miss requests?)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=gpurun_out/r3e_sort_incoherent.log
: > $L
B="python bench.py --workload incoherent --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads"
echo "== as given" >> $L; timeout -k 10 200 $B >> $L 2>&1
for bits in 4 5 6 7; do
  echo "== sorted, cell bits $bits" >> $L; RTK_AMD_SORT_CELL_BITS=$bits timeout -k 10 200 $B --sort-rays >> $L 2>&1
done
echo "== sorted, cell bits 6 + octant" >> $L; RTK_AMD_SORT_CELL_BITS=6 RTK_AMD_SORT_OCTANT=1 timeout -k 10 200 $B --sort-rays >> $L 2>&1
echo "== sorted, cell bits 6, origin key" >> $L; RTK_AMD_SORT_CELL_BITS=6 RTK_AMD_SORT_KEY=0 timeout -k 10 200 $B --sort-rays >> $L 2>&1
python - <<'PY'
import json
for line in open("gpurun_out/r3e_sort_incoherent.log"):
    if line.startswith("=="): print(line.strip())
    elif line.startswith("{"):
        d = json.loads(line); print("   value %.1f Mrays/s  kernel_ms %.3f" % (d["value"], d["roofline"]["kernel_ms"]))
PY
