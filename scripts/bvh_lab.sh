#!/bin/bash
# Tree-quality lab (CPU only, no GPU, no oracle): builds BVHs of the config-2 scene with several builders and counts
# node / leaf / triangle visits of an ordered closest-hit traversal on samples of the config-2 and config-3 rays.
# Calibration: "lbvh -w 4" reproduces the device build's node count (462584) and visit counts (21.9 / 28.3).
#   bash scripts/bvh_lab.sh > profiles/r02_tree_quality_lab.log
set -e
cd "$(dirname "$0")/.."
W=${TMPDIR:-/tmp}/bvh_lab; mkdir -p $W
python3 - <<PY
import sys; sys.path.insert(0, '.')
import numpy as np
from rtk_amd import synth
synth.scene_for_config(2).tofile('$W/tris_1m.f32')
r = synth.rays_pinhole(4096, 4096)
np.ascontiguousarray(r[np.arange(0, 4096 * 4096, 257)]).tofile('$W/rays_coh.bin')
synth.rays_incoherent(65536).tofile('$W/rays_inc.bin')
PY
g++ -O2 -fopenmp -o $W/lab scripts/bvh_lab.cpp
cd $W
echo "# builder sweep at one triangle per leaf (cn 0.5): LBVH vs full binned SAH vs LBVH with SAH-rebuilt subtrees vs PLOC"
./lab -b lbvh; ./lab -b sah; ./lab -b hyb -T 256; ./lab -b hyb -T 4096; ./lab -b ploc -r 4 2>/dev/null; ./lab -b ploc -r 16 2>/dev/null
echo "# leaf size trade (what the CPU oracle's 16.6 node visits are made of): SAH with larger leaves"
./lab -b sah -cn 1 -ml 4; ./lab -b sah -cn 2 -ml 4; ./lab -b lbvh -cn 2 -ml 8
echo "# collapse rule: greedy largest-area (cm 0) vs fixed two binary levels (cm 1, the reference's rtk.c:1572-1592)"
./lab -b lbvh -cm 1; ./lab -b sah -cm 1
echo "# Morton key width: 63, 48, 42, 30 bits"
./lab -b lbvh -ks 0; ./lab -b lbvh -ks 15; ./lab -b lbvh -ks 21; ./lab -b lbvh -ks 33
echo "# node width: 4, 6, 8 children (full distance sort), 8 with nearest-first-then-slot-order"
./lab -b lbvh -w 4; ./lab -b lbvh -w 6; ./lab -b lbvh -w 8; ./lab -b lbvh -w 8 -om 1
echo "# SAH treelets: every maximal LBVH subtree of <= T triangles rebuilt by binned SAH (T = 8 .. 256; 8 bins at 64 and 32)"
./lab -b hyb -T 8; ./lab -b hyb -T 16; ./lab -b hyb -T 32; ./lab -b hyb -T 64; ./lab -b hyb -T 64 -bins 8; ./lab -b hyb -T 32 -bins 8
echo "# split position chosen by SAH along the Morton order (no re-ordering of triangles) inside subtrees of <= T"
./lab -b mswp -T 16; ./lab -b mswp -T 64; ./lab -b mswp -T 256; ./lab -b mswp -T 1024
echo "# fixed groups of T consecutive sorted triangles rebuilt by SAH, radix tree over the group borders above (groups straddle Morton jumps)"
./lab -b grp -T 32 -bins 8; ./lab -b grp -T 64 -bins 8; ./lab -b grp -T 128 -bins 8
echo "# which open child the greedy collapse opens next: area x triangle count, area saved by opening, triangle count (largest area = the first line of this log)"
./lab -b lbvh -cc 1; ./lab -b lbvh -cc 2; ./lab -b lbvh -cc 3
echo "# what one wave could do for a 64-triangle treelet: at most 8 bins and never more than items; the same binned over the node's box instead of its centroid bounds"
./lab -b hyb -T 64 -bins -8; ./lab -b hyb -T 64 -bins -8 -bob 1
