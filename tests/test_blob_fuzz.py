"""Fuzz of the blob loader (rtk_dev_scene_upload_buffer's host-side validator, rtk_upload.hip): random corruptions of
valid blobs must be refused or accepted, never crash or read outside the buffer. Runs in a child process (a crash must
not take pytest down) with the blob placed at the END of a page-aligned mapping followed by an unreadable guard page,
so that any read past the declared size faults."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, mmap, sys
import numpy as np
sys.path.insert(0, %(root)r)
from rtk_amd import api, synth
from oracle import pyoracle
L = api.lib()
libc = C.CDLL(None)
libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
rng = np.random.RandomState(%(seed)d)
blobs = [pyoracle.build_scene([dict(positions=synth.triangle_soup(n, 0.3, seed=7 + n))]).data.copy() for n in (1, 5, 64, 700)]
PAGE = mmap.PAGESIZE
refused = accepted = 0
for it in range(%(iters)d):
    good = blobs[it %% len(blobs)]
    size = good.size
    span = (size + PAGE - 1) // PAGE * PAGE
    mm = mmap.mmap(-1, span + PAGE)
    base = C.addressof(C.c_char.from_buffer(mm))
    assert libc.mprotect(C.c_void_p(base + span), PAGE, 0) == 0          # guard page: PROT_NONE
    start = base + span - size                                            # the blob ends exactly at the guard page
    bad = good.copy()
    mode = it %% 4
    if mode == 0:                                                         # a few random bytes
        for _ in range(rng.randint(1, 6)):
            bad[rng.randint(0, size)] = rng.randint(0, 256)
    elif mode == 1:                                                       # a random 64-bit word replaced by an extreme value
        w = rng.randint(0, size // 8)
        bad[8 * w:8 * w + 8] = np.frombuffer(np.uint64(rng.choice([0, 1, 127, 128, 2**32, 2**63, 2**64 - 1, 2**64 - 128, size, size - 1, size + 1])).tobytes(), np.uint8)
    elif mode == 2:                                                       # a pointer inside the node section set to another plausible offset
        hdr_leaf = int(good[40:48].view(np.uint64)[0])
        w = rng.randint((128 + 96) // 8, max(hdr_leaf // 8, (128 + 96) // 8 + 1))
        bad[8 * w:8 * w + 8] = np.frombuffer(np.uint64(rng.randint(0, size) & ~rng.choice([0, 1, 63, 127])).tobytes(), np.uint8)
    else:                                                                 # truncated: the buffer is shorter than the header says
        pass
    C.memmove(start, bad.ctypes.data, size)
    avail = size if mode != 3 else rng.randint(0, size)
    src = start if mode != 3 else base + span - avail
    if mode == 3 and avail:
        C.memmove(src, bad.ctypes.data, avail)
    h = L.rtk_dev_scene_upload_buffer(C.c_void_p(src), avail)
    err = api.last_error()
    if h:
        accepted += 1
        L.rtk_dev_scene_free(C.c_void_p(h))
    elif "HIP device" in err:
        accepted += 1                                                     # passed validation; only the (absent) GPU stopped it
    else:
        refused += 1
    del bad
    mm.close()
print("fuzz ok refused=%%d accepted=%%d" %% (refused, accepted))
'''


def test_blob_loader_survives_random_corruption(oracle, api):
    code = CHILD % dict(root=ROOT, seed=1234, iters=1600)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, "loader crashed or hung: rc=%d\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    refused = int(r.stdout.split("refused=")[1].split()[0])
    assert refused > 400                                               # most corruptions of pointers / headers are caught
