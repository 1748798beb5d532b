"""The drop-in claim for a C host: examples/host_demo.c, written against rtk.h only, compiles
with gcc and links against librtk_amd.so (CPU); on a GPU box it runs and its per-ray and batch
paths agree."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "host_demo")


def _build():
    if not os.path.exists(os.path.join(ROOT, "rtk_amd", "librtk_amd.so")):
        import __graft_entry__
        __graft_entry__.build()
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "host_demo.c"), "-L" + os.path.join(ROOT, "rtk_amd"), "-lrtk_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "rtk_amd"), "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", EXE])


def test_c_host_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c_host_runs():
    _build()
    r = subprocess.run([EXE, "5000", "8192"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout
