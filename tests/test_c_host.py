"""The drop-in claim for a C host: examples/host_demo.c, written against rtk.h only, compiles
with gcc and links against librtk_amd.so (CPU); on a GPU box it runs and its per-ray and batch
paths agree."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "host_demo")


EXE2 = os.path.join(ROOT, "examples", "host_tasks")
EXE3 = os.path.join(ROOT, "examples", "host_latency")


def _build():
    if not os.path.exists(os.path.join(ROOT, "rtk_amd", "librtk_amd.so")):
        import __graft_entry__
        __graft_entry__.build()
    for src, exe in (("host_demo.c", EXE), ("host_tasks.c", EXE2), ("host_latency.c", EXE3)):
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "examples", src), "-L" + os.path.join(ROOT, "rtk_amd"), "-lrtk_amd", "-lpthread",
                               "-Wl,-rpath," + os.path.join(ROOT, "rtk_amd"), "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe])


def test_c_host_compiles_and_links():
    _build()
    assert os.path.exists(EXE) and os.path.exists(EXE2) and os.path.exists(EXE3)


@pytest.mark.gpu
def test_c_host_with_its_own_thread_pool_runs():
    """examples/host_tasks.c: the task graph on four pthreads (CPU task builder), the blob traced on the GPU, the same
    rays through the multi-GPU context on a device-built scene, and a host-callback filter."""
    _build()
    r = subprocess.run([EXE2, "100000", "65536"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 differences" in r.stdout and "tasks run on 4 threads" in r.stdout


@pytest.mark.gpu
def test_c_host_runs():
    _build()
    r = subprocess.run([EXE, "5000", "8192"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout


@pytest.mark.gpu
def test_per_ray_calls_from_several_c_threads_agree():
    """examples/host_latency.c: rtk_trace_ray in a loop from 1, 2 and 4 pthreads. Thread k traces the same rays in
    every round it takes part in, so the hit count of n threads is the sum of the first n single-thread counts."""
    _build()
    r = subprocess.run([EXE3, "20000", "300", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("rtk_trace_ray:")]
    assert len(lines) == 3, r.stdout
    hits = [int(l.split(",")[-1].split()[0]) for l in lines]
    assert hits[0] > 0 and hits[1] > hits[0] and hits[2] > hits[1]
