#!/bin/bash
# round 3: refit pass 2 on compact climber lists with one compare-and-swap per meeting: build tests, then build times
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_build.py tests/test_gpu_api_rows.py tests/test_gpu_trace.py -x -q -m gpu > gpurun_out/r3m_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r3m_pytest.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/build_timing.py 2>&1 | grep -v amdgpu.ids
