mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r5/pytest_k.log 2>&1; rc=$?; tail -4 gpurun_out/r5/pytest_k.log; echo "pytest rc=$rc"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['without_image_hint']['value'])"
