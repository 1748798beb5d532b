# round 5: what rtk_packet_beam2's frame time is sensitive to: idle vector instructions per triangle test, idle scalar instructions per
# node step (variant libraries built with PACKET_ASMFLAGS=-DEXP_VALU_TRI=n / -DEXP_SALU_NODE=n), against the library as it is
run() { RTK_AMD_LIB=$1 timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-workloads 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms'])" || exit 1; }
run "" base
for v in "$@"; do run $PWD/variants/libs/librtk_$v.so $v; done
run "" base
