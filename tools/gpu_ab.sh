#!/bin/bash
# A/B of alternative library builds on the headline bench: tools/gpu_ab.sh libA.so libB.so ...
for lib in "$@"; do
  for rep in 1 2; do
    TOPPRA_HIP_LIB=$PWD/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_EXTRA:-} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$lib', 'traj/s %.3e'%j['value'], 'kernel_ms %.3f'%j['roofline']['kernel_ms'], 'ok', j['ok_fraction'])"
  done
done
