#!/bin/bash
# occupancy experiment: pad dynamic LDS so that 3 / 2 / 1 blocks fit per CU
for pad in 0 20000 15000 ; do
  TPR_LDS_PAD=$pad python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('pad $pad', 'kernel_ms %.3f'%j['roofline']['kernel_ms'])"
done
