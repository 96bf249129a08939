#!/bin/bash
python tools/gpu_param_pcr_check.py 2>&1 | grep -v amdgpu.ids | grep -E "^time|False"
python -m pytest tests/test_gpu_param.py tests/test_gpu_golden.py -x -q 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(json.dumps(j['end_to_end'], indent=None))"
