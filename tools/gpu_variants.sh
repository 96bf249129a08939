#!/bin/bash
# parity of every kernel variant on the 7-dof shape + a short bench of each
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from toppra_amd import batch
from oracle import oracle as orc
d = batch.make_synthetic_batch(2048, 7, 200, seed=5)
ref = orc.solve_batch(d["coef"], d["breaks"], d["grid"], d["vlim"], d["alim"], nthreads=0)
for v in (1, 2):
    try:
        got = batch.solve_batch(d["coef"], d["breaks"], d["grid"], d["vlim"], d["alim"], variant=v)
    except Exception as e:
        print("variant", v, "ERR", e); continue
    ok = all(np.array_equal(got[k], ref[k], equal_nan=True) for k in ("K", "sd2", "u", "status"))
    print("variant", v, "bit-exact" if ok else "MISMATCH maxdev %g" % np.nanmax(np.abs(got["sd2"] - ref["sd2"])))
PY
for v in ${VARIANTS:-2}; do
  python bench.py --steps 5 --warmup 2 --variant $v --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('variant', j['config']['kernel_variant'], 'traj/s %.3e'%j['value'], 'kernel_ms %.3f'%j['roofline']['kernel_ms'])"
done
