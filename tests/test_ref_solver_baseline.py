"""The reference's own compiled seidel solver (oracle/_ref) as bench.py's second CPU baseline: loaded under stand-in
parent modules (as on the GPU box, where /root/reference does not exist) and driven by the restated passes, it must give
the oracle's bits -- which pins the oracle once more, against the reference's compiled code run end to end."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from toppra_amd import batch
from oracle import oracle as orc, ref_solver_baseline as rb
assert "toppra" not in sys.modules
out = {}
for d, N, seed, scale in ((7, 60, 3, 1.0), (3, 25, 4, 1e-3), (12, 30, 5, 1.0)):
    data = batch.make_synthetic_batch(5, d, N, seed=seed)
    coef = data["coef"] * scale
    want = orc.solve_batch(coef, data["breaks"], data["grid"], data["vlim"], data["alim"])
    same = True
    for k in range(5):
        vel, acc = rb.constraint_tuples(coef[k], data["breaks"], data["grid"], data["vlim"][k], data["alim"][k])
        w = rb.make_wrapper([rb.PrecomputedConstraint(vel, False), rb.PrecomputedConstraint(acc, True)], None, data["grid"])
        sdd, sd, K = rb.parameterization(w, 0.0, 0.0)
        if want["status"][k] == 0:
            same &= bool(np.array_equal(sd, np.sqrt(want["sd2"][k])) and np.array_equal(sdd, want["u"][k]) and np.array_equal(K, want["K"][k]))
        else:
            same &= sd is None or bool(np.isnan(sd).any())
    out["d%%d" %% d] = same
out["standin"] = type(sys.modules["toppra"]).__name__ == "module" and not hasattr(sys.modules["toppra"], "__file__")
r = rb.time_passes(batch.make_synthetic_batch(8, 7, 40), 8, 2)
out["pool_ok"] = r["ok"] == 8 and r["processes"] == 2
print(json.dumps(out))
""" % ROOT


def test_compiled_reference_solver_under_standin_modules_gives_the_oracles_bits():
    from oracle import ref_solver_baseline as rb
    if not rb.available():
        pytest.skip("oracle/_ref holds no compiled reference solver (built where /root/reference exists)")
    env = dict(os.environ, TPR_REF_FORCE_STANDIN="1")
    pr = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert pr.returncode == 0, pr.stderr[-2000:]
    out = json.loads(pr.stdout.strip().splitlines()[-1])
    assert out == {"d7": True, "d3": True, "d12": True, "standin": True, "pool_ok": True}, out
