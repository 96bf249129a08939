"""The lane-level certificates of kernel family 3 -- the PRODUCT's source, toppra_amd/csrc/tpr_cert_lane.hip.inc, compiled for
the host through tests/host_cert/hip_shim -- against the CPU restatement of the reference, stage LP by stage LP
(tests/host_cert/host_cert.cpp; driver: tools/host_cert_hunt.py).  Every certified answer must be the reference's bits (u, x,
active pair), and no certificate may answer an LP on which the reference fails: the property "the default mode is sound"
that the GPU can only show on whole trajectories is checked here per LP, on adversarial families included.  No GPU needed."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("host_cert_hunt", os.path.join(ROOT, "tools", "host_cert_hunt.py"))
hunt = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(hunt)


@pytest.fixture(scope="module")
def harness():
    return hunt.build()


@pytest.mark.parametrize("family,seed,B,min_upper,min_lower", [
    ("natural", 0, 40, 0.99, 0.999),     # 7 dof, N = 200: the benchmark's batch
    ("natural", 9, 24, 0.95, 0.999),     # 12 dof: slim blocks, block stride 16 of the row numbering
    ("tight", 1, 48, 0.95, 0.99),        # velocity limits that bind: cold starts, slides along the box row of x
    ("boundary", 2, 48, 0.97, 0.9),      # non-zero end velocities
    ("collocation", 3, 48, 0.97, 0.999),
    ("scaled", 5, 48, 0.3, 0.3),         # paths scaled by 1e-6 .. 1: most stages fail the row-norm guard -- and must not be answered
    ("sliver", 0, 96, 0.5, 0.5),         # three rows through one point to 1e-8 .. 1e-13 (the reference itself fails on many)
    ("parallel", 4, 48, 0.5, 0.5),       # two joints parallel to 1e-6 .. 1e-14
    ("lower_ties", 5, 48, 0.9, 0.9),     # near-ties between prefix records of the lower-bound LP's run
])
def test_certificates_return_the_references_bits(harness, family, seed, B, min_upper, min_lower):
    (coef, breaks, grid, vlim, alim, sd_end, flags, mode), _ = hunt.workloads(family, B, seed)
    res, rc = hunt.run(coef, breaks, grid, vlim, alim, sd_end, flags, mode)
    assert res["mismatch"] == 0 and res["ref_failed_cert_answered"] == 0 and rc == 0, res
    assert res["upper_cert"] >= min_upper * res["upper"] and res["lower_cert"] >= min_lower * res["lower"], res
    if family in ("natural", "tight", "collocation"):  # the equality stage (the first backward stage) is certified too
        assert res["eq_upper_cert"] + res["eq_lower_cert"] >= 1.8 * res["eq_upper"], res


@pytest.mark.parametrize("family,seed,B", [("natural", 0, 24), ("sliver", 0, 64), ("lower_ties", 5, 48), ("feasible", 0, 24)])
def test_the_minimum_form_of_the_slides(harness, family, seed, B):
    """The slides fold their verdicts into sign bits (every kernel but TOPPRAsd's) or into a running minimum (TOPPRAsd's
    instantiations: tpr_cert_lane.hip.inc, cert_slide_visit): the tests above run the sign-bit form, this one the other."""
    (coef, breaks, grid, vlim, alim, sd_end, flags, mode), _ = hunt.workloads(family, B, seed)
    res, rc = hunt.run(coef, breaks, grid, vlim, alim, sd_end, flags, mode, minform=True)
    assert res["mismatch"] == 0 and res["ref_failed_cert_answered"] == 0 and rc == 0, res
    both, _ = hunt.run(coef, breaks, grid, vlim, alim, sd_end, flags, mode)
    for k in ("upper_cert", "lower_cert", "feas_upper_cert", "feas_lower_cert"):
        assert abs(both[k] - res[k]) <= 2, (k, both[k], res[k])  # (an exact tie at a tolerance boundary is the only difference)


@pytest.mark.parametrize("seed", [0, 2])
def test_feasible_set_certificates(harness, seed):
    (coef, breaks, grid, vlim, alim, sd_end, flags, mode), _ = hunt.workloads("feasible", 32, seed)
    res, rc = hunt.run(coef, breaks, grid, vlim, alim, sd_end, flags, mode)
    assert res["mismatch"] == 0 and rc == 0, res
    assert res["feas_upper_cert"] >= 0.97 * res["feas_upper"] and res["feas_lower_cert"] >= 0.99 * res["feas_lower"], res


def test_the_round3_fast_certificates_are_caught_when_they_answer_too_much(harness):
    """Control: the harness does distinguish.  With the trace checks switched off (`legacy`: the round-2/3 certificates, which
    bound the reference's last pivot only) the acceptance on the moved pairs is higher -- what the sound mode refuses is what
    the fast mode answered without being able to vouch for the reference's earlier pivots."""
    (coef, breaks, grid, vlim, alim, sd_end, flags, mode), _ = hunt.workloads("natural", 32, 0)
    sound, _ = hunt.run(coef, breaks, grid, vlim, alim, sd_end, flags, mode)
    fast, _ = hunt.run(coef, breaks, grid, vlim, alim, sd_end, flags, mode, legacy=True)
    assert fast["upper_moved_cert"] > sound["upper_moved_cert"] and fast["mismatch"] == 0 and sound["mismatch"] == 0
