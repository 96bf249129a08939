#!/usr/bin/env python
"""What bit-exactness costs ONCE SOUNDNESS IS KEPT: the second measurement build
(python -m toppra_amd.build --sound-tolerance: -DTPR_SOUND_TOLERANCE -- the product's trace-following certificates returning
the verified vertex from a reciprocal estimate instead of replicating the reference's last pivot, reciprocal quotients in
the forward 1-variable LP; the cooperative batches' full iteration untouched) against the PRODUCT on the headline batch and
on the adversarial families: kernel time, max |d sd^2|, |d K|, |d u|, and whether every status code / NaN pattern is identical.

    python tools/gpu_sound_tolerance_report.py        # parent = product library; a child process loads the other one
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "toppra_amd", "libtoppra_hip_stol.so")
TMP = "/tmp/tpr_stol"


def cases():
    from toppra_amd import batch
    import gpu_sliver_hunt as sliver
    import gpu_tolerance_report as tr
    out = {}
    data = batch.make_synthetic_batch(65536, 7, 200)
    out["headline_65536x7x200"] = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], None, None)
    out["irregular_65536x7x200"] = tr.irregular(65536, 7, 200, 5, 1000)
    rng = np.random.default_rng(77)
    d2 = batch.make_synthetic_batch(32768, 7, 200, seed=78)
    out["scaled_32768x7x200"] = (d2["coef"] * 10.0 ** rng.uniform(-6, 0, size=(32768, 1, 1, 1)), d2["breaks"], d2["grid"], d2["vlim"], d2["alim"], None, None)
    for B, d, N, seed in ((32768, 7, 60, 201), (32768, 4, 50, 202)):
        args, _ = sliver.family(B, d, N, seed)
        out["sliver_%dx%dx%d" % (B, d, N)] = args
    # near-parallel joints (tests/test_gpu_fullsize.py::test_near_parallel_rows_are_bit_exact)
    B, d, N = 32768, 7, 120
    rng = np.random.default_rng(901)
    way = rng.standard_normal((B, 5, d))
    eps = 10.0 ** rng.uniform(-14, -6, size=B)
    scale = rng.choice([1.0, -1.0, 0.5, 2.0, 3.0], size=B)
    src, dst = rng.integers(0, d, size=B), rng.integers(0, d, size=B)
    dst = np.where(dst == src, (src + 1) % d, dst)
    rows = np.arange(B)
    way[rows, :, dst] = way[rows, :, src] * (scale * (1 + eps))[:, None]
    coef, breaks = batch.spline_coefficients(np.linspace(0, 1, 5), way)
    vmax = 10 + 20 * rng.random((B, d)); amax = 10 + 2 * rng.random((B, d))
    amax[rows, dst] = amax[rows, src] * np.abs(scale) * (1 + 0.02 * rng.standard_normal(B))
    vmax[rows, dst] = vmax[rows, src] * np.abs(scale) * (1 + 0.02 * rng.standard_normal(B))
    out["near_parallel_%dx%dx%d" % (B, d, N)] = (coef, breaks, np.linspace(0, 1, N + 1), np.ascontiguousarray(np.stack([-vmax, vmax], -1)),
                                                 np.ascontiguousarray(np.stack([-amax, amax], -1)), None, None)
    return out


def solve_all(tag, cs):
    import torch
    from toppra_amd import batch
    dev = torch.device("cuda", 0)
    res = {}
    for name, args in cs.items():
        dv = [None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in args]
        o = batch.solve_batch(*dv, variant=3)
        torch.cuda.synchronize()
        ms = batch.solve_batch_timed(*dv[:5], o, reps=5, sd_start=dv[5], sd_end=dv[6], variant=3)
        np.savez("%s_%s_%s.npz" % (TMP, tag, name), **{k: v.cpu().numpy() for k, v in o.items() if v is not None})
        res[name] = ms
    return res


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        cs = {k[:-4].split("_in_", 1)[1]: None for k in os.listdir("/tmp") if k.startswith("tpr_stol_in_")}
        loaded = {}
        for name in cs:
            z = np.load("/tmp/tpr_stol_in_%s.npz" % name, allow_pickle=True)
            loaded[name] = tuple(None if z["a%d" % i].dtype == object else z["a%d" % i] for i in range(7))
        print(json.dumps(solve_all("stol", loaded)))
        return
    if not os.path.exists(LIB):
        sys.exit("build the library first: python -m toppra_amd.build --sound-tolerance")
    cs = cases()
    for name, args in cs.items():
        np.savez("/tmp/tpr_stol_in_%s.npz" % name, **{"a%d" % i: (np.array(None, dtype=object) if a is None else a) for i, a in enumerate(args)})
    ms_p = solve_all("product", cs)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, TOPPRA_HIP_LIB=LIB), cwd=ROOT, capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        sys.exit("child failed: " + out.stderr[-800:])
    ms_s = json.loads(lines[-1])
    rep = {}
    for name in cs:
        a = dict(np.load("%s_product_%s.npz" % (TMP, name))); b = dict(np.load("%s_stol_%s.npz" % (TMP, name)))
        dmax = lambda k: float(np.nanmax(np.abs(a[k] - b[k]))) if np.isfinite(a[k]).any() else 0.0
        rep[name] = {"product_kernel_ms": round(ms_p[name], 4), "sound_tolerance_kernel_ms": round(ms_s[name], 4),
                     "status_identical": bool(np.array_equal(a["status"], b["status"])),
                     "nan_pattern_identical": bool(all(np.array_equal(np.isnan(a[k]), np.isnan(b[k])) for k in ("K", "sd2", "u"))),
                     "max_abs_dsd2": dmax("sd2"), "max_abs_dK": dmax("K"), "max_abs_du": dmax("u"),
                     "ok_fraction": float((a["status"] == 0).mean())}
    h = rep["headline_65536x7x200"]
    rep["summary"] = {"statuses_identical_everywhere": bool(all(r["status_identical"] and r["nan_pattern_identical"] for r in rep.values())),
                      "worst_max_abs_dsd2": max(r["max_abs_dsd2"] for r in rep.values()),
                      "bit_exactness_costs_with_soundness_kept": "%.1f %% of the headline kernel time" % (100 * (1 - h["sound_tolerance_kernel_ms"] / h["product_kernel_ms"]))}
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
