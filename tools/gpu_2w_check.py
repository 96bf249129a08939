"""Round 4: kernel family 3 on two waves per block (variant 5) against the one-wave form (variant 3) and the full iteration:
bits and kernel times.   python tools/gpu_2w_check.py [quick]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch as tb
dev = torch.device("cuda", 0)
quick = len(sys.argv) > 1


def to_dev(*arrs):
    return [None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]


def same(a, b, keys=("K", "sd2", "u")):
    bad = a["status"] != b["status"]
    for k in keys:
        eq = (a[k] == b[k]) | (torch.isnan(a[k]) & torch.isnan(b[k]))
        bad |= ~eq.reshape(eq.shape[0], -1).all(dim=1)
    return int(bad.sum().item())


total = 0
for B, d, N in ((65536, 7, 200), (131072, 7, 200), (262144, 7, 200), (65536, 6, 500), (65536, 4, 100), (16384, 7, 200), (65536, 8, 100)):
    data = tb.make_synthetic_batch(B, d, N)
    dv = to_dev(*(data[k] for k in ("coef", "breaks", "grid", "vlim", "alim")))
    ref = tb.solve_batch(*dv, variant=3)
    line = "B %6d d %d N %3d:" % (B, d, N)
    for variant in (3, 5):
        out = tb.solve_batch(*dv, variant=variant)
        ms = tb.solve_batch_timed(*dv, out, 10, variant=variant)
        bad = same(out, ref)
        total += bad
        line += "  variant %d %.3f ms (%.1f M traj/s), %d differ" % (variant, ms, B / ms / 1e3, bad)
    print(line, flush=True)
rng = np.random.default_rng(7)
for d, N, B in ((7, 120, 16384), (3, 60, 16384), (8, 64, 8192), (5, 90, 16384), (1, 50, 8192), (2, 33, 5000)):
    data = tb.make_synthetic_batch(B, d, N, seed=400 + d)
    base = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], None, None)
    cases = [("natural", base, True),
             ("scaled", (data["coef"] * 10.0 ** rng.uniform(-6, 0, size=(B, 1, 1, 1)),) + base[1:], True),
             ("tight", base[:3] + (data["vlim"] * 10.0 ** rng.uniform(-1.5, -0.3, size=(B, 1, 1)),) + base[4:], True),
             ("boundary", base[:5] + (np.where(rng.random(B) < 0.5, 0.2 * rng.random(B), 0.0), np.where(rng.random(B) < 0.7, 0.5 * rng.random(B), 0.0)), True),
             ("collocation", base, False), ("acc_only", base[:3] + (None,) + base[4:], True)]
    for name, args, interp in cases:
        dv = to_dev(*args)
        full = tb.solve_batch(*dv, interpolation=interp, strict=True)
        out = tb.solve_batch(*dv, interpolation=interp, variant=5)
        bad = same(out, full)
        # controllable sets alone, and with sd as an output
        Kc = tb.controllable_sets_batch(*dv[:5], np.zeros(B), 0.3 * np.ones(B), interp, variant=5)
        Kf = tb.controllable_sets_batch(*dv[:5], np.zeros(B), 0.3 * np.ones(B), interp, strict=True)
        badk = int((~((Kc == Kf) | (torch.isnan(Kc) & torch.isnan(Kf))).reshape(B, -1).all(dim=1)).sum().item())
        sdo = tb.solve_batch(*dv, interpolation=interp, variant=5, want_sd=True)
        bads = same(sdo, full) + int((~((sdo["sd"] == torch.sqrt(full["sd2"])) | (torch.isnan(sdo["sd"]) & torch.isnan(full["sd2"])))).any(dim=1).sum().item())
        total += bad + badk + bads
        print("%-11s d %d N %3d B %5d ok %.3f: variant 5 %d differ, controllable sets %d differ, with sd %d differ" % (
            name, d, N, B, float((full["status"] == 0).double().mean()), bad, badk, bads), flush=True)
print("TOTAL differing: %d" % total)
