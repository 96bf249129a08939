#!/usr/bin/env python
"""Why the batches' simplex walk gives up (counters of a -DTPR_DEBUG_PREDICT build)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch as tb, _capi
names = ["no warm pair / empty box", "pair out of range", "|det| <= 1e-7", "both multipliers <= 0", "primal step: iteration cap",
         "primal step: unbounded", "residual within tolerance band (no violated row)", "dual pivot: iteration cap",
         "dual pivot: no leaving row", "two box rows", "guard: near-parallel row", "guard: short row", "guard: |v1d|", "guard: violation bound"]
B, d, N = 65536, 7, 200
data = tb.make_synthetic_batch(B, d, N)
L = _capi.load()
buf = (C.c_ulonglong * 16)()
tb.solve_batch(data["coef"][:64], data["breaks"], data["grid"], data["vlim"][:64], data["alim"][:64], variant=3)
L.tpr_debug_walk_fail(buf)
tb.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], variant=3)
L.tpr_debug_walk_fail(buf)
tot = sum(buf)
print("walk give-ups per trajectory: %.4f" % (tot / B))
for n, v in zip(names, buf):
    if v: print("  %-55s %9d  %.4f per trajectory" % (n, v, v / B))
h = (C.c_uint * 2560)()
L.tpr_debug_walk_hist(h)
h = np.array(h[:]).reshape(5, 512)
for name, row in zip(("third row (virtual index: 0-3 box, 4/5 x_next, 6+blk*D+k)", "p", "q"), h[:3]):
    nz = np.flatnonzero(row)
    print(name, {int(k): int(row[k]) for k in nz})
st = h[3][:N]
print("by stage (bins of 10):", st.reshape(-1, 10).sum(1).tolist())
print("-log10|residual of the third row| histogram (bin 0: exactly 0):", {int(k): int(h[4][k]) for k in np.flatnonzero(h[4])})
t = h[3][256:256 + 64]
names3 = ["x_next+row", "box high1+row", "two acceleration rows"]
print("upper-bound LP active-pair transitions (per trajectory):")
for needy in (0, 1):
    for same in (0, 1):
        for a in range(3):
            for b in range(3):
                v = t[needy * 32 + same * 16 + a * 4 + b]
                if v: print("  %s %s  %-22s -> %-22s %8.3f" % ("needy " if needy else "lane  ", "same pair" if same else "new pair ", names3[a], names3[b], v / B))
why = h[4][100:108]
print("why the lane-level certificate of the PROPOSED pair failed (per trajectory):",
      {n: round(int(v) / B, 3) for n, v in zip(["-", "unusable pair", "dual infeasible", "row violated", "row in tolerance band", "guards", "-", "-"], why) if v})
print("where the batches' answer sits relative to the lane-level search (warm pair p, q; L = the row whose line was searched, w = its blocker), per trajectory:")
lab = ["p", "q", "L", "w"]
for w, wn in enumerate(["-", "unusable pair", "dual infeasible", "row violated", "row in tolerance band", "guards", "-", "-"]):
    row = h[4][128 + 16 * w:128 + 16 * w + 16]
    if row.sum():
        print("  %-22s" % wn, {"+".join(l for i, l in enumerate(lab) if c >> i & 1) or "none": round(int(v) / B, 3) for c, v in enumerate(row) if v})
gd = h[3][416:448]
print("verified vertices refused by a guard (per trajectory):", {"+".join(n for i, n in enumerate(["parallel row", "|v1d|", "violation bound", "row 5 in the band (pair not warm)", "limit range"]) if c >> i & 1) or "other (norms, roles, exact test)": round(int(v) / B, 4) for c, v in enumerate(gd) if v})
