"""Round-6: the 15-dof TOPPRAsd incident at row level.  For every stage where family 3's fastest profile leaves family 2's although x
still agrees, rebuild the stage's rows with the CPU restatement and say which row's quotient each of the two answers is."""
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from toppra_amd import batch, _capi
from oracle import oracle
from tests.test_gpu_instantiations import _problem
_capi.init(0)
d = 15
data, grid, sd0, sd1 = _problem(d, 700 + d, False)
desired = np.random.default_rng(d).uniform(0.5, 5.0, size=96)
args = (data["coef"], data["breaks"], grid, data["vlim"], data["alim"], desired, sd0, sd1)
a = batch.solve_desired_duration_batch(*args, variant=2, interpolation=True)
b = batch.solve_desired_duration_batch(*args, variant=3, interpolation=True)
N = grid.shape[-1] - 1
hist = {}
for t in range(96):
    if a["alpha"][t] != 1.0 or np.array_equal(a["u"][t], b["u"][t], equal_nan=True):
        continue  # (alpha = 1: sd2 / u ARE the fastest profile)
    W = oracle.Wrapper(data["coef"][t], data["breaks"] if data["breaks"].ndim == 1 else data["breaks"][t], grid, data["vlim"][t], data["alim"][t])
    A, Bm, Cm = W.a_arr, W.b_arr, W.c_arr
    for i in range(N):
        if a["sd2"][t, i] == b["sd2"][t, i] and a["u"][t, i] != b["u"][t, i]:
            x, delta = a["sd2"][t, i], grid[i + 1] - grid[i]
            ar, br, cr = A[i].copy(), Bm[i].copy(), Cm[i].copy()
            ar[0], br[0], cr[0] = -2 * delta, -1.0, a["K"][t, i + 1, 0]
            ar[1], br[1], cr[1] = 2 * delta, 1.0, -a["K"][t, i + 1, 1]
            with np.errstate(divide="ignore", invalid="ignore"):
                q = -(br * x + cr) / ar
            up = ar > 1e-10
            order = np.argsort(np.where(up, q, np.inf))
            j_ref = int(np.flatnonzero(up & (q == a["u"][t, i]))[0]) if (up & (q == a["u"][t, i])).any() else -1
            j_got = int(np.flatnonzero(up & (q == b["u"][t, i]))[0]) if (up & (q == b["u"][t, i])).any() else -1
            def name(j):
                if j < 0: return "no row"
                if j < 2: return "x_next row %d" % j
                m = j - 2; blk, k = divmod(m, d)
                return "block %d (%s%s) joint %d" % (blk, "+-"[blk & 1], "q(s_i)" if blk < 2 else "q(s_i+1)", k)
            print("traj %2d stage %2d: reference = row %3d [%s] (rank %d); family 3 = row %3d [%s] (rank %d); u %.17g vs %.17g" % (
                t, i, j_ref, name(j_ref), int(np.flatnonzero(order == j_ref)[0]) if j_ref >= 0 else -1, j_got, name(j_got),
                int(np.flatnonzero(order == j_got)[0]) if j_got >= 0 else -1, a["u"][t, i], b["u"][t, i]))
            hist[(name(j_ref), name(j_got))] = hist.get((name(j_ref), name(j_got)), 0) + 1
print("pairs (reference row, family-3 row): count")
for k, v in sorted(hist.items(), key=lambda kv: -kv[1]): print("  ", k, v)
