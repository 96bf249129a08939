"""Round-6 forensics: where do the controllable sets of family 3 differ from family 2 (one dof, plain solve)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from toppra_amd import batch, _capi
_capi.init(0)
np.set_printoptions(precision=17, linewidth=220)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 5
data = batch.make_synthetic_batch(128, d, 60, seed=11)
args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
a = batch.solve_batch(*args, variant=2)
b = batch.solve_batch(*args, variant=3)
K2, K3 = a["K"], b["K"]
diff = ~((K2 == K3) | (np.isnan(K2) & np.isnan(K3)))
print("trajectories with wrong K:", int(diff.any(axis=(1, 2)).sum()), "of", len(K2), "; lower bound wrong:", int(diff[:, :, 0].sum()), "upper bound wrong:", int(diff[:, :, 1].sum()))
t = int(np.argmax(diff.any(axis=(1, 2))))
st = np.flatnonzero(diff[t].any(axis=1))
print("trajectory", t, "stages wrong:", st.tolist())
for i in st[-4:]:
    print(" stage", i, "v2", K2[t, i], "v3", K3[t, i])
print("statuses v2/v3:", a["status"][:16].tolist(), b["status"][:16].tolist())
# how many wrong per stage index (all trajectories)
print("wrong entries per stage:", diff.any(axis=2).sum(axis=0).tolist())
