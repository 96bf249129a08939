import os, sys
sys.path.insert(0, os.path.abspath(sys.argv[1]))
import numpy as np
from toppra_amd import batch
np.set_printoptions(precision=17, linewidth=200)
for d in (11, 13):
    for B, N in ((1, 2), (1, 6), (3, 6)):
        data = batch.make_synthetic_batch(B, d, N, seed=60 + d)
        args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
        Xf = batch.feasible_sets_batch(*args, True, variant=2, strict=True)
        # poison the allocator's next block so that an untouched output shows
        junk = batch.feasible_sets_batch(*(a if i else a * 1.7 for i, a in enumerate(args)), True, variant=2, strict=True)
        del junk
        Xs = batch.feasible_sets_batch(*args, True, variant=3, sound=True)
        Xq = batch.feasible_sets_batch(*args, True, variant=3, sound=False)
        print("d", d, "B", B, "N", N)
        print(" full  ", Xf.reshape(B, -1))
        print(" sound ", Xs.reshape(B, -1))
        print(" fast  ", Xq.reshape(B, -1))
