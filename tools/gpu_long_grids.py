import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from toppra_amd import batch
from oracle import oracle as orc
for (B,d,N,nw) in [(33000,7,1500,5),(40000,5,700,12),(35000,8,260,30)]:
    data=batch.make_synthetic_batch(B,d,N,seed=N,n_waypoints=nw)
    fast=batch.solve_batch(data["coef"],data["breaks"],data["grid"],data["vlim"],data["alim"])
    v3=batch.solve_batch(data["coef"],data["breaks"],data["grid"],data["vlim"],data["alim"],variant=3)
    idx=np.arange(0,B,max(1,B//96))
    ref=orc.solve_batch(data["coef"][idx],data["breaks"],data["grid"],data["vlim"][idx],data["alim"][idx],nthreads=0)
    ok=all(np.array_equal(fast[k][idx],ref[k],equal_nan=True) and np.array_equal(v3[k],fast[k],equal_nan=True) for k in ("K","sd2","u","status"))
    print(B,d,N,nw,"bit-exact vs oracle sample + variant3==auto:",ok, "ok frac",(fast["status"]==0).mean())
