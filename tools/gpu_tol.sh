#!/bin/bash
# what bit-exactness costs: timing + deviation of relaxed-arithmetic builds against the product build
for lib in build_dbg/lib_base.so "$@"; do
  TOPPRA_HIP_LIB=$PWD/$lib python - <<PY
import os,sys,numpy as np
sys.path.insert(0,os.getcwd())
from toppra_amd import batch
data=batch.make_synthetic_batch(65536,7,200)
out=batch.solve_batch(data["coef"],data["breaks"],data["grid"],data["vlim"],data["alim"],variant=3)
np.savez("/tmp/out_%s.npz"%os.path.basename("$lib"),**out)
PY
done
python - "$@" <<'PY'
import sys,os,numpy as np
ref=dict(np.load("/tmp/out_lib_base.so.npz"))
for lib in sys.argv[1:]:
    o=dict(np.load("/tmp/out_%s.npz"%os.path.basename(lib)))
    print(lib,"status equal",np.array_equal(o["status"],ref["status"]),
          "max|dsd2| %.3e"%np.nanmax(np.abs(o["sd2"]-ref["sd2"])),"max|dK| %.3e"%np.nanmax(np.abs(o["K"]-ref["K"])),
          "max|du| %.3e"%np.nanmax(np.abs(o["u"]-ref["u"])), "nan pattern equal", np.array_equal(np.isnan(o["sd2"]),np.isnan(ref["sd2"])))
PY
tools/gpu_ab.sh build_dbg/lib_base.so "$@"
