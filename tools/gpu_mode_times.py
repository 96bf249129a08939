"""Kernel time (HIP events inside the library) of the throughput families in fast and sound certificate modes."""
import sys, os, numpy as np, torch, time
sys.path.insert(0, os.getcwd())
from toppra_amd import batch as tb
dev = torch.device("cuda", 0)
for B, d, N in ((65536, 7, 200), (65536, 6, 100)):
    data = tb.make_synthetic_batch(B, d, N)
    dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    for variant in (3, 2):
        for sound in (False, True):
            out = tb.solve_batch(*dv, variant=variant, sound=sound)
            ms = tb.solve_batch_timed(*dv, out, 10, variant=variant, sound=sound)
            print("B %d d %d N %d variant %d sound %d kernel ms %.3f" % (B, d, N, variant, sound, ms), flush=True)
