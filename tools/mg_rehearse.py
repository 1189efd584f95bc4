"""tools/mg_rehearse.py <N> <d> <world> [ranks=0] [nb] [chunks] [sag] [link_GBps] [owner_first] -- one rank's share of a `world`-rank fit at
full size on ONE GPU (multi_gpu.rehearse_rank: measured compute, MODELLED transfer), one JSON line per rank; laid against
DESIGN section 5's table.  ranks: comma-separated."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import multi_gpu, _lib
import bench

N, d, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ranks = [int(r) for r in (sys.argv[4] if len(sys.argv) > 4 else "0").split(",")]
opt = lambda i, cast: (cast(sys.argv[i]) if len(sys.argv) > i and sys.argv[i] not in ("", "-") else None)
nb, chunks, sag, rate, ofirst = opt(5, int), opt(6, int), opt(7, int), opt(8, float), opt(9, int)
X, y, Xo = bench.synth(N, d, 16, np.float64)
params, s = np.array([1.0, 0.5 * np.sqrt(d)]), 1.0
for r in ranks:
    out = multi_gpu.rehearse_rank(N, d, r, world, X, y, params, s, nb=nb, chunks=chunks, sag=sag, fits=3,
                                  link_GBps=rate or 100.0, owner_first=ofirst)
    print(json.dumps(out), flush=True)
