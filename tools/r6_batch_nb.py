"""tools/r6_batch_nb.py -- config 5's lock-step batches (8 and 64 restarts x N = 8192) over the outer block width and the 128 x 64-tile threshold"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import mlii
from bench import synth
N, d = 8192, 8
X, y, _ = synth(N, d, 4, np.float64)
rs = np.random.RandomState(2)
w = rs.uniform(0.25, 2, 64) * np.sqrt(d); h = rs.uniform(0.5, 2, 64); sn = rs.uniform(0.5, 2, 64)
th = np.column_stack([h, w, sn])
with mlii.BatchEvaluator(X, y) as ev:
    for env in ({}, {"GPX_POTRF_NB": "512"}, {"GPX_POTRF_NB": "256"}, {"GPX_SYRK_BN64_TILES": "1300"}, {"GPX_SYRK_BN64_TILES": "5200"},
                {"GPX_POTRF_TWO_PART_BATCH": "32768"}, {"GPX_POTRF_TWO_PART_BATCH": "400000"}, {"GPX_POTRF_FOLD_ROWS": "0"}):
        for k in list(os.environ):
            if k.startswith("GPX_"):
                del os.environ[k]
        os.environ.update(env)
        out = []
        for rows, reps in ((8, 10), (16, 6), (64, 3)):
            ev(th[:rows])
            t0 = time.perf_counter()
            for _ in range(reps):
                ev(th[:rows])
            out.append("%d: %.3f ms/restart" % (rows, (time.perf_counter() - t0) / reps / rows * 1e3))
        print("%-40s %s" % (env, "   ".join(out)), flush=True)
