"""Repeat the C multi-GPU schedule with eight ranks as threads sharing GPU 0 (tests/_thread_world.py) and compare log_lh, alpha and
mean of every repetition with the first bit for bit: python tools/r6_soak_mg_threads.py [reps=6]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _thread_world import run_thread_world       # noqa: E402

reps = int(dict(a.split("=", 1) for a in sys.argv[1:]).get("reps", 6))
for N, nb, sag, dt in ((8492, 512, True, 0), (9000, 256, True, 1), (12288 + 77, 512, False, 0)):
    first, bad = None, 0
    for rep in range(reps):
        res = run_thread_world(8, N, 3, nb, 40, dtype_id=dt, sag=sag)
        cur = (res["log_lh"], np.asarray(res["alpha"]).tobytes(), np.asarray(res["mean"]).tobytes(), res["log_lh2"])
        if first is None:
            first = cur
        elif cur != first:
            bad += 1
            print("N=%d nb=%d dtype %d rep %d differs: log_lh %r vs %r" % (N, nb, dt, rep, cur[0], first[0]), flush=True)
    print("N=%d nb=%d sag=%s dtype %d: %d of %d repetitions differed (each: two fits of eight ranks)" % (N, nb, sag, dt, bad, reps - 1), flush=True)
