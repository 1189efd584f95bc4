"""tools/r3_batch_split.py [rows] -- `rows` restarts x N = 8192: one lock-step batch against g concurrent lock-step
batches of rows / g on g host threads (each thread has its own handle and stream set): ms per restart, median of 5."""
import os, sys, time, statistics, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import mlii
import bench
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, d = 8192, 8
X, y, _ = bench.synth(N, d, 4, np.float64)
rs = np.random.RandomState(2)
th = np.column_stack([rs.uniform(0.5, 2, 64), rs.uniform(0.25, 2, 64) * np.sqrt(d), rs.uniform(0.5, 2, 64)])[:rows]
ref = None
for g in (1, 2, 4):
    if rows % g:
        continue
    evs = [mlii.BatchEvaluator(X, y) for _ in range(g)]
    parts = np.array_split(np.arange(rows), g)
    out = np.empty(rows)
    # persistent worker threads: the library's look-ahead streams and scratch are per host thread
    go, done, stop = threading.Barrier(g + 1), threading.Barrier(g + 1), [False]
    def worker(i):
        while True:
            go.wait()
            if stop[0]: return
            out[parts[i]] = evs[i](th[parts[i]])
            done.wait()
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(g)]
    for t in ths: t.start()
    def run():
        go.wait(); done.wait()
    run(); run()
    tt = []
    for _ in range(5):
        t0 = time.perf_counter(); run(); tt.append(time.perf_counter() - t0)
    stop[0] = True; go.wait()
    for t in ths: t.join()
    if ref is None: ref = out.copy()
    print("rows %2d in %d concurrent batch(es): %.4f s = %.3f ms per restart  (max |diff| vs one batch %.2e)"
          % (rows, g, statistics.median(tt), statistics.median(tt) / rows * 1e3, np.nanmax(np.abs(np.where(np.isfinite(ref), out - ref, 0.0)))))
    for e in evs: e.close()
