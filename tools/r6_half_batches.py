"""tools/r6_half_batches.py -- config 5's per-GPU share (8 restarts x N = 8192): one lock-step batch of 8 against two
half-batches of 4 on two handles / host threads, the second one started `delay` ms after the first (is a chain-bound
group worth overlapping with an update-bound one?)."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import mlii
from bench import synth
N, d = 8192, 8
X, y, _ = synth(N, d, 4, np.float64)
rs = np.random.RandomState(2)
w = rs.uniform(0.25, 2, 64) * np.sqrt(d); h = rs.uniform(0.5, 2, 64); sn = rs.uniform(0.5, 2, 64)
th = np.column_stack([h, w, sn])
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ev = [mlii.BatchEvaluator(X, y), mlii.BatchEvaluator(X, y)]
def timed(fn, reps=10):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps, r
t8, r8 = timed(lambda: ev[0](th[:ROWS]))
t4, r4 = timed(lambda: ev[0](th[:ROWS // 2]))
print("lock-step %d: %.3f ms (%.3f per restart); %d alone: %.3f ms (%.3f per restart)" % (ROWS, t8 * 1e3, t8 / ROWS * 1e3, ROWS // 2, t4 * 1e3, t4 / (ROWS // 2) * 1e3), flush=True)
for delay_ms in (0.0, 1.0, 2.0, 4.0, 6.0, 8.0, 10.0, 12.0):
    out = [None, None]
    def both():
        def second():
            if delay_ms:
                time.sleep(delay_ms * 1e-3)
            out[1] = ev[1](th[ROWS // 2:ROWS])
        t = threading.Thread(target=second)
        t.start()
        out[0] = ev[0](th[:ROWS // 2])
        t.join()
        return np.concatenate(out)
    tt, rr = timed(both)
    same = np.array_equal(rr, r8, equal_nan=True)
    print("two halves, delay %4.1f ms: %.3f ms (%.3f per restart)  bit-identical to lock-step: %s" % (delay_ms, tt * 1e3, tt / ROWS * 1e3, same), flush=True)
