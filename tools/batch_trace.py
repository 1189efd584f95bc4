"""tools/batch_trace.py <rows> -- one lock-step evaluation of <rows> restarts x N = 8192 (after two warm-ups), for
rocprofv3 --kernel-trace (tools/batch_trace.sh turns the trace into a timeline)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_processes_amd import mlii
import bench
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, d = 8192, 8
X, y, _ = bench.synth(N, d, 4, np.float64)
rs = np.random.RandomState(2)
th = np.column_stack([rs.uniform(0.5, 2, 64), rs.uniform(0.25, 2, 64) * np.sqrt(d), rs.uniform(0.5, 2, 64)])
with mlii.BatchEvaluator(X, y) as ev:
    for _ in range(3):
        ev(th[:rows])
