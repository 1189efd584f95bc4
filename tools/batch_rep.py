import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPX_BATCH_MAX"] = "3"
from gaussian_processes_amd import mlii
from oracle import gp_oracle as orc
N, d = 1350, 2
X, y, _ = orc.synth_inputs(N, d, 4)
rs = np.random.RandomState(5)
thetas = np.column_stack([rs.uniform(0.5, 2, 5), rs.uniform(0.3, 1.5, 5) * np.sqrt(d), rs.uniform(0.6, 2, 5)])
ref = [orc.OracleGP("gaussian", thetas[i, :2], X, y, thetas[i, 2]).log_lh for i in range(5)]
for dtype in ["float64", "float32", "float64", "float32"]:
    for rep in range(6):
        llh = mlii.log_lh_batch(X, y, thetas, dtype=dtype)
        err = max(abs(llh[i] - ref[i]) / abs(ref[i]) for i in range(1, 5))
        print(dtype, rep, "max rel err %.2e" % err, flush=True)
