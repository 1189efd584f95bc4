#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - <<'PY' 2>&1 | tail -15
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import importlib, numpy as np
import test_gpu_configs as t
bad = 0
for rep in range(150):
    try:
        t.test_two_async_fits_on_two_streams_of_one_thread_do_not_share_the_panel_scratch()
    except AssertionError as e:
        bad += 1
        print("rep", rep, "FAILED:", str(e).replace("\n", " ")[:300])
print("failures:", bad, "of 150")
PY
