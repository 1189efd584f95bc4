#!/bin/bash
# tools/grad_prof.sh -- rocprofv3 kernel stats of fit + dloglh_dtheta at N = 8192, d = 8 (diagnostic)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp_stats -o g -- python3 ${GRAFT_REPO_ROOT:-/root/repo}/tools/grad_bench.py 8192 8 > /tmp/g.out 2>/tmp/g.err
cat /tmp/g.out
f=$(find /tmp/gp_stats -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-200
