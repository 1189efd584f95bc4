#!/bin/bash
# potrf time of the chain-bound sizes (handle path, fp64 d = 8): tools/r6_small_fits.sh [tag]
for n in 2048 4096 8192 8191 12288; do
  python bench.py --problem-n $n --problem-d 8 --problem-m 1024 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-prof 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('n=%d fit+predict %.3f ms potrf %.3f ms (%.3f of peak) log_lh %r' % (d['config']['N'], d['ms_per_step'], d['stages_ms']['potrf'], d.get('potrf_frac_of_peak', 0), d.get('log_lh')))"
done
python bench.py --problem-n 4096 --problem-d 8 --problem-m 1024 --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-prof 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('f32 n=%d fit+predict %.3f ms potrf %.3f ms log_lh %r' % (d['config']['N'], d['ms_per_step'], d['stages_ms']['potrf'], d.get('log_lh')))"
