#!/bin/bash
# small sizes: stage times of the handle path
cd "$(dirname "$0")/.."
for n in 256 512 1024 2048; do
  python bench.py --problem-n $n --problem-d 8 --problem-m 256 --dtype f64 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('n=$n value %.1f us' % (j['value']*1e6), j['stages_ms'])"
done
python tools/api_lat.py 2>&1 | tail -12
