#!/bin/bash
# A/B of environment settings on the whole bench step, alternating runs: tools/ab_env.sh <rounds> "VAR=val ..." "VAR=val ..." ...
# prints ms/step and img/s per setting and round (same box, same process sequence)
R=$1; shift
for r in $(seq 1 $R); do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-resident-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['ms_per_step'],2), round(d['value'],1))"
  done
done
