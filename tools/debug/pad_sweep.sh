#!/bin/bash
# experiment: does breaking the power-of-two stride between streams change the HBM rate?
R=${GRAFT_REPO_ROOT:-.}
for pad in ${PADS:-0 4 12 20 36 68 132}; do
  echo -n "pad $pad:"
  for rep in 1 2 3 4 5 6; do
    v=$(python3 $R/bench.py --workload ${1:-toa_binaural_limiter_s16} --pad-kb $pad --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']/1000,1))")
    echo -n " $v"
  done
  echo
done
