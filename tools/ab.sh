#!/bin/bash
# A/B helper on one GPU box: runs bench workloads several times, prints Gsamples/s.  tools/ab.sh <reps> <workload>...
R=${GRAFT_REPO_ROOT:-.}
reps=$1; shift
for w in "$@"; do
  echo -n "$w:"
  for i in $(seq $reps); do
    v=$(python3 $R/bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']/1000,1))")
    echo -n " $v"
  done
  echo
done
