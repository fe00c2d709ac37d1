#!/bin/bash
# tools/ab_signal.sh <reps> <signal> <workload>...
R=${GRAFT_REPO_ROOT:-.}
reps=$1; sig=$2; shift; shift
for w in "$@"; do
  echo -n "$w ($sig):"
  for i in $(seq $reps); do
    v=$(python3 $R/bench.py --workload $w --signal $sig --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']/1000,1))")
    echo -n " $v"
  done
  echo
done
