#!/bin/bash
# A/B of the product library against another build of it on ONE box:
#   tools/ab_lib.sh <label> <other libiamf_hip.so> [bench.py args...]  -> gpurun_out/ab_<label>.txt, alternating, 3 rounds
label=$1; other=$2; shift 2
out=gpurun_out/ab_$label.txt
: > $out
for r in 1 2 3; do
  for v in product other; do
    if [ $v = other ]; then export IAMF_HIP_LIB=$PWD/$other; else unset IAMF_HIP_LIB; fi
    line=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-configs --repeats 1 --placement-tries 1 "$@" 2>/dev/null | tail -1)
    echo "$v $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"], d["roofline"]["kernel"])')" >> $out
  done
done
unset IAMF_HIP_LIB
cat $out
