#!/bin/bash
# A/B of two builds of libiamf_hip.so on ONE box (box-to-box spread is larger than most kernel changes):
#   tools/ab.sh <label> [bench.py args...]   -> gpurun_out/ab_<label>.txt, alternating base / new, 3 rounds
# base = iac_amd/lib/libiamf_hip_base.so (a build of an older commit), new = iac_amd/lib/libiamf_hip.so
label=$1; shift
out=gpurun_out/ab_$label.txt
: > $out
for r in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export IAMF_HIP_LIB=$PWD/iac_amd/lib/libiamf_hip_base.so; else unset IAMF_HIP_LIB; fi
    line=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-configs --repeats 1 "$@" 2>/dev/null | tail -1)
    echo "$v $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"], d["roofline"]["kernel"])')" >> $out
  done
done
cat $out
