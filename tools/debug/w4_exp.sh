#!/bin/bash
# Timing-only elimination builds of render_wide4.hpp (IAMF_W4_EXP=n, WRONG results by construction):
#   tools/debug/w4_exp.sh build   -> iac_amd/lib/exp<n>/libiamf_hip.so for n = 1..4   (run in the authoring container)
#   tools/debug/w4_exp.sh run     -> gpurun_out/w4_exp.txt: product and every variant on cfg2 / cfg3, quiet and hot, same box
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for n in ${EXPS:-1 2 3 4 6}; do
    make -s -C iac_amd/csrc -j6 EXTRA=-DIAMF_W4_EXP=$n BUILD=$PWD/iac_amd/csrc/build_exp$n OUTDIR=$PWD/iac_amd/lib/exp$n $PWD/iac_amd/lib/exp$n/libiamf_hip.so
  done
  exit 0
fi
out=gpurun_out/w4_exp.txt
: > $out
for wl in 714_ssJ_limiter_s16 toa_ssH_limiter_s16; do
  for sig in quiet hot; do
    for n in 0 ${EXPS:-1 2 3 4 6}; do
      if [ $n = 0 ]; then unset IAMF_HIP_LIB; else export IAMF_HIP_LIB=$PWD/iac_amd/lib/exp$n/libiamf_hip.so; fi
      line=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-configs --repeats 1 --placement-tries 1 --workload $wl --signal $sig 2>/dev/null | tail -1)
      echo "$wl $sig exp$n $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"])')" >> $out
    done
  done
done
cat $out
