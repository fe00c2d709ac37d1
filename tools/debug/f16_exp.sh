#!/bin/bash
# Timing-only elimination builds of render_fir16.hpp (IAMF_F16_EXP=n, WRONG results by construction):
#   tools/debug/f16_exp.sh build  -> iac_amd/lib/f16exp<n>/libiamf_hip.so        (run in the authoring container)
#   tools/debug/f16_exp.sh run    -> gpurun_out/f16_exp.txt: product and every variant on the HRTF workload, same box
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for n in ${EXPS:-1 2 3}; do
    make -s -C iac_amd/csrc -j6 EXTRA=-DIAMF_F16_EXP=$n BUILD=$PWD/iac_amd/csrc/build_exp_f$n OUTDIR=$PWD/iac_amd/lib/f16exp$n $PWD/iac_amd/lib/f16exp$n/libiamf_hip.so
  done
  exit 0
fi
out=gpurun_out/f16_exp.txt
: > $out
for n in 0 ${EXPS:-1 2 3}; do
  if [ $n = 0 ]; then unset IAMF_HIP_LIB; else export IAMF_HIP_LIB=$PWD/iac_amd/lib/f16exp$n/libiamf_hip.so; fi
  line=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-configs --repeats 1 --placement-tries 1 --workload toa_hrtf256_limiter_s16 2>/dev/null | tail -1)
  echo "hrtf256 exp$n $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"])')" >> $out
done
cat $out
