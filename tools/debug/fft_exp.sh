#!/bin/bash
# Timing-only elimination builds of render_fir_fft.hpp (IAMF_FFT_EXP=n, WRONG results by construction):
#   tools/debug/fft_exp.sh build  -> iac_amd/lib/fftexp<n>/libiamf_hip.so        (run in the authoring container)
#   tools/debug/fft_exp.sh run    -> gpurun_out/fft_exp.txt: product and every variant on the HRTF workload, same box
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for n in ${EXPS:-1 2 3 4 5}; do
    make -s -C iac_amd/csrc -j6 EXTRA=-DIAMF_FFT_EXP=$n BUILD=$PWD/iac_amd/csrc/build_exp_t$n OUTDIR=$PWD/iac_amd/lib/fftexp$n $PWD/iac_amd/lib/fftexp$n/libiamf_hip.so
  done
  exit 0
fi
out=gpurun_out/fft_exp.txt
: > $out
for n in 0 ${EXPS:-1 2 3 4 5}; do
  if [ $n = 0 ]; then unset IAMF_HIP_LIB; else export IAMF_HIP_LIB=$PWD/iac_amd/lib/fftexp$n/libiamf_hip.so; fi
  line=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-verify --repeats 1 --placement-tries 1 --streams 1024 --workload toa_hrtf256_limiter_s16 --no-facade --hrir-scale 0.0481 2>/dev/null | tail -1)
  echo "hrtf256 exp$n $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"])')" >> $out
done
cat $out
