#!/bin/bash
# same-box A/B of libiamf_hip_base.so (the round's starting build) against the current build on the workloads the
# limiter / prefetch work of round 4 touches.  -> gpurun_out/r4/ab_<label>.txt
label=${1:-r4}; out=gpurun_out/r4/ab_$label.txt; mkdir -p gpurun_out/r4; : > $out
run() {  # name, bench args
  name=$1; shift
  for r in 1 2; do
    for v in base ${VARIANTS} new; do
      if [ $v = base ]; then export IAMF_HIP_LIB=$PWD/iac_amd/lib/libiamf_hip_base.so; elif [ $v = new ]; then unset IAMF_HIP_LIB; else export IAMF_HIP_LIB=$PWD/iac_amd/lib/$v/libiamf_hip.so; fi
      line=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs --no-facade --repeats 3 --placement-tries 1 --pcm-placement-tries 1 "$@" 2>/dev/null | tail -1)
      echo "$name $v $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("verified",{}).get("max_lsb"))')" >> $out
    done
  done
}
for w in ${WORKS:-headline512 headline2048 lpcm4096 dmx2048 hrtf1024}; do
  case $w in
    headline512) run headline512 --workload toa_binaural_limiter_s16 --streams 512;;
    headline1024) run headline1024 --workload toa_binaural_limiter_s16 --streams 1024;;
    headline2048) run headline2048 --workload toa_binaural_limiter_s16 --streams 2048;;
    lpcm512) run lpcm512 --workload toa_binaural_limiter_s16_lpcm16 --streams 512;;
    lpcm4096) run lpcm4096 --workload toa_binaural_limiter_s16_lpcm16 --streams 4096;;
    dmx2048) run dmx2048 --workload 710_downmix_stereo_limiter_s16 --streams 2048;;
    hrtf1024) run hrtf1024 --workload toa_hrtf256_limiter_s16 --streams 1024;;
    cfg2) run cfg2 --workload 714_ssJ_limiter_s16 --streams 3072;;
    cfg3) run cfg3 --workload toa_ssH_limiter_s16 --streams 2048;;
    demix) run demix --workload scalable_714_ssJ_limiter_s16 --streams 2048;;
  esac
done
cat $out
