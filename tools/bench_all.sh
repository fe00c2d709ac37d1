#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line of every workload -> gpurun_out/profiles/<round>_<workload>_bench.json
# (after tools/profile_all.sh has produced the PMC summaries the lines take their `traffic` from; copy them into profiles/ first).
set -e
ROUND=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p "$OUT"
for W in ${WORKLOADS:-toa_binaural_limiter_s16 714_ssJ_limiter_s16 toa_ssH_limiter_s16 toa_hrtf256_limiter_s16 scalable_714_ssJ_limiter_s16 toa_projection_binaural_limiter_s16 714_downmix_512_limiter_s16 710_downmix_stereo_limiter_s16 toa_plus_stereo_binaural_limiter_s16 714_plus_stereo_ssJ_limiter_s16}; do
  python3 $R/bench.py --workload $W --steps 20 --warmup 3 > "$OUT/${ROUND}_${W}_bench.json" 2> "$OUT/${W}.bench.log"
  echo "done $W"
done
python3 $R/bench.py --signal quiet --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/${ROUND}_toa_binaural_quiet_bench.json" 2>> "$OUT/quiet.bench.log"
