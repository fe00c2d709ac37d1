#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats and the two HBM PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md) of `bench.py` for every
# workload, then condenses them into profiles/<round>_<workload>_* with tools/prof_summary.py.
#   tools/profile_all.sh r01            -> gpurun_out/profiles/r01_*   (copy into profiles/ afterwards)
set -e
ROUND=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for W in ${WORKLOADS:-toa_binaural_limiter_s16 714_ssJ_limiter_s16 toa_ssH_limiter_s16 toa_hrtf256_limiter_s16 scalable_714_ssJ_limiter_s16 toa_projection_binaural_limiter_s16 714_downmix_512_limiter_s16 710_downmix_stereo_limiter_s16 toa_plus_stereo_binaural_limiter_s16 714_plus_stereo_ssJ_limiter_s16}; do
  P=$R/gpurun_out/prof_$W
  rm -rf "$P"
  # default placement search: the timed launches run on the pair of buffers the bench line is measured on
  ARGS="$R/bench.py --workload $W --no-cpu-baseline --no-extra-configs --repeats 1 --steps 10 --warmup 2"
  rocprofv3 --kernel-trace --stats -d "$P/trace" -o t --output-format csv -- python3 $ARGS > "$P.trace.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$P/pmc_fetch" -o t --output-format csv -- python3 $ARGS > "$P.fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$P/pmc_write" -o t --output-format csv -- python3 $ARGS > "$P.write.log" 2>&1
  F=64
  python3 "$R/tools/prof_summary.py" "$P" "$OUT/${ROUND}_$W" "python3 bench.py --workload $W --no-cpu-baseline --no-extra-configs --repeats 1 --steps 10 --warmup 2" $((512 * F * 1024)) 10 > /dev/null
  python3 $R/bench.py --workload $W --no-extra-configs --steps 20 --warmup 3 > "$OUT/${ROUND}_${W}_bench.json" 2> "$P.bench.log"
  echo "done $W"
done
