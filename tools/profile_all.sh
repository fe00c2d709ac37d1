#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats and the two HBM PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md) of `bench.py` for every
# workload AT THE STREAM COUNT THE BENCH LINE REPORTS IT AT (the r02 summaries of configs 2 / 3 were
# traced at 512 streams while the line ran them at 3072 / 2048), then condenses them into
# profiles/<round>_<workload>_* with tools/prof_summary.py; an unprofiled run of the same command gives the bench line and
# the card's rocm-smi read-out, which go into the summary too.
#   tools/profile_all.sh r03            -> gpurun_out/profiles/r03_*   (copy into profiles/ afterwards)
set -e
ROUND=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# workload:streams per GPU, as bench.py's default line (EXTRA_CONFIGS) runs them
for WS in ${WORKLOADS:-toa_binaural_limiter_s16:512 714_ssJ_limiter_s16:3072 toa_ssH_limiter_s16:2048 toa_hrtf256_limiter_s16:1024 toa_binaural_limiter_s16_lpcm16:4096 scalable_714_ssJ_limiter_s16:2048 toa_ssB_lfe_limiter_s16:4096 710_downmix_stereo_limiter_s16:2048}; do
  W=${WS%%:*}; S=${WS##*:}
  P=$R/gpurun_out/prof_$W
  rm -rf "$P"
  # the placement search of the default line (NOTEBOOK.md 3): the timed launches run on the buffers the bench line is measured on
  CMD="bench.py --workload $W --streams $S --no-cpu-baseline --no-extra-configs --no-verify --no-facade --repeats 1 --steps 10 --warmup 2"
  rocprofv3 --kernel-trace --stats -d "$P/trace" -o t --output-format csv -- python3 $R/$CMD > "$P.trace.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$P/pmc_fetch" -o t --output-format csv -- python3 $R/$CMD > "$P.fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$P/pmc_write" -o t --output-format csv -- python3 $R/$CMD > "$P.write.log" 2>&1
  F=64
  python3 $R/bench.py --workload $W --streams $S --no-extra-configs --no-cpu-baseline --no-facade --steps 20 --warmup 3 > "$OUT/${ROUND}_${W}_bench.json" 2> "$P.bench.log"
  python3 "$R/tools/prof_summary.py" "$P" "$OUT/${ROUND}_$W" "python3 $CMD" $((S * F * 1024)) 10 "$OUT/${ROUND}_${W}_bench.json" > /dev/null
  echo "done $W"
done
