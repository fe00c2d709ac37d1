#!/bin/bash
# SQ wave-cycle counters of the four BASELINE single-GPU workloads (rocprofv3 --pmc, kernel trace only: the
# combination gpurun allows), condensed by tools/sq_summary.py.   tools/sq_all.sh r02
set -e
ROUND=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/sq"
for WS in ${SQ_WORKLOADS:-toa_binaural_limiter_s16:512 714_ssJ_limiter_s16:3072 toa_ssH_limiter_s16:2048 toa_hrtf256_limiter_s16:1024 toa_binaural_limiter_s16_lpcm16:4096}; do
  W=${WS%%:*}; S=${WS##*:}
  ARGS="$R/bench.py --workload $W --streams $S --no-facade --no-verify --no-cpu-baseline --no-extra-configs --placement-tries 1 --repeats 1 --steps 6 --warmup 2"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES \
      -d "$R/gpurun_out/sq/$W" -o t --output-format csv -- python3 $ARGS > "$R/gpurun_out/sq_$W.log" 2>&1 || echo "rocprof failed for $W"
  echo "done $W"
done
mkdir -p "$R/gpurun_out/profiles"
python3 "$R/tools/sq_summary.py" "$R/gpurun_out/sq" "$R/gpurun_out/profiles/${ROUND}_sq_counters.json" > /dev/null
