set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/clk; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for SIG in hot quiet zero; do
  rm -rf /tmp/p_$SIG
  rocprofv3 --kernel-trace --stats -d /tmp/p_$SIG -o t --output-format csv -- python3 $R/bench.py --workload toa_hrtf256_limiter_s16 --streams 1024 --signal $SIG --no-cpu-baseline --no-extra-configs --no-verify --no-facade --repeats 1 --steps 10 --warmup 2 --placement-tries 1 > $O/hrtf_$SIG.log 2>&1
  f=$(find /tmp/p_$SIG -name "*kernel_stats.csv" | head -1)
  echo "== $SIG" >> $O/summary.txt; head -6 $f | cut -c1-200 >> $O/summary.txt
done
cat $O/summary.txt
