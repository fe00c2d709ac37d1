# kernel trace of the facade rates of bench.py with and without the fused LPCM form of the single handle (one-off)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for V in unpack fused; do
  if [ $V = unpack ]; then export IAMF_HIP_FACADE_UNPACK=1; else unset IAMF_HIP_FACADE_UNPACK; fi
  rm -rf /tmp/pf_$V
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$V -o t --output-format csv -- python3 $R/bench.py --no-extra-configs --no-verify --repeats 1 --placement-tries 1 --steps 2 --warmup 1 > /dev/null 2>&1
  echo "== $V"; head -8 $(find /tmp/pf_$V -name "*kernel_stats.csv" | head -1) | cut -c1-170
done
