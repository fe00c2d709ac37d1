# the LPCM-fed headline at several shard sizes + a kernel trace that shows which kernel ran (tools/debug: one-off)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lpcm; mkdir -p $O
for S in 512 1024 2048; do
  python3 $R/bench.py --workload toa_binaural_limiter_s16_lpcm16 --streams $S --no-cpu-baseline --no-extra-configs --no-facade --repeats 3 > $O/b_$S.json 2> $O/b_$S.err
  python3 -c "import json,sys; d=json.loads(open('$O/b_$S.json').read().strip().splitlines()[-1]); print($S, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['verified'])"
done
IAMF_HIP_LPCM_UNFUSED=1 python3 $R/bench.py --workload toa_binaural_limiter_s16_lpcm16 --streams 1024 --no-cpu-baseline --no-extra-configs --no-facade --repeats 1 --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_lp -o t --output-format csv -- python3 $R/bench.py --workload toa_binaural_limiter_s16_lpcm16 --streams 1024 --no-cpu-baseline --no-extra-configs --no-verify --no-facade --repeats 1 --steps 10 --warmup 2 > $O/trace.log 2>&1
head -4 $(find /tmp/p_lp -name "*kernel_stats.csv" | head -1) | cut -c1-220
