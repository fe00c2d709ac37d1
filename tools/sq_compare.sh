cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in scalable_714_ssJ_limiter_s16 714_ssJ_limiter_s16; do
  for pm in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
    tag=$(echo $pm | cut -c1-18 | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $pm -d $R/gpurun_out/sq2/$w/$tag -o t --output-format csv -- python3 $R/bench.py --workload $w --signal quiet --no-cpu-baseline --frames 64 --steps 3 --warmup 1 > $R/gpurun_out/sq2/$w.$tag.log 2>&1 || echo fail $w $tag
  done
done
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ['GRAFT_REPO_ROOT']
for w in ['scalable_714_ssJ_limiter_s16','714_ssJ_limiter_s16']:
    acc=collections.defaultdict(float); n=collections.Counter()
    for f in glob.glob(f'{R}/gpurun_out/sq2/{w}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'render_wide4' in r['Kernel_Name']:
                acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
    print(w, {k: round(v/max(n[k],1)/1e6,2) for k,v in sorted(acc.items())})
PY
