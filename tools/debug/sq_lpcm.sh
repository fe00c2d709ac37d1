cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
w=${SQ_WORKLOAD:-toa_binaural_limiter_s16_lpcm16}; S=${SQ_STREAMS:-4096}
for pm in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS"; do
  tag=$(echo $pm | cut -c1-18 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pm -d $R/gpurun_out/sq_lp/$tag -o t --output-format csv -- python3 $R/bench.py --workload $w --no-cpu-baseline --frames 64 --steps 3 --warmup 1 --streams $S --placement-tries 1 --no-verify --no-facade --repeats 1 > $R/gpurun_out/sq_lp/$tag.log 2>&1 || echo fail $tag
done
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ['GRAFT_REPO_ROOT']
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for f in glob.glob(f'{R}/gpurun_out/sq_lp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        for tag in ('render_fast',):
            if tag in r['Kernel_Name']:
                acc[tag][r['Counter_Name']]+=float(r['Counter_Value']); n[tag][r['Counter_Name']]+=1
for tag in acc:
    print(tag, {k: round(v/max(n[tag][k],1)/1e6,2) for k,v in sorted(acc[tag].items())}, '(millions per launch)')
PY
