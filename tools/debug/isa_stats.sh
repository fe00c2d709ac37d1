#!/bin/bash
# Static picture of one kernel of one translation unit, no GPU needed: registers, spills, and the instruction mix between
# the s_barriers of its main loop.   tools/debug/isa_stats.sh iamf_render_lpcm.hip 'render_fast_kernelILi16ELi2ELi0ELb0ELb0ELb1E'
src=$1; pat=$2; out=${3:-/tmp/isa_stats}
mkdir -p $out && cd $out || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -Wno-unused-function -Wno-undefined-internal $EXTRA --save-temps -c /root/repo/iac_amd/csrc/$src -o tu.o 2>&1 | grep -E "error" -A3
S=$(ls *-hip-amdgcn-amd-amdhsa-gfx950.s | head -1)
name=$(grep -o "^_Z[A-Za-z0-9_]*${pat}[A-Za-z0-9_]*:" $S | head -1 | tr -d ':')
echo "kernel $name"
awk -v n="$name:" 'index($0,n)==1{f=1} f{print} /s_endpgm/{if(f){exit}}' $S > k.s
grep -A12 "\.name: *$name" $S | grep -E "vgpr_count|sgpr_count|spill|private_segment_fixed"
echo "scratch ops: $(grep -c 'scratch_' k.s)   v_readlane: $(grep -c v_readlane k.s)  v_writelane: $(grep -c v_writelane k.s)  total valu: $(grep -c '^\s*v_' k.s)"
L=$(grep -n "Loop Header: Depth=1" k.s | tail -1 | cut -d: -f1)
awk -v L=$L 'NR>=L {if ($1 ~ /^s_barrier/) {print "  up to barrier at line "NR":  valu="v" salu="s" ds="d" vmem="m" nop="n; v=0;s=0;d=0;m=0;n=0} else if ($1 ~ /^v_/) v++; else if ($1 ~ /^s_nop/) n++; else if ($1 ~ /^s_/) s++; else if ($1 ~ /^ds_/) d++; else if ($1 ~ /^(global|buffer|flat|scratch)_/) m++;}' k.s
