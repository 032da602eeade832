#!/bin/bash
# Kernel durations of the INDEL level-0 launches under rocprofv3 (a lone 2048-position forward through the packed entry), once per
# setting of the switch named in $1 (default: MURAL_INDEL_ENC0_DOWN) -- e.g.  bash tools/archive/r5_l0_ab.sh MURAL_INDEL_DEC0
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
SW=${1:-MURAL_INDEL_ENC0_DOWN}
for q in 1 0; do
  export $SW=$q
  rm -rf /tmp/l0ab_$q
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/l0ab_$q -- python3 $REPO/tools/bench_indel.py 2048 packed > /tmp/l0ab_$q.log 2>&1
  echo "$SW=$q"; python3 - <<P
import csv,glob,collections
f=glob.glob('/tmp/l0ab_$q/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'indel_' in n or 'convblock_kernel<8' in n or 'conv1d_direct_kernel<1, 4, 4' in n: d[n.split('mural::')[-1][:44]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for k,v in d.items(): print('  %-44s n=%d med=%.1f min=%.1f'%(k,len(v),sorted(v)[len(v)//2],min(v)))
P
done
