#!/bin/bash
# A/B of the level-0 kernels' tile queue under rocprofv3 (kernel durations of a lone 2048-position forward)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for q in 1 0; do
  export MURAL_INDEL_L0_QUEUE=$q
  rm -rf /tmp/l0ab_$q
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/l0ab_$q -- python3 $REPO/tools/bench_indel.py 2048 packed > /tmp/l0ab_$q.log 2>&1
  echo "QUEUE=$q"; python3 - <<P
import csv,glob,collections
f=glob.glob('/tmp/l0ab_$q/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'indel_' in n: d[n.split('indel_')[1][:24]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for k,v in d.items(): print('  %-40s n=%d med=%.1f min=%.1f'%(k,len(v),sorted(v)[len(v)//2],min(v)))
P
done
