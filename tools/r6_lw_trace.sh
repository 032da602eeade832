#!/bin/bash
# kernel durations of the packed long-window path (R = 4000) under rocprofv3, 2048 and 512 windows per call
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for N in 2048 512; do
rm -rf /tmp/lw_tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/lw_tr -- python3 $REPO/tools/r6_lw_trace.py $N > /tmp/lw_tr.log 2>&1
echo "== $N windows per call"
python3 - <<P
import csv,glob,collections
f=glob.glob('/tmp/lw_tr/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
# last call: from the last encode_kmer launch on
last=max(i for i,r in enumerate(rows) if 'encode_kmer' in r['Kernel_Name'])
t0=int(rows[last]['Start_Timestamp'])
for r in rows[last:]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print('  %8.1f %7.1f  %s  grid %s' % (s, d, r['Kernel_Name'].replace('mural::','').replace('(anonymous namespace)::','')[:60], r.get('Grid_Size_X', r.get('Grid_Size', '?'))))
P
done
