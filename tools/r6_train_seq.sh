#!/bin/bash
# every launch of ONE steady-state training step (tools/bench_train_sym.py under rocprofv3 --kernel-trace): start, duration, queue, name
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/tr_seq
# FREE=1: the free-running loop of the bench's training leg (tools/train_only.py) instead of individually synchronised steps
if [ "${FREE:-0}" = 1 ]; then PROG="$REPO/tools/train_only.py 40"; else PROG=$REPO/tools/bench_train_sym.py; fi
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_seq -- python3 $PROG > /tmp/tr_seq.log 2>&1
tail -2 /tmp/tr_seq.log
python3 - <<P
import csv,glob
f=glob.glob('/tmp/tr_seq/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
marks=[i for i,r in enumerate(rows) if 'encode_kmer' in r['Kernel_Name']]
import os
k=len(marks)//2 if os.environ.get('FREE')=='1' else len(marks)-3      # FREE: a step in the middle of the timed loop
a,b=marks[k],marks[k+1]
t0=int(rows[a]['Start_Timestamp'])
qs={}
for r in rows[a:b]:
    q=qs.setdefault(r.get('Queue_Id','?'), len(qs))
    s=(int(r['Start_Timestamp'])-t0)/1e3; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print('%8.1f %7.1f q%d %s%s  grid %s' % (s, d, q, '  '*q, r['Kernel_Name'].replace('mural::','').replace('(anonymous namespace)::','')[:90], r.get('Grid_Size_X','?')))
print('launches', b-a, 'span us', (int(rows[b]['Start_Timestamp'])-t0)/1e3)
P
