#!/bin/bash
# per-kernel durations of a lone INDEL forward (2048 positions, packed entry) under rocprofv3: every kernel, median per launch and the
# number of launches per forward -- where the next microsecond is
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/indel_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/indel_tr -- python3 $REPO/tools/bench_indel.py ${1:-2048} packed > /tmp/indel_tr.log 2>&1
tail -1 /tmp/indel_tr.log
python3 - <<P
import csv,glob,collections
f=glob.glob('/tmp/indel_tr/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name'].replace('mural::','').replace('(anonymous namespace)::','')[:64]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
nf=min(len(v) for v in d.values() if len(v) > 8)      # forwards traced = launches of the rarest repeated kernel
tot=0
rows=[]
for k,v in d.items():
    per=len(v)/nf
    med=sorted(v)[len(v)//2]
    rows.append((per*med,k,per,med)); tot+=per*med
print('forwards %d, sum of medians per forward %.1f us'%(nf,tot))
for t,k,per,med in sorted(rows,reverse=True)[:40]: print('  %-64s x%-4.1f med=%7.1f  %7.1f us (%4.1f%%)'%(k,per,med,t,100*t/tot))
P
