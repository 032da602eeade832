#!/bin/bash
# kernel durations of the long-window path (R = 4000, dense entry, 2048 windows per call) under rocprofv3
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/lw_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lw_tr -- python3 $REPO/tools/archive/r4_long_window.py ${1:-2048} > /tmp/lw_tr.log 2>&1
head -3 /tmp/lw_tr.log
python3 - <<P
import csv,glob,collections
f=glob.glob('/tmp/lw_tr/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name'][:70]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
tot=sum(sum(v) for v in d.values())
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:14]: print('  %-70s n=%4d tot=%9.1f (%4.1f%%) med=%.1f'%(k,len(v),sum(v),100*sum(v)/tot,sorted(v)[len(v)//2]))
P
