#!/bin/bash
# kernel durations of the dense-input training route (encode_onehot, dense_to_symbols and what else differs from the symbol route)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/dn_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dn_tr -- python3 $REPO/tools/archive/r5_dense_route.py 20 > /tmp/dn_tr.log 2>&1
tail -6 /tmp/dn_tr.log
python3 - <<P
import csv,glob,collections
f=glob.glob('/tmp/dn_tr/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    if any(t in k for t in ('encode','dense_to','sym_hist','first_train','first_tables')): print('  %-60s n=%4d med=%.1f min=%.1f'%(k,len(v),sorted(v)[len(v)//2],min(v)))
P
