cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/lw_tr
LW_R=16000 rocprofv3 --kernel-trace --output-format csv -d /tmp/lw_tr -- python3 $REPO/tools/r6_lw_trace.py 512 > /tmp/lw_tr.log 2>&1
python3 - <<P
import csv,glob
f=glob.glob('/tmp/lw_tr/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
marks=[i for i,r in enumerate(rows) if 'encode_kmer' in r['Kernel_Name']]
a=marks[-2]; b=marks[-1]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print('  %8.1f %7.1f  %s  grid %s' % (s, d, r['Kernel_Name'].replace('mural::','').replace('(anonymous namespace)::','')[:70], r.get('Grid_Size_X','?')))
P
