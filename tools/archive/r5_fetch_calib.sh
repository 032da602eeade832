#!/bin/bash
# FETCH_SIZE calibration per read pattern (tools/microbench/fetch_calib.hip): bytes read / (FETCH_SIZE x 1024) per kernel
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $REPO/tools/microbench/fetch_calib.hip || exit 1
rm -rf /tmp/fc
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fc -- /tmp/fetch_calib > /tmp/fc.log 2>&1
tail -1 /tmp/fc.log
python3 - <<'P'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob('/tmp/fc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE': d[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']) * 1024)
N = 1 << 30
for k, v in d.items():
    if k.startswith('fill'): continue
    print('%-12s FETCH_SIZE x 1024 = %s MB per launch; bytes read / counter = %s' % (k, ['%.1f' % (x / 1e6) for x in v], ['%.3f' % (N / x) for x in v]))
P
